"""Mean per launch of every counter / kernel in a rocprofv3 *_counter_collection.csv, or of the durations in a
*_kernel_trace.csv:   python tools/pmc_summary.py <csv> [kernel-substring]"""
import collections
import csv
import sys


def main():
    path = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    rows = list(csv.DictReader(open(path)))
    if rows and "Counter_Name" in rows[0]:
        # one row per (dispatch, counter, [dimension]); sum the dimensions of a dispatch, then average over dispatches
        per = collections.defaultdict(float)
        for r in rows:
            if pat in r["Kernel_Name"]:
                per[(r["Kernel_Name"].split("(")[0], r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
        acc = collections.defaultdict(list)
        for (k, c, _), v in per.items():
            acc[(k, c)].append(v)
        print("kernel,counter,mean_per_launch,launches")
        for (k, c), v in sorted(acc.items()):
            print(f'"{k}",{c},{sum(v) / len(v):.0f},{len(v)}')
    else:
        acc = collections.defaultdict(list)
        for r in rows:
            if pat in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        print("kernel,launches,mean_us,min_us,max_us")
        for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
            print(f'"{k}",{len(v)},{sum(v) / len(v):.2f},{min(v):.2f},{max(v):.2f}')


if __name__ == "__main__":
    main()
