"""profiles/traffic.json from two rocprofv3 PMC passes over tools/kernel_bench.py (run on the GPU box):

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o pf -- python tools/kernel_bench.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o pw -- python tools/kernel_bench.py
    python tools/make_traffic.py gpurun_out/pmc_fetch/pf_counter_collection.csv gpurun_out/pmc_write/pw_counter_collection.csv

Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes: the counters are in KB, gfx950 reports HALF
of the fetched bytes in FETCH_SIZE.  The calibration is checked on fused_empty_kernel, whose traffic is known exactly
(one float4 read of the reference mask and one float4 write of the rendered mask per lane)."""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {"fused_vertex_kernel": "fused_vertex_kernel", "bin_kernel<1, false>": "bin_kernel<count>",
         "bin_alloc_kernel": "bin_alloc_kernel", "bin_kernel<1, true>": "bin_kernel<fill>",
         "fused_empty_kernel": "fused_empty_kernel", "fused_tile_kernel<false>": "fused_tile_kernel<lean>",
         "fused_tile_kernel<true>": "fused_tile_kernel<slow>", "fused_finish_kernel": "fused_finish_kernel"}


def mean_last(path, counter, last=50):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        for pat, name in NAMES.items():
            if pat in r["Kernel_Name"]:
                acc[name].append(float(r["Counter_Value"]))
                break
    return {k: sum(v[-last:]) / len(v[-last:]) for k, v in acc.items()}


def main():
    fetch, write = mean_last(sys.argv[1], "FETCH_SIZE"), mean_last(sys.argv[2], "WRITE_SIZE")
    kern = {k: {"FETCH_SIZE": round(fetch.get(k, 0.0), 1), "WRITE_SIZE": round(write.get(k, 0.0), 1)} for k in NAMES.values()}
    hbm = {k: int(round((2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024)) for k, v in kern.items()}
    B, H, W = 8, 720, 1280
    stage = ["fused_empty_kernel", "fused_tile_kernel<lean>", "fused_tile_kernel<slow>"]
    out = {"round": 1,
           "command": "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python tools/kernel_bench.py ; rocprofv3 --pmc WRITE_SIZE "
                      "--kernel-trace -- python tools/kernel_bench.py  (separate passes, 8 views 1280x720 xArm7, mean of the last 50 launches)",
           "unit": "bytes per launch of the fused op",
           "calibration": "fused_empty_kernel streams the empty tiles' reference pixels in and mask pixels out (about 0.91 x %d MB each "
                          "way); FETCH_SIZE reads about half of that (the gfx950 half-count), WRITE_SIZE about all of it.  Correction "
                          "applied: hbm = 2*FETCH_SIZE + WRITE_SIZE (KB -> bytes x1024)." % round(B * H * W * 4 / 1e6, 1),
           "kernels_KB": kern,
           "hbm_bytes_tile_stage": {k: hbm[k] for k in stage},
           "hbm_bytes_per_launch": sum(hbm[k] for k in stage),
           "hbm_bytes_lean_kernel": hbm["fused_tile_kernel<lean>"],
           "hbm_bytes_whole_op": sum(hbm.values()),
           "note": "hbm_bytes_per_launch covers the stage the roofline figure is quoted on (empty-tile stream + work-list tile "
                   "kernels: together they touch every pixel once).  It is below the algorithmic figure because the fused pass reads "
                   "ref and writes mask once per pixel (8 B/px) where SURVEY 8d budgets 16 B/px; the x2 FETCH correction is an upper "
                   "bound for the lean kernel's narrow (4-byte) loads."}
    sys.path.insert(0, ROOT)
    from bench import algorithmic_bytes_per_frame
    from easyhec_amd.robot import load_robot
    out["algorithmic_bytes_per_launch"] = algorithmic_bytes_per_frame(load_robot("xarm7"), H, W) * B
    json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("hbm_bytes_per_launch", "hbm_bytes_lean_kernel", "hbm_bytes_whole_op",
                                          "algorithmic_bytes_per_launch")}))


if __name__ == "__main__":
    main()
