"""profiles/traffic.json from two rocprofv3 PMC passes over tools/step_bench.py -- the launch form bench.py times
(ehr_solver_step, reference masks bound, no mask output).  Run on the GPU box through tools/gpu_traffic.sh:

    gpurun -- 'bash tools/gpu_traffic.sh r03'      # separate FETCH_SIZE / WRITE_SIZE passes, then this script

Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes: the counters are in KB; on gfx950 FETCH_SIZE
reports HALF of the bytes of a wide coalesced read, so  hbm = 2 * FETCH_SIZE + WRITE_SIZE  (an upper bound for kernels
whose loads are narrow gathers)."""
import collections
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHAIN = ["vb_vertex_kernel", "vb_job_kernel", "vb_slow_kernel", "vb_resolve_kernel", "vb_composite_kernel"]
DOMINANT = "vb_job_kernel"


def mean_last(path, counter, last=100):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        for name in CHAIN:
            if name in r["Kernel_Name"]:
                per[(name, r["Dispatch_Id"])] += float(r["Counter_Value"])  # sum the counter's dimensions (XCDs)
                break
    acc = collections.defaultdict(list)
    for (name, d), v in sorted(per.items(), key=lambda kv: int(kv[0][1])):
        acc[name].append(v)
    # a kernel with only a handful of launches is not part of the step (the general-triangle pass runs in the set-up's
    # stateless render calls; the solver step launches it only once a step has needed it): it does not count
    return {k: sum(v[-last:]) / len(v[-last:]) for k, v in acc.items() if len(v) >= last}


def main():
    fetch, write = mean_last(sys.argv[1], "FETCH_SIZE"), mean_last(sys.argv[2], "WRITE_SIZE")
    commit = sys.argv[3] if len(sys.argv) > 3 else "unknown"
    workload = sys.argv[4] if len(sys.argv) > 4 else "xarm7_1280x720_8view"
    kern = {k: {"FETCH_SIZE_KB": round(fetch.get(k, 0.0), 1), "WRITE_SIZE_KB": round(write.get(k, 0.0), 1)} for k in CHAIN}
    hbm = {k: int(round((2.0 * v["FETCH_SIZE_KB"] + v["WRITE_SIZE_KB"]) * 1024)) for k, v in kern.items()}
    sys.path.insert(0, ROOT)
    from bench import algorithmic_bytes_per_frame, csrc_sha16
    from easyhec_amd.robot import load_robot
    from easyhec_amd.synthetic import WORKLOADS
    wl = WORKLOADS[workload]
    B, H, W = wl["views"], wl["H"], wl["W"]
    rb = load_robot(wl.get("robot", "xarm7"))
    alg = algorithmic_bytes_per_frame(rb, H, W) * B
    G = 12 * rb.num_verts + 12 * rb.num_tris
    out = {"round": 6, "commit": commit, "csrc_sha16": csrc_sha16(), "workload": workload,
           "launch_form": "ehr_solver_step, reference masks bound (ehr_fused_bind_ref), mask = NULL: what bench.py times",
           "command": "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python tools/step_bench.py ; the same with --pmc WRITE_SIZE "
                      f"(separate passes, workload {workload}, mean of the last 100 launches of every kernel)",
           "unit": "bytes per step (all kernels of one ehr_solver_step)",
           "correction": "hbm = 2*FETCH_SIZE + WRITE_SIZE (counters in KB -> bytes x1024; gfx950 half-count of wide reads)",
           "kernels_KB": kern,
           "hbm_bytes_per_kernel": hbm,
           "hbm_bytes_dominant_kernel": hbm[DOMINANT],
           "hbm_bytes_whole_op": sum(hbm.values()),
           "algorithmic_bytes_per_launch": alg,
           "ratio_whole_op_to_algorithmic": round(sum(hbm.values()) / alg, 3),
           "needed_bytes_estimate": {
               "geometry_in (packed corner table + indices, once per view)": B * (48 * rb.num_tris),
               "clip_space_vertices (write once, gather)": 2 * B * 16 * rb.num_verts,
               "raster_records (40 B per triangle and view: write + read ~1.7x)": int(2.7 * B * 40 * rb.num_tris),
               "reference_mask (only tiles a link draws into, ~9 % of the image)": int(0.09 * B * H * W * 4),
               "note": "SURVEY 8d's algorithmic figure budgets 16 B per pixel + 2 x geometry (%d B); with the reference bound "
                       "and no mask output the step needs none of the per-pixel traffic outside the links' tiles" % (2 * G)},
           }
    name = "traffic.json" if workload == "xarm7_1280x720_8view" else f"traffic_{workload}.json"
    json.dump(out, open(os.path.join(ROOT, "profiles", name), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
