"""profiles/counters.json: what binds the dominant kernel besides bandwidth -- VALU issue utilisation from the SQ counters,
the kernel's register file use and spills from the compiler's resource-usage remarks.  Inputs (all produced by committed
tools, see tools/gpu_prof.sh and tools/kres.py):

    python tools/make_counters.py <tag>_sq_counters.csv <tag>_kernel_us.csv <kres.txt> <commit>

valu_util = SQ_INSTS_VALU x 4 cycles / (SIMDs x kernel cycles): a wave64 VALU instruction occupies its SIMD's issue port for
four cycles, the chip has 256 CUs x 4 SIMDs, kernel cycles = rocprofv3's mean duration x the 2.4 GHz engine clock."""
import csv
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

KERNEL = "vb_job_kernel<false, false, false>"  # (COVER, MERGE, LAZY)
SIMDS, CLOCK_GHZ = 1024, 2.4


def main():
    sq, us, kres, commit = sys.argv[1:5]
    cnt = {}
    for r in csv.DictReader(open(sq)):
        if KERNEL in r["kernel"]:
            cnt[r["counter"]] = float(r["mean_per_launch"])
    dur = None
    chain = {}
    for r in csv.DictReader(open(us)):
        if "ehr::vb_" in r["kernel"] and int(r["launches"]) > 10:
            chain[r["kernel"].replace("void ", "")] = float(r["mean_us"])
        if KERNEL in r["kernel"]:
            dur = float(r["mean_us"])
    res = {}
    for line in open(kres):
        m = re.match(r"(\S+)\s+VGPR (\S+) SGPR (\S+) spillV (\S+) spillS (\S+) scratch (\S+) LDS (\S+) occ (\S+)", line)
        if m:
            res[m.group(1)] = {"vgprs": int(m.group(2)), "spilled_vgprs": int(m.group(4)), "spilled_sgprs": int(m.group(5)),
                               "scratch_bytes_per_lane": int(m.group(6)), "lds_bytes": int(m.group(7)), "waves_per_simd": int(m.group(8))}
    job = next((v for k, v in res.items() if k.startswith("vb_job_kernelILb0ELb0ELb0E")), {})
    cycles = dur * 1e-6 * CLOCK_GHZ * 1e9
    from bench import csrc_sha16
    out = {"commit": commit, "csrc_sha16": csrc_sha16(), "kernel": KERNEL, "workload": "xarm7_1280x720_8view",
           "kernel_us_rocprof": dur, "chain_kernels_us": chain, "chain_total_us": round(sum(chain.values()), 2),
           "SQ_INSTS_VALU": cnt.get("SQ_INSTS_VALU"), "SQ_INSTS_SALU": cnt.get("SQ_INSTS_SALU"), "SQ_INSTS_LDS": cnt.get("SQ_INSTS_LDS"),
           "SQ_ACTIVE_INST_ANY": cnt.get("SQ_ACTIVE_INST_ANY"), "SQ_WAVE_CYCLES": cnt.get("SQ_WAVE_CYCLES"), "SQ_WAVES": cnt.get("SQ_WAVES"),
           "simds": SIMDS, "clock_ghz": CLOCK_GHZ,
           "valu_util": round(cnt.get("SQ_INSTS_VALU", 0.0) * 4.0 / (SIMDS * cycles), 4),
           "wave_active_frac": round(cnt.get("SQ_ACTIVE_INST_ANY", 0.0) / max(cnt.get("SQ_WAVE_CYCLES", 1.0), 1.0), 4),
           "resources": job, "resources_all_kernels": res,
           "source": "rocprofv3 --pmc SQ_* pass and --kernel-trace pass of bench.py (tools/gpu_prof.sh); hipcc -Rpass-analysis=kernel-resource-usage (tools/kres.py)"}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
