"""Static instruction census of the kernels of one HIP translation unit (slow multiplies, divisions, LDS shuffles, DPP):
    python tools/isa_stats.py ehr_vbuf.hip [kernel-name-substring ...]"""
import os, re, subprocess, sys
from collections import Counter
CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "easyhec_amd", "csrc")
out = "/tmp/isa_stats.s"
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-mllvm",
                       "-amdgpu-use-amdgpu-trackers=1", "-S", "--cuda-device-only", sys.argv[1], "-o", out], cwd=CSRC,
                      stderr=subprocess.DEVNULL)
s = open(out).read()
for m in re.finditer(r"\n(_ZN3ehr\S+):\s*; @", s):
    n = m.group(1)
    if len(sys.argv) > 2 and not any(k in n for k in sys.argv[2:]):
        continue
    i = m.start()
    j = s.index(".Lfunc_end", i)
    f = s[i:j]
    c = Counter(re.findall(r"^\s+((?:v|s|ds|global|scratch|buffer)_[a-z0-9_]+)", f, re.M))
    print(re.sub(r"^_ZN3ehrL?\d+", "", n)[:34], "instr", sum(c.values()), "| bpermute", c["ds_bpermute_b32"], "swizzle", c["ds_swizzle_b32"],
          "dpp", len(re.findall(r"row_shr|row_ror|quad_perm|row_bcast|row_mirror|row_half", f)), "| mul_lo", c["v_mul_lo_u32"],
          "mul_hi", c["v_mul_hi_u32"] + c["v_mul_hi_i32"], "mad64", c["v_mad_u64_u32"] + c["v_mad_i64_i32"], "| int-div", c["v_rcp_iflag_f32_e32"],
          "f-div", c["v_div_scale_f32"] // 2, "| readlane", c["v_readlane_b32"], "writelane", c["v_writelane_b32"], "nop", c["s_nop"])
