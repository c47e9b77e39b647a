"""Kernel resource usage of one HIP translation unit (registers, spills, LDS, occupancy):
    python tools/kres.py ehr_vbuf.hip [extra hipcc flags]"""
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "easyhec_amd", "csrc")
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-mllvm",
       "-amdgpu-use-amdgpu-trackers=1"] + sys.argv[2:] + ["-Rpass-analysis=kernel-resource-usage", sys.argv[1], "-o", "/tmp/kres.so"]
out = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
cur, rows = None, {}
for line in out.stderr.splitlines():
    m = re.search(r"remark: +(.*?) \[-Rpass", line)
    if not m:
        if "error" in line:
            print(line)
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = t.split(":", 1)[1].strip()
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1)
        rows[cur][k.strip()] = v.strip()
for k, v in rows.items():
    name = re.sub(r"^_ZN3ehrL?\d+", "", k)[:34]
    print(f"{name:36s} " + " ".join(f"{a} {v.get(b)}" for a, b in [
        ("VGPR", "VGPRs"), ("SGPR", "SGPRs"), ("spillV", "VGPRs Spill"), ("spillS", "SGPRs Spill"),
        ("scratch", "ScratchSize [bytes/lane]"), ("LDS", "LDS Size [bytes/block]"), ("occ", "Occupancy [waves/SIMD]")]))
