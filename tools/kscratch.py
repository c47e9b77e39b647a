"""Scratch (spill) instructions and code size per function of one translation unit's gfx950 ISA:
    python tools/kscratch.py ehr_vbuf.hip [extra hipcc flags]"""
import os, re, subprocess, sys
CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "easyhec_amd", "csrc")
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-mllvm", "-amdgpu-use-amdgpu-trackers=1",
       "-S", "--cuda-device-only"] + sys.argv[2:] + [sys.argv[1], "-o", "/tmp/kscratch.s"]
subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
name, cnt, lines = None, {}, {}
for line in open("/tmp/kscratch.s"):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        name = m.group(1)
        cnt[name] = [0, 0]
        lines[name] = 0
    elif line.startswith(".Lfunc_end"):
        name = None
    elif name:
        if not line.lstrip().startswith((";", ".")):
            lines[name] += 1
        if "scratch_store" in line:
            cnt[name][0] += 1
        if "scratch_load" in line:
            cnt[name][1] += 1
for k, (a, b) in cnt.items():
    short = re.sub(r"^_ZN3ehrL?\d+", "", k)[:40]
    print(f"{short:42s} instructions {lines[k]:6d}  scratch_store {a:4d}  scratch_load {b:4d}")
