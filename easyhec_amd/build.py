"""Builds libehr_hip.so (hand-written HIP for gfx950) in-tree with hipcc.  No torch headers are needed: the
library's boundary is the plain C ABI of include/ehr.h."""
import glob
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "libehr_hip.so")
ARCH = "gfx950"
# -amdgpu-use-amdgpu-trackers: the target's own register-pressure trackers during scheduling; the job kernel sits at the
# 128-VGPR limit and every spilled register shows in its duration (7 instead of 10 spilled VGPRs, +1 % frames/s)
FLAGS = ["-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-mllvm", "-amdgpu-use-amdgpu-trackers=1", "-ldl"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def needs_build():
    if not os.path.exists(LIB):
        return True
    mt = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(_HERE, "..", "include", "ehr.h")]
    return any(os.path.getmtime(d) > mt for d in deps)


def build(force=False, verbose=False):
    """Compile every HIP translation unit for gfx950 and link libehr_hip.so next to this file."""
    if not force and not needs_build():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libehr_hip.so (the HIP path has no fallback)")
    extra = os.environ.get("EHR_HIPCC_FLAGS", "").split()  # e.g. -DEHR_PHASE_TIMING for tools/phase_profile.py
    cmd = [hipcc, f"--offload-arch={ARCH}"] + FLAGS + extra + sources() + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
