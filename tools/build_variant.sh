#!/bin/bash
# A/B build of the library under ab/: tools/build_variant.sh <name> [extra hipcc flags]   -> ab/libehr_<name>.so
#   EHR_LIB=$PWD/ab/libehr_<name>.so python tools/step_bench.py
n=$1; shift
mkdir -p "$(dirname "$0")/../ab"
cd "$(dirname "$0")/../easyhec_amd/csrc" && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -mllvm -amdgpu-use-amdgpu-trackers=1 -ldl "$@" *.hip -o ../../ab/libehr_$n.so
