#!/bin/bash
# Runs ON THE GPU BOX: one rocprofv3 --pmc pass per argument group over tools/kernel_bench.py.
#   gpurun -- 'bash tools/gpu_pmc.sh tag "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"'
tag=$1; shift
out=$PWD/gpurun_out
i=0
for grp in "$@"; do
  i=$((i+1))
  (cd /tmp && export TMPDIR=/tmp && timeout 150 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/${tag}_pmc$i -o p -- python $OLDPWD/tools/kernel_bench.py > /dev/null 2> $out/${tag}_pmc$i.err)
  python tools/pmc_summary.py $(ls $out/${tag}_pmc$i/*counter_collection.csv | head -1) ehr::vb
done
