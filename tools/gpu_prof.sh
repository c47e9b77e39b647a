#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): kernel trace + SQ counters of the bench workload into gpurun_out/<tag>_*.
#   gpurun -- 'bash tools/gpu_prof.sh r02_a'
tag=${1:-prof}
out=$PWD/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_trace -o t -- python $OLDPWD/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-side --no-with-mask --no-drop-in > $out/${tag}_bench.json 2> $out/${tag}_trace.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $out/${tag}_sq -o s -- python $OLDPWD/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-side --no-with-mask --no-drop-in > /dev/null 2> $out/${tag}_sq.err
cd $OLDPWD
python tools/pmc_summary.py $(ls $out/${tag}_trace/*kernel_trace.csv | head -1) > $out/${tag}_kernel_us.csv
python tools/pmc_summary.py $(ls $out/${tag}_sq/*counter_collection.csv | head -1) ehr > $out/${tag}_sq_counters.csv
cat $out/${tag}_kernel_us.csv | head -20
