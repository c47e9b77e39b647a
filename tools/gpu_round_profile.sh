#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): everything profiles/ holds for one state of the kernels, under one tag.
#   gpurun --timeout 2400 -- 'bash tools/gpu_round_profile.sh r05_b <commit>'
# kernel trace + SQ counters of bench.py (gpu_prof.sh), HBM traffic of the timed launch form (gpu_traffic.sh), the compiler's
# resource report, counters.json / traffic.json (they name the commit and the SHA-256 of easyhec_amd/csrc they describe),
# kernel traces of the two side workloads the 60 % target is met on, the three-op step and the scoring op.
tag=${1:-prof}; commit=${2:-unknown}
out=$PWD/gpurun_out
bash tools/gpu_prof.sh $tag > $out/${tag}_prof.log 2>&1
bash tools/gpu_traffic.sh $tag $commit > $out/${tag}_traffic.log 2>&1
python tools/kres.py ehr_vbuf.hip > $out/${tag}_kernel_resources.txt 2>&1
python tools/make_counters.py $out/${tag}_sq_counters.csv $out/${tag}_kernel_us.csv $out/${tag}_kernel_resources.txt $commit > $out/${tag}_counters.json
for w in franka_1920x1080_16view xarm7_1280x720_64view xarm7_640x480_1view; do
  bash tools/gpu_traffic.sh ${tag}_$w $commit $w > $out/${tag}_${w}_traffic.log 2>&1   # -> profiles/traffic_<workload>.json (+ a copy under gpurun_out/)
  (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_$w -o t -- python $OLDPWD/tools/step_bench.py $w 200 > $out/${tag}_${w}_bench.txt 2> $out/${tag}_$w.err)
  python tools/pmc_summary.py $(ls $out/${tag}_$w/*kernel_trace.csv | head -1) ehr > $out/${tag}_${w}_kernel_us.csv
  rm -rf $out/${tag}_$w   # (raw traces are tens of MB: gpurun copies back at most 64 MiB)
done
for v in three_ops three_ops_batched import_swap_only; do
  (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_$v -o t -- python $OLDPWD/tools/three_op_bench.py --graph --steps 30 --only $v > /dev/null 2> $out/${tag}_$v.err)
  python tools/pmc_summary.py $(ls $out/${tag}_$v/*kernel_trace.csv | head -1) > $out/${tag}_${v}_graph_kernel_us.csv
  rm -rf $out/${tag}_$v
done
python tools/three_op_bench.py > $out/${tag}_three_op_bench.txt 2>&1
python tools/three_op_bench.py --graph --steps 50 >> $out/${tag}_three_op_bench.txt 2>&1
python tools/score_bench.py > $out/${tag}_score_bench.json 2>/dev/null
cp $out/${tag}_trace/t_kernel_stats.csv $out/${tag}_solver_step_kernel_stats.csv
rm -rf $out/${tag}_trace $out/${tag}_sq $out/${tag}_traffic1 $out/${tag}_traffic2
timeout 900 python bench.py > $out/${tag}_bench_full.json 2> $out/${tag}_bench_full.err
tail -3 $out/${tag}_prof.log; tail -3 $out/${tag}_traffic.log; head -30 $out/${tag}_counters.json
