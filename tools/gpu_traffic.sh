#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): HBM traffic of the timed launch form, FETCH_SIZE and WRITE_SIZE in separate
# rocprofv3 --pmc passes over tools/step_bench.py, then profiles/traffic.json (copied back through gpurun_out/).
#   gpurun -- 'bash tools/gpu_traffic.sh r03 <commit> [workload]'      (workload: a key of easyhec_amd.synthetic.WORKLOADS;
#   the headline writes profiles/traffic.json, any other one profiles/traffic_<workload>.json)
tag=${1:-traffic}; commit=${2:-unknown}; wl=${3:-xarm7_1280x720_8view}
out=$PWD/gpurun_out
i=0
for c in FETCH_SIZE WRITE_SIZE; do
  i=$((i+1))
  (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/${tag}_traffic$i -o p -- python $OLDPWD/tools/step_bench.py $wl 150 > /dev/null 2> $out/${tag}_traffic$i.err)
done
python tools/make_traffic.py $(ls $out/${tag}_traffic1/*counter_collection.csv | head -1) $(ls $out/${tag}_traffic2/*counter_collection.csv | head -1) $commit $wl > $out/${tag}_traffic.json
if [ "$wl" = "xarm7_1280x720_8view" ]; then cp profiles/traffic.json $out/traffic.json; else cp profiles/traffic_$wl.json $out/traffic_$wl.json; fi
rm -rf $out/${tag}_traffic1 $out/${tag}_traffic2
