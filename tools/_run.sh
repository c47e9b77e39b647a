set -x
timeout 600 bash tools/gpu_prof.sh r02_a
timeout 700 bash tools/gpu_pmc.sh r02 "FETCH_SIZE" "WRITE_SIZE"
timeout 120 python tools/make_traffic.py gpurun_out/r02_pmc1/p_counter_collection.csv gpurun_out/r02_pmc2/p_counter_collection.csv
timeout 300 python bench.py > gpurun_out/r02_a_bench_default.json 2> gpurun_out/r02_a_bench_default.err
cat gpurun_out/r02_a_bench_default.json
timeout 300 python tools/three_op_bench.py > gpurun_out/r02_three_op.txt 2>&1
tail -5 gpurun_out/r02_three_op.txt
cp profiles/traffic.json gpurun_out/traffic.json
