set +e
echo "== soak"; timeout 600 python tools/soak.py --scale 0.25 2>&1 | tail -6
echo "== online_loop"; timeout 300 python tools/online_loop.py --explore-iters 2 --epochs 300 --candidates 200 2>&1 | tail -3
echo "== score_bench"; timeout 300 python tools/score_bench.py --reps 2 2>&1 | tail -2
echo "== gpu_check"; timeout 300 python tools/gpu_check.py 2>&1 | tail -4
echo "== traj_check"; timeout 300 python tools/traj_check.py 2>&1 | tail -3
echo "== kernel_bench"; timeout 200 python tools/kernel_bench.py 2>&1 | tail -2
