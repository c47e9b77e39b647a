"""Upper bound of what running the two halves of a batch as concurrent chains could buy: two independent 4-view solves
on two streams (no dependency between them at all) against one 8-view solve.
    python tools/overlap_probe.py [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import time
import torch
import bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda", 0)


def run(trs, streams):
    for _ in range(20):
        for tr, s in zip(trs, streams):
            with torch.cuda.stream(s):
                tr.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for tr, s in zip(trs, streams):
            with torch.cuda.stream(s):
                tr.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6


for graph in (True, False):
    bench.VIEWS_PER_GPU = 8
    p8 = bench.build_problem(0, 1, dev, graph=graph)
    t8 = run([p8["trainer"]], [torch.cuda.Stream()])
    bench.VIEWS_PER_GPU = 4
    pa = bench.build_problem(0, 1, dev, graph=graph)
    pb = bench.build_problem(0, 1, dev, graph=graph)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    t4 = run([pa["trainer"]], [sa])
    t44 = run([pa["trainer"], pb["trainer"]], [sa, sb])
    t44s = run([pa["trainer"], pb["trainer"]], [sa, sa])
    print(f"graph={graph}: 8 views {t8:.1f} us | 4 views {t4:.1f} us | 2 x 4 views, two streams {t44:.1f} us | "
          f"2 x 4 views, one stream {t44s:.1f} us", flush=True)
