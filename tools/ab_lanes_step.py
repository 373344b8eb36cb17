#!/usr/bin/env python3
"""Experiment: upper bound of "batch lanes" on the whole step -- two independent trainers at B=16 replayed side by side on two
streams against one trainer at B=32 (per model).  Two model copies, so weight traffic is doubled: pessimistic for lanes."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
for kind in ("text2mel", "ssrn"):
    full = bench.Trainer(kind, 32, dev, 0, 1, True); full.prepare()
    a = bench.Trainer(kind, 16, dev, 0, 1, True); a.prepare()
    b = bench.Trainer(kind, 16, dev, 1, 1, True); b.prepare()
    def lanes():
        with torch.cuda.stream(sa): a.step()
        with torch.cuda.stream(sb): b.step()
    def serial():
        a.step(); b.step()
    print("%s: B=32 %.3f ms | two B=16 side by side %.3f ms | two B=16 back to back %.3f ms" % (kind, timeit(full.step), timeit(lanes), timeit(serial)), flush=True)
    del full, a, b
    torch.cuda.empty_cache()
