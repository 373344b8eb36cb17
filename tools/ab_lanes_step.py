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
# the two models' steps side by side (independent models: what a node training both could do)
t2m = bench.Trainer("text2mel", 32, dev, 0, 1, True); t2m.prepare()
ssr = bench.Trainer("ssrn", 32, dev, 0, 1, True); ssr.prepare()
def both_serial():
    t2m.step(); ssr.step()
def both_side():
    with torch.cuda.stream(sa): t2m.step()
    with torch.cuda.stream(sb): ssr.step()
print("text2mel + ssrn (B=32 each): back to back %.3f ms | side by side on two streams %.3f ms" % (timeit(both_serial), timeit(both_side)), flush=True)
