#!/usr/bin/env python3
"""Experiment (GPU box): does the chip have idle capacity that two half-batch steps running side by side could use?  One SSRN (or Text2Mel)
trainer at B = 32 against two independent trainers at B = 16 replayed on two streams at once, and against the same two replayed one after the other."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda", 0)
kind = sys.argv[1] if len(sys.argv) > 1 else "ssrn"
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
full = bench.Trainer(kind, 32, dev, 0, 1, True); full.prepare()
t_full = timeit(full.step)
a = bench.Trainer(kind, 16, dev, 0, 1, True); a.prepare()
b = bench.Trainer(kind, 16, dev, 1, 1, True); b.prepare()
t_half = timeit(a.step)
def serial(): a.step(); b.step()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def conc():
    with torch.cuda.stream(s1): a.step()
    with torch.cuda.stream(s2): b.step()
t_ser = timeit(serial); t_con = timeit(conc)
print("%s: B=32 one step %.3f ms | B=16 one step %.3f ms | two B=16 steps serial %.3f ms | two B=16 steps on two streams %.3f ms" % (kind, t_full, t_half, t_ser, t_con))
