#!/usr/bin/env python3
"""Experiment: the Text2Mel step and the SSRN step of bench.py (independent models) replayed back to back on one stream
against replayed side by side on two streams."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
t2m = bench.Trainer("text2mel", 32, dev, 0, 1, True); ssr = bench.Trainer("ssrn", 32, dev, 0, 1, True)
t2m.prepare(); ssr.prepare()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def serial():
    t2m.step(); ssr.step()
def side():
    with torch.cuda.stream(sa): t2m.step()
    with torch.cuda.stream(sb): ssr.step()
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for _ in range(2):
    print("serial %.3f ms   side by side %.3f ms" % (timeit(serial), timeit(side)), flush=True)
print("losses", t2m.loss, ssr.loss)
