#!/usr/bin/env python3
"""Experiment (GPU box): how much a replayed training step slows down while G workgroups of another stream stay resident (what RCCL's
channels do during an overlapped all-reduce).  build/ab/cu_steal.so = tools/cu_steal.hip."""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "build", "ab", "cu_steal.so"))
lib.cu_steal.argtypes = [ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p]
side = torch.cuda.Stream()
for kind in ("text2mel", "ssrn"):
    tr = bench.Trainer(kind, 32, dev, 0, 1, True); tr.prepare()
    for _ in range(5): tr.step()
    torch.cuda.synchronize()
    line = kind + ":"
    for G in (0, 8, 16, 32, 64):
        torch.cuda.synchronize()
        if G:
            lib.cu_steal(G, int(0.25 * 2.0e9), ctypes.c_void_p(side.cuda_stream))      # ~125-250 ms resident
        time.sleep(0.005)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8): tr.step()
        e1.record(); e1.synchronize()
        line += "  G=%d %.3f ms" % (G, e0.elapsed_time(e1) / 8)
        torch.cuda.synchronize()
    print(line, flush=True)
    del tr; torch.cuda.empty_cache()
