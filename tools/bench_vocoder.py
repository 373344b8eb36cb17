#!/usr/bin/env python3
"""Griffin-Lim vocoder timing at the synthesis size (MAX_FRAME_NUM 325 -> 1300 linear frames, 64 iterations).
Usage: python tools/bench_vocoder.py [B ...]     (default 1 16)"""
import sys, time
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import torch
from spoofsv_amd.vocoder import Vocoder

v = Vocoder(1024, 256)
T = 1300
for B in [int(a) for a in sys.argv[1:]] or [1, 16]:
    S = torch.rand(B, 513, T, device="cuda")
    a = v.random_angles(B, T)
    v.griffinlim(S, a, 2)
    torch.cuda.synchronize()
    t = time.time()
    reps = 3
    for _ in range(reps):
        v.griffinlim(S, a, 64)
    torch.cuda.synchronize()
    ms = (time.time() - t) * 1e3 / reps
    flops = 64 * 2 * 2.0 * 1026 * 1024 * B * T
    print("griffinlim B=%d T=%d 64 it: %.2f ms  (%.2f ms/utterance, DFT GEMMs alone %.0f TFLOP/s-equivalent)" % (B, T, ms, ms / B, flops / ms / 1e9), flush=True)
