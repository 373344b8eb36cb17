#!/usr/bin/env python3
"""Profiling target: column-incremental free run at full size (326 frames).  python tools/prof_synth.py [batch]"""
import sys, time
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import torch
from spoofsv_amd import harness, train
from spoofsv_amd.tts import melSyn
dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 20
torch.manual_seed(1234)
m = melSyn(34, True, 200, 128, 80, 256); m.apply(train.init_weights); m = m.to(dev).eval()
N, frames = 80, 326
text = torch.randint(2, 33, (B, 1, N), device=dev); text[:, :, -1] = 1
spk = 0.04 + 0.05 * torch.rand(B, 200, 1, device=dev)
with torch.no_grad():
    harness._free_run(m, text, spk, frames, 80, incremental=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    harness._free_run(m, text, spk, frames, 80, incremental=True)
    torch.cuda.synchronize(); print("B=%d incremental free run: %.3f ms/frame" % (B, (time.perf_counter() - t0) / frames * 1e3), flush=True)
