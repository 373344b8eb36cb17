#!/usr/bin/env python3
"""Diagnostic (GPU box): isolated time of the weight-gradient launch at the hot layer shapes -- for SSV_ABL tuning builds of the
library (SSV_HIP_LIB=...), whose results are wrong by construction and cannot run inside the training step."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spoofsv_amd
from spoofsv_amd import ops

def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

B = 32
out = []
for (C, L, k, d) in ((256, 325, 3, 3), (512, 186, 3, 3), (512, 1300, 3, 1), (512, 1300, 1, 1), (256, 325, 1, 1)):
    x = torch.randn(B, C, L, device="cuda")
    w = torch.randn(2 * C, C, k, device="cuda") * 0.03
    dy = torch.randn(B, 2 * C, L, device="cuda") * 1e-4
    xa, dya = ops.amax_of(x), ops.amax_of(dy)
    wg = t(lambda: ops._conv_bwd_weight(dy, dy.stride(0), x, x.stride(0), w.shape, k, d, 1, None, dya, xa))
    out.append("C%d L%d k%d %.1f us (%.0f TF/s)" % (C, L, k, wg, 2.0 * B * 2 * C * C * k * L / wg / 1e6))
print(os.environ.get("SSV_HIP_LIB", "default")[-24:], " | ".join(out), flush=True)
