#!/usr/bin/env python3
"""Diagnostic (GPU box): does the row alignment of the (B, C, L) operands matter to the conv kernels?  Times the forward,
data-gradient and weight-gradient launches of one highway-sized conv at neighbouring lengths (rows 16-byte aligned when L % 4 == 0)."""
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
for (C, k, d) in ((256, 3, 3), (512, 3, 3), (256, 1, 1)):
    for L in ((320, 324, 325, 326, 328) if C == 256 else (184, 186, 188, 192)):
        x = torch.randn(B, C, L, device="cuda")
        w = torch.randn(2 * C, C, k, device="cuda") * 0.03
        dy = torch.randn(B, 2 * C, L, device="cuda")
        xa, dya = ops.amax_of(x), ops.amax_of(dy)
        f = t(lambda: ops.conv1d(x, w, None, k, d, True))
        wg = t(lambda: ops._conv_bwd_weight(dy, dy.stride(0), x, x.stride(0), w.shape, k, d, 1, None, dya, xa))
        print("C=%d k=%d L=%d   fwd %.1f us (%.0f TF/s)   wgrad %.1f us (%.0f TF/s)" % (
            C, k, L, f, 2.0 * B * 2 * C * C * k * L / f / 1e6, wg, 2.0 * B * 2 * C * C * k * L / wg / 1e6), flush=True)
