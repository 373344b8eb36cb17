#!/usr/bin/env python3
"""Diagnostic (GPU box): relative L2 error of the split-fp16, split-bf16 and exact-fp32 conv kernels (forward, data gradient,
weight gradient) against a float64 CPU reference on real layer shapes -- unit-scale operands, and "wide" ones shaped like
gradients (tiny overall scale, per-row magnitudes spread over 2^12, a few outliers) to exercise the power-of-two scales."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spoofsv_amd
from spoofsv_amd import ops

torch.manual_seed(0)
rl2 = lambda a, b: float((a.double().cpu() - b).norm() / b.norm())
SHAPES = [(4, 256, 325, 3, 1, True), (4, 256, 325, 3, 27, True), (4, 512, 186, 3, 3, False), (2, 256, 1300, 3, 3, False), (4, 256, 325, 1, 1, False)]
for (B, C, L, k, d, causal) in SHAPES:
    for wide in (False, True):
        x = torch.randn(B, C, L)
        w = torch.randn(2 * C, C, k) * 0.03
        dy = torch.randn(B, 2 * C, L)
        if wide:
            dy = dy * 1e-7 * torch.exp2(torch.randint(-12, 1, (B, 2 * C, 1)).float())
            dy[0, 0, :3] = 3e-5
            x = x * torch.exp2(torch.randint(-8, 3, (B, C, 1)).float())
        pad = d * (k - 1)
        xd = x.double().requires_grad_(True)
        wd = w.double().requires_grad_(True)
        xin = F.pad(xd, (pad, 0)) if causal else F.pad(xd, (pad // 2, pad // 2))
        yd = F.conv1d(xin, wd, None, dilation=d)
        yd.backward(dy.double())
        for prec in ("f16x2", "bf16x3", "fp32"):
            spoofsv_amd.set_precision(prec)
            xg = x.cuda().requires_grad_(True)
            wg = w.cuda().requires_grad_(True)
            y = ops.conv1d(xg, wg, None, k, d, causal)
            y.backward(dy.cuda())
            print("B%d C%d L%d k%d d%-2d %s %-6s  fwd %.2e  dgrad %.2e  wgrad %.2e" % (
                B, C, L, k, d, "wide" if wide else "unit", prec, rl2(y.detach(), yd.detach()), rl2(xg.grad, xd.grad), rl2(wg.grad, wd.grad)), flush=True)
