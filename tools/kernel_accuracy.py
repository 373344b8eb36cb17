#!/usr/bin/env python3
"""Diagnostic (GPU box): relative L2 error of the split-bf16 and exact-fp32 conv kernels (forward, data gradient, weight
gradient) against a float64 CPU reference, random operands, real layer shapes."""
import ctypes, os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spoofsv_amd
from spoofsv_amd import ops
torch.manual_seed(0)
rl2 = lambda a, b: float((a.double().cpu() - b).norm() / b.norm())
for (B, C, L, k, d, causal) in [(4, 256, 325, 3, 1, True), (4, 256, 325, 3, 27, True), (4, 512, 186, 3, 3, False), (2, 256, 1300, 3, 3, False), (4, 256, 325, 1, 1, False)]:
    x = torch.randn(B, C, L); w = torch.randn(2 * C, C, k) * 0.03; dy = torch.randn(B, 2 * C, L)
    pad = d * (k - 1)
    xd = x.double().requires_grad_(True); wd = w.double().requires_grad_(True)
    xin = F.pad(xd, (pad, 0)) if causal else F.pad(xd, (pad // 2, pad // 2))
    yd = F.conv1d(xin, wd, None, dilation=d)
    yd.backward(dy.double())
    for prec in ("bf16x3", "fp32"):
        spoofsv_amd.set_precision(prec)
        xg = x.cuda().requires_grad_(True); wg = w.cuda().requires_grad_(True)
        y = ops.conv1d(xg, wg, None, k, d, causal)
        y.backward(dy.cuda())
        print("B%d C%d L%d k%d d%-2d %-6s  fwd %.2e  dgrad %.2e  wgrad %.2e" % (B, C, L, k, d, prec, rl2(y.detach(), yd.detach()), rl2(xg.grad, xd.grad), rl2(wg.grad, wd.grad)), flush=True)
