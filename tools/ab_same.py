#!/usr/bin/env python3
"""Tuning aid (GPU box): conv forward / data gradient / weight gradient outputs of the library named by SSV_HIP_LIB on fixed seeded
inputs, saved to (AB_SAVE=1) or compared bit for bit with build/ab/ab_same.pt -- for changes that must not change a single bit."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spoofsv_amd import ops
dev = "cuda:0"
torch.manual_seed(7)
outs = {}
for (B, C, L, k, dil) in [(4, 256, 325, 3, 3), (3, 200, 777, 3, 27), (2, 80, 130, 1, 1), (2, 512, 1300, 3, 1)]:
    x = (torch.randn(B, C, L, device=dev) * torch.logspace(-6, 3, L, device=dev)).requires_grad_(True)
    w = (torch.randn(2 * C, C, k, device=dev) * 0.05).requires_grad_(True)
    bias = torch.randn(2 * C, device=dev, requires_grad=True)
    y = ops.conv1d(x, w, bias, k, dil, False)
    gy = torch.randn_like(y)
    gx, gw, gb = torch.autograd.grad(y, (x, w, bias), gy)
    for n, t in (("y", y), ("gx", gx), ("gw", gw), ("gb", gb)):
        outs["%s_%d_%d_%d_%d" % (n, C, L, k, dil)] = t.detach().cpu()
for (B, C, L, k, dil, causal) in [(4, 256, 325, 3, 3, False), (2, 512, 1300, 3, 1, True), (3, 200, 100, 1, 1, False), (2, 513, 300, 1, 1, False)]:
    x = torch.randn(B, C, L, device=dev, requires_grad=True)
    w = (torch.randn(2 * C, C, k, device=dev) * 0.05).requires_grad_(True)
    ps = [torch.randn(n, device=dev, requires_grad=True) for n in (2 * C, C, C, C, C)]
    if C <= 512:
        y = ops.highway_conv1d(x, w, ps[0], ps[1], ps[2], ps[3], ps[4], k, dil, causal)
        gs = torch.autograd.grad(y, [x, w] + ps, torch.randn_like(y))
        for n, t in zip(["hy", "hgx", "hgw", "hgb", "hg1", "hb1", "hg2", "hb2"], (y,) + gs):
            outs["%s_%d_%d_%d" % (n, C, L, k)] = t.detach().cpu()
    w1 = (torch.randn(C, C, 1, device=dev) * 0.05).requires_grad_(True)
    for act in (0, 1, 2):
        y = ops.pointwise_conv_ln_act(x, w1, ps[1], ps[2], ps[3], None, act)
        gs = torch.autograd.grad(y, [x, w1, ps[1], ps[2], ps[3]], torch.randn_like(y))
        for n, t in zip(["py", "pgx", "pgw", "pgb", "pgg", "pgbt"], (y,) + gs):
            outs["%s%d_%d_%d" % (n, act, C, L)] = t.detach().cpu()
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "build", "ab", "ab_same.pt")
if os.environ.get("AB_SAVE") == "1":
    torch.save(outs, path); print("saved", len(outs))
else:
    ref = torch.load(path)
    bad = [k for k in outs if not torch.equal(outs[k], ref[k])]
    print("bit-identical" if not bad else "DIFFERENT: %s" % ", ".join("%s %.3g" % (k, (outs[k] - ref[k]).abs().max()) for k in bad))
