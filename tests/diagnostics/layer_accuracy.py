#!/usr/bin/env python3
"""Diagnostic (GPU box): gradient error of a stack of n highway layers vs a float64 CPU evaluation, both arithmetic modes."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import spoofsv_amd
from oracle import tts_oracle as TO
from spoofsv_amd import train
from spoofsv_amd.tts import highwayConv
rl2 = lambda a, b: float((a.double().cpu() - b).norm() / b.norm())
torch.manual_seed(0)
B, C, L = 4, 256, 325
for n in (1, 2, 4, 8, 16):
    layers = [highwayConv(C, 3, 3 ** (i % 4), causal=True) for i in range(n)]
    for l in layers:
        l.apply(train.init_weights)
        with torch.no_grad():
            for p in l.parameters():
                if p.dim() == 1: p.add_(0.2 * torch.randn_like(p))
    x = torch.randn(B, C, L); dy = torch.randn(B, C, L)
    sds = [{("hc." + k): v.detach().double().requires_grad_(True) for k, v in l.state_dict().items()} for l in layers]
    xd = x.double().requires_grad_(True)
    h = xd
    for i, sd in enumerate(sds):
        h = TO.highway_conv(h, sd, "hc", 3, 3 ** (i % 4), True)
    h.backward(dy.double())
    for prec in ("f16x2", "bf16x3", "fp32"):
        spoofsv_amd.set_precision(prec)
        ls = [l.cuda() for l in layers]
        for l in ls:
            for p in l.parameters(): p.grad = None
        xg = x.cuda().requires_grad_(True)
        y = xg
        for l in ls: y = l(y)
        y.backward(dy.cuda())
        e = lambda name: max(rl2(dict(ls[i].named_parameters())[name].grad, sds[i]["hc." + name].grad) for i in range(n))
        print("n=%-2d %-6s y %.2e  dx %.2e  worst over layers: conv.w %.2e conv.b %.2e ln1.w %.2e ln1.b %.2e ln2.w %.2e" %
              (n, prec, rl2(y.detach(), h.detach()), rl2(xg.grad, xd.grad), e("conv.weight"), e("conv.bias"), e("ln1.weight"), e("ln1.bias"), e("ln2.weight")), flush=True)
        for l in ls: l.cpu()
