#!/usr/bin/env python3
"""Diagnostic (GPU box): the weight gradient of the last 1x1 conv of the decoder tail, operands taken from a float64 run of
the chain (x = ReLU output of the previous block, dpre = LayerNorm/sigmoid/loss backward): kernel error vs float64 and the
cancellation factor  sum|terms| / |sum terms|  of the reduction."""
import ctypes, os, sys
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import spoofsv_amd
from spoofsv_amd import ops
torch.manual_seed(0)
B, L = 4, 325
dims, acts = (256, 256, 256, 80), (1, 1, 2)
ps = [[torch.randn(co, ci, 1, dtype=torch.float64) * (2.0 / ci) ** 0.5, 0.1 * torch.randn(co, dtype=torch.float64), 1 + 0.2 * torch.randn(co, dtype=torch.float64),
       0.2 * torch.randn(co, dtype=torch.float64)] for ci, co in zip(dims[:-1], dims[1:])]
x = torch.randn(B, dims[0], L, dtype=torch.float64); gt = torch.rand(B, dims[-1], L, dtype=torch.float64)
for p in ps:
    for t in p: t.requires_grad_(True)
h = x
keep = []
for p, a in zip(ps, acts):
    pre = F.conv1d(h, p[0], p[1]); pre.retain_grad()
    y = F.layer_norm(pre.permute(0, 2, 1), (p[0].shape[0],), p[2], p[3], 1e-5).permute(0, 2, 1)
    keep.append((h, pre))
    h = torch.relu(y) if a == 1 else torch.sigmoid(y)
(torch.mean(torch.abs(gt - h)) + torch.mean(-gt * torch.log(h + 1e-8) - (1 - gt) * torch.log(1 - h + 1e-8))).backward()
for i, (xin, pre) in enumerate(keep):
    dpre = pre.grad
    dw64 = torch.einsum("bmt,bct->mc", dpre, xin.detach())
    absum = torch.einsum("bmt,bct->mc", dpre.abs(), xin.detach().abs())
    amp = float(absum.norm() / dw64.norm())
    line = "block %d: cancellation sum|terms|/|sum| = %.0f (x mean %.2f rms %.2f; dpre rms %.2e)" % (i, amp, float(xin.mean()), float(xin.pow(2).mean().sqrt()), float(dpre.pow(2).mean().sqrt()))
    for prec in ("bf16x3", "fp32"):
        spoofsv_amd.set_precision(prec)
        dw = ops._conv_bwd_weight(dpre.float().cuda().contiguous(), dpre.shape[1] * L, xin.detach().float().cuda().contiguous(), xin.shape[1] * L, (dpre.shape[1], xin.shape[1], 1))
        e = float((dw[:, :, 0].double().cpu() - dw64).norm() / dw64.norm())
        # the same operands rounded to float32 first (what any float32 implementation sees)
        dw32in = torch.einsum("bmt,bct->mc", dpre.float().double(), xin.detach().float().double())
        line += " | %s err %.1e (fp32-rounded inputs alone: %.1e)" % (prec, e, float((dw32in - dw64).norm() / dw64.norm()))
    print(line, flush=True)
