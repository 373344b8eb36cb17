#!/usr/bin/env python3
"""Diagnostic (GPU box): the fused 1x1 conv + LayerNorm (+ activation) operator against float64, forward and every gradient, per shape."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spoofsv_amd import ops
torch.manual_seed(0)
rl2 = lambda a, b: float((a.double().cpu() - b).norm() / b.norm())
for (B, Cin, Cout, L, act, with_s) in [(8, 80, 256, 325, 0, False), (8, 512, 513, 1300, 0, False), (4, 512, 513, 1300, 1, False), (4, 513, 513, 1300, 2, False), (4, 256, 256, 325, 1, True), (4, 80, 256, 325, 1, True), (4, 256, 80, 325, 2, False), (4, 128, 512, 186, 1, False)]:
    x = torch.randn(B, Cin, L); w = torch.randn(Cout, Cin, 1) * 0.05; bias = torch.randn(Cout) * 0.1
    gam = torch.rand(Cout) + 0.5; bet = torch.randn(Cout) * 0.3
    s = torch.randn(B, Cout, 1) * 0.2 if with_s else None
    pre = F.conv1d(x.double(), w.double(), bias.double()) + (s.double() if with_s else 0)
    n = F.layer_norm(pre.permute(0, 2, 1), (Cout,), gam.double(), bet.double(), 1e-5).permute(0, 2, 1)
    ref = torch.relu(n) if act == 1 else torch.sigmoid(n) if act == 2 else n
    dy = torch.randn(B, Cout, L)
    ref_in = [t.double().requires_grad_(True) for t in (x, w, bias, gam, bet)]
    pre_r = F.conv1d(ref_in[0], ref_in[1], ref_in[2]) + (s.double() if with_s else 0)
    n_r = F.layer_norm(pre_r.permute(0, 2, 1), (Cout,), ref_in[3], ref_in[4], 1e-5).permute(0, 2, 1)
    (torch.relu(n_r) if act == 1 else torch.sigmoid(n_r) if act == 2 else n_r).backward(dy.double())
    ins = [t.cuda().requires_grad_(True) for t in (x, w, bias, gam, bet)]
    y = ops.pointwise_conv_ln_act(*ins, s.cuda() if with_s else None, act)
    y.backward(dy.cuda())
    torch.cuda.synchronize()
    print("B%d %d->%d L%d act%d s=%s: y %.2e | gradients %s" % (B, Cin, Cout, L, act, with_s, rl2(y.detach(), ref),
          " ".join("%s %.1e" % (n_, rl2(a.grad, b_.grad)) for n_, a, b_ in zip(("x", "w", "bias", "gamma", "beta"), ins, ref_in))))
