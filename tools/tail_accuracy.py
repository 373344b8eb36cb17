#!/usr/bin/env python3
"""Diagnostic (GPU box): gradient error of the 1x1-conv + LayerNorm (+ activation) blocks and the loss head vs float64."""
import os, sys
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import spoofsv_amd
from spoofsv_amd import ops
rl2 = lambda a, b: float((a.double().cpu() - b).norm() / b.norm())
torch.manual_seed(0)
B, L = 4, 325


def ref_block(x, w, b, g, be, act):
    h = F.conv1d(x, w, b)
    y = F.layer_norm(h.permute(0, 2, 1), (w.shape[0],), g, be, 1e-5).permute(0, 2, 1)
    return torch.relu(y) if act == 1 else (torch.sigmoid(y) if act == 2 else y)


for name, dims, acts, loss in [("tail 256-256-256-80 + bd/l1 loss", (256, 256, 256, 80), (1, 1, 2), True),
                               ("head 80-256-256", (80, 256, 256), (1, 0), False),
                               ("one 256-256 relu", (256, 256), (1,), False),
                               ("one 256-80 sigmoid + loss", (256, 80), (2,), True),
                               ("one 256-80 sigmoid, random dy", (256, 80), (2,), False),
                               ("one 256-80 none, random dy", (256, 80), (0,), False),
                               ("one 80-256 none", (80, 256), (0,), False),
                               ("ssrn tail 512-513-513 + loss", (512, 513, 513), (1, 2), True)]:
    ps = []
    for cin, cout in zip(dims[:-1], dims[1:]):
        ps.append([torch.randn(cout, cin, 1) * (2.0 / cin) ** 0.5, 0.1 * torch.randn(cout), 1 + 0.2 * torch.randn(cout), 0.2 * torch.randn(cout)])
    x = torch.randn(B, dims[0], L); gt = torch.rand(B, dims[-1], L); dy = torch.randn(B, dims[-1], L)
    pd = [[t.double().requires_grad_(True) for t in p] for p in ps]
    xd = x.double().requires_grad_(True)
    h = xd
    for p, a in zip(pd, acts): h = ref_block(h, *p, a)
    if loss:
        g64 = gt.double()
        (torch.mean(torch.abs(g64 - h)) + torch.mean(-g64 * torch.log(h + 1e-8) - (1 - g64) * torch.log(1 - h + 1e-8))).backward()
    else:
        h.backward(dy.double())
    for prec in ("bf16x3", "fp32"):
        spoofsv_amd.set_precision(prec)
        pg = [[t.cuda().requires_grad_(True) for t in p] for p in ps]
        xg = x.cuda().requires_grad_(True)
        y = xg
        for p, a in zip(pg, acts): y = ops.pointwise_conv_ln_act(y, p[0], p[1], p[2], p[3], None, a)
        if loss:
            l1, bd = ops.spec_losses(y, gt.cuda()); (l1 + bd).backward()
        else:
            y.backward(dy.cuda())
        errs = ["L%d w %.1e b %.1e g %.1e be %.1e" % (i, rl2(q[0].grad, r[0].grad), rl2(q[1].grad, r[1].grad) if float(r[1].grad.norm()) > 0 else -1, rl2(q[2].grad, r[2].grad), rl2(q[3].grad, r[3].grad))
                for i, (q, r) in enumerate(zip(pg, pd))]
        print("%-34s %-6s y %.1e dx %.1e | %s" % (name, prec, rl2(y.detach(), h.detach()), rl2(xg.grad, xd.grad), " | ".join(errs)), flush=True)
