#!/usr/bin/env python3
"""Random-shape check of the conv kernels' three products against float64 (GPU box): B, channels, length, dilation, causal drawn at random,
both split modes; prints the worst relative L2 error per product and every case above its bar.  python tools/fuzz_wgrad.py [cases] [seed]"""
import os, sys, random
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spoofsv_amd
from spoofsv_amd import ops
n, seed = (int(sys.argv[1]) if len(sys.argv) > 1 else 100), (int(sys.argv[2]) if len(sys.argv) > 2 else 0)
rnd = random.Random(seed)
gen = torch.Generator().manual_seed(seed)
worst, bad = {}, 0
for case in range(n):
    B, Cin, Cout = rnd.randint(1, 9), rnd.choice([8, 24, 64, 80, 128, 200, 256]), rnd.choice([16, 40, 100, 128, 130, 256, 300])
    L, d, causal, k = rnd.randint(8, 420), rnd.choice([1, 1, 3, 3, 5, 8, 9, 13, 27]), rnd.random() < 0.5, rnd.choice([3, 3, 3, 1])
    if B * L < 128:
        L = 128 // B + 8
    x = torch.randn(B, Cin, L, generator=gen); w = torch.randn(Cout, Cin, k, generator=gen) * 0.05; dy = torch.randn(B, Cout, L, generator=gen)
    pad = d * (k - 1)
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yd = F.conv1d(F.pad(xd, (pad, 0)) if causal else F.pad(xd, (pad // 2, pad // 2)), wd, None, dilation=d)
    yd.backward(dy.double())
    ref = (yd.detach(), xd.grad, wd.grad)
    for prec, tol in (("f16x2", 2e-6), ("bf16x3", 3e-5)):
        prev = spoofsv_amd.set_precision(prec)
        try:
            xg, wg = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
            y = ops.conv1d(xg, wg, None, k, d, causal); y.backward(dy.cuda()); torch.cuda.synchronize()
        finally:
            spoofsv_amd.set_precision(prev)
        for name, a, b in zip(("fwd", "dgrad", "wgrad"), (y.detach(), xg.grad, wg.grad), ref):
            e = float((a.double().cpu() - b).norm() / b.norm())
            worst[(prec, name)] = max(worst.get((prec, name), 0.0), e)
            if not (e <= tol):
                bad += 1
                print("ABOVE THE BAR: %s %s B%d Cin%d Cout%d L%d k%d d%d %s: %.3e" % (prec, name, B, Cin, Cout, L, k, d, "causal" if causal else "same", e), flush=True)
print("%d cases, %d above the bar; worst:" % (n, bad), {"%s %s" % k_: "%.2e" % v for k_, v in sorted(worst.items())})
sys.exit(1 if bad else 0)
