#!/usr/bin/env python3
"""Timing of the k = 3 weight gradient on three step shapes (GPU box), for ablation / variant builds (SSV_HIP_LIB): us per call, operands re-used."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spoofsv_amd
from spoofsv_amd import ops
B = 32
out = []
for (C, L, d) in ((256, 325, 3), (512, 186, 3), (512, 1300, 1)):
    x = torch.randn(B, C, L, device="cuda")
    dy = torch.randn(B, 2 * C, L, device="cuda") * 1e-4
    xa, dya = ops.amax_of(x), ops.amax_of(dy)
    run = lambda: ops._conv_bwd_weight(dy, dy.stride(0), x, x.stride(0), (2 * C, C, 3), 3, d, 1, None, dya, xa)
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    out.append("C%d L%d: %.1f us" % (C, L, e0.elapsed_time(e1) * 50))
print("%-10s %s" % (os.environ.get("AB_TAG", "default"), " | ".join(out)), flush=True)
