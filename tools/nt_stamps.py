#!/usr/bin/env python3
"""Diagnostic (GPU box, library built with -DSSV_NT_STAMP): where one wave of the weight-gradient kernel spends the cycles of a step.
Prints, for 8 consecutive steps of wave 0 of workgroup 0: MFMA section | commit (wait for the tile's loads, split, LDS write) | issue of the
next loads | split of the next dH chunk (last tap only) | barrier wait -- in shader-clock cycles (s_memtime, 100 MHz x ... see output)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spoofsv_amd
from spoofsv_amd import ops, _lib
B = 32
for (C, L, k, d) in ((256, 325, 3, 3), (512, 186, 3, 3), (512, 1300, 3, 1)):
    x = torch.randn(B, C, L, device="cuda")
    w = torch.randn(2 * C, C, k, device="cuda") * 0.03
    dy = torch.randn(B, 2 * C, L, device="cuda") * 1e-4
    xa, dya = ops.amax_of(x), ops.amax_of(dy)
    for _ in range(3):
        ops._conv_bwd_weight(dy, dy.stride(0), x, x.stride(0), w.shape, k, d, 1, None, dya, xa)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 64)()
    rc = _lib.lib().ssv_debug_nt_stamps(buf)
    print("C%d L%d k%d rc=%d" % (C, L, k, rc))
    import numpy as np
    wgb = (ctypes.c_ulonglong * (4096 * 4))()
    ctypes.CDLL(_lib.LIBPATH).ssv_debug_nt_wg(wgb)
    a = np.frombuffer(wgb, dtype=np.uint64).reshape(4096, 4).astype(np.int64)
    a = a[a[:, 1] > 0]
    t0 = a[:, 0].min()
    ent, ext = (a[:, 0] - t0) * 0.01, (a[:, 1] - t0) * 0.01
    loop = a[:, 3] - a[:, 2]
    q = lambda v: "%.1f / %.1f / %.1f" % tuple(np.percentile(v, [10, 50, 90]))
    print("  %d workgroups: entry us 10/50/90 %% = %s (last %.1f) | exit us = %s (last %.1f) | residence us = %s | entry -> end of chunk loop, cycles = %s" % (
        len(a), q(ent), ent.max(), q(ext), ext.max(), q(ext - ent), q(loop)))
    prev_end = None
    for s in range(8):
        t = [buf[s * 8 + i] for i in range(7)]
        gap = (t[0] - prev_end) if prev_end else 0
        print("  step %2d: mfma %5d | wait for tile %5d | split + LDS write %5d | issue loads %4d | split dH %5d | barrier %5d | total %5d (+%d to next)" % (
            24 + s, t[1] - t[0], t[6] - t[1], t[2] - t[6], t[3] - t[2], t[4] - t[3], t[5] - t[4], t[5] - t[0], gap))
        prev_end = t[5]
