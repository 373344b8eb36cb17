#!/usr/bin/env python3
"""Diagnostic (GPU box, library built with -DSSV_NT_STAMP): where one wave of the k = 3 weight gradient's ring kernel spends a chunk.
Chunks 8 .. 15 of wave 0 of workgroup 0, in shader-clock cycles: tap 0 (MFMAs + fragment reads + the next chunk's 8 dH loads, one per group) | split +
LDS stores of block n+2 | tap 1 (+ block n+3's 4 loads) | tap 2 | in-place split of the next dH | barrier."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spoofsv_amd
from spoofsv_amd import ops, _lib
B = 32
for (C, L, k, d) in ((256, 325, 3, 3), (512, 186, 3, 3), (512, 1300, 3, 1), (256, 650, 3, 27)):
    x = torch.randn(B, C, L, device="cuda")
    w = torch.randn(2 * C, C, k, device="cuda") * 0.03
    dy = torch.randn(B, 2 * C, L, device="cuda") * 1e-4
    xa, dya = ops.amax_of(x), ops.amax_of(dy)
    for _ in range(3):
        ops._conv_bwd_weight(dy, dy.stride(0), x, x.stride(0), w.shape, k, d, 1, None, dya, xa)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 64)()
    rc = _lib.lib().ssv_debug_nt3r_stamps(buf)
    wgb = (ctypes.c_ulonglong * (4096 * 4))()
    ctypes.CDLL(_lib.LIBPATH).ssv_debug_nt3r_wg(wgb)
    a = np.frombuffer(wgb, dtype=np.uint64).reshape(4096, 4).astype(np.int64)
    a = a[a[:, 1] > 0]
    t0 = a[:, 0].min()
    ent, ext = (a[:, 0] - t0) * 0.01, (a[:, 1] - t0) * 0.01
    loop = a[:, 3] - a[:, 2]
    q = lambda v: "%.1f / %.1f / %.1f" % tuple(np.percentile(v, [10, 50, 90]))
    print("C%d L%d d%d rc=%d: %d workgroups: exit us 10/50/90 %% = %s (last %.1f) | residence us = %s | chunk loop cycles = %s" % (
        C, L, d, rc, len(a), q(ext), ext.max(), q(ext - ent), q(loop)))
    prev = None
    for s in range(8):
        t = [buf[s * 8 + i] for i in range(8)]
        print("  chunk %2d: tap0 + dH loads %5d | commit %5d | tap1 + X loads %5d | tap2 %5d | split dH %5d | barrier %5d" % (
            8 + s, t[2] - t[0], t[3] - t[2], t[5] - t[3], t[6] - t[5], t[7] - t[6], 0 if prev is None else t[0] - prev))
        prev = t[7]
