#!/usr/bin/env python3
"""Diagnostic (GPU box, library built with -DSSV_NN_STAMP): where one wave of the forward conv kernel spends the cycles of a K chunk:
taps (MFMAs + weight-fragment re-loads) | commit of the next chunk's input tile (wait, split, LDS write) | issue of the chunk after's loads | barrier."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spoofsv_amd
from spoofsv_amd import ops, _lib
B = 32
for (Cin, Cout, L, k, d) in ((256, 512, 325, 3, 3), (512, 1024, 186, 3, 3), (512, 1024, 1300, 3, 1)):
    x = torch.randn(B, Cin, L, device="cuda")
    w = torch.randn(Cout, Cin, k, device="cuda") * 0.03
    for _ in range(3):
        y = ops.conv1d(x, w, None, k, d, True)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 128)()
    rc = _lib.lib().ssv_debug_nn_stamps(buf)
    print("C%d->%d L%d k%d rc=%d" % (Cin, Cout, L, k, rc))
    prev = None
    for ch in range(min(Cin // 32, 16)):
        t = [buf[ch * 8 + i] for i in range(5)]
        if t[1] == 0: continue
        print("  chunk %2d: taps %5d | commit %5d | issue loads %5d | barrier %5d | total %5d (+%d)" % (
            ch, t[1] - t[0], t[2] - t[1] if t[2] else 0, (t[3] - t[2]) if t[2] else t[3] - t[1], t[4] - t[3], t[4] - t[0], (t[0] - prev) if prev else 0))
        prev = t[4]
    a = [buf[120 + i] for i in range(4)]
    print("  workgroup %s: entry -> first chunk ready %d | chunk loop %d | epilogue (stores issued) %d cycles; entry %d cycles after workgroup (0, 0)'s" % (
        os.environ.get("NN_WG", "0"), a[1] - a[0], a[2] - a[1], a[3] - a[2], a[0] - buf[127]))
