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
    rt = (ctypes.c_ulonglong * 4096)()
    _lib.lib().ssv_debug_nn_realtime(rt)
    nwg = min(2048, B * ((Cout + 63) // 64) * ((L + 111) // 112))
    ent = sorted(rt[2 * i] for i in range(nwg) if rt[2 * i]); ext = sorted(rt[2 * i + 1] for i in range(nwg) if rt[2 * i + 1])
    if ent and ext:
        t0_ = ent[0]
        q = lambda v, f: (v[int(f * (len(v) - 1))] - t0_) / 100.0
        print("  s_memrealtime, us after the first entry (%d workgroups): entries median %.1f, 90 %% %.1f, last %.1f | exits first %.1f, median %.1f, last %.1f" % (
            len(ent), q(ent, .5), q(ent, .9), q(ent, 1), q(ext, 0), q(ext, .5), q(ext, 1)))
        import statistics  # noqa
        for lo_ in range(0, nwg, 256):
            seg = [(rt[2 * i + 1] - t0_) / 100.0 for i in range(lo_, min(lo_ + 256, nwg)) if rt[2 * i + 1]]
            dur = [(rt[2 * i + 1] - rt[2 * i]) / 100.0 for i in range(lo_, min(lo_ + 256, nwg)) if rt[2 * i + 1]]
            print("    workgroups %4d-%4d: exit median %.1f us (min %.1f, max %.1f), duration median %.1f" % (lo_, lo_ + len(seg) - 1, statistics.median(seg), min(seg), max(seg), statistics.median(dur)))
    a = [buf[120 + i] for i in range(4)]
    print("  workgroup %s: entry -> first chunk ready %d | chunk loop %d | epilogue (stores issued) %d cycles; entry %d cycles after workgroup (0, 0)'s" % (
        os.environ.get("NN_WG", "0"), a[1] - a[0], a[2] - a[1], a[3] - a[2], a[0] - buf[127]))
