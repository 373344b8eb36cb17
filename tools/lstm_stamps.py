#!/usr/bin/env python3
"""Diagnostic (GPU box, library built with -DSSV_NN_STAMP): the LSTM wavefront step of the GE2E embedder (BASELINE config 5: 880 utterances x 120
frames, hidden 768), as seen by wave 0 of workgroup (0, 0) of the LAST launch (layer 2 alone, frame 119: 3072 x 880 x 1536, 128 x 64 tiles):
shader-clock cycles of a PAIR of K chunks -- MFMAs of chunk c | split + LDS write of chunk c + 1 | issue of the loads of chunk c + 2 | barrier |
the second chunk of the pair (MFMAs, commit, loads, barrier) -- and when the launch's workgroups enter and leave (s_memrealtime)."""
import ctypes, os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spoofsv_amd
from spoofsv_amd import _lib
from spoofsv_amd.ge2e import SpeechEmbedder
raw = ctypes.CDLL(_lib.LIBPATH)
for prec in ("f16x2", "bf16x3"):
    spoofsv_amd.set_precision(prec)
    torch.manual_seed(0)
    m = SpeechEmbedder().to("cuda:0").eval()
    x = torch.randn(880, 120, 40, device="cuda:0")
    with torch.no_grad():
        for _ in range(2): m(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); m(x); e1.record(); torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 128)()
    raw.ssv_debug_nn_stamps(buf)
    print("%s: embedder forward %.2f ms (stamp build)" % (prec, e0.elapsed_time(e1)))
    prev = None
    for ch in range(0, 15, 2):
        t = [buf[ch * 8 + i] for i in range(6)]
        if not t[5]: continue
        print("  chunks %2d,%2d: mfma %4d | commit %4d | issue loads %4d | barrier %4d | second chunk %5d | pair %5d (+%d)" % (
            ch, ch + 1, t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], t[5] - t[0], (t[0] - prev) if prev else 0))
        prev = t[5]
    a = [buf[120 + i] for i in range(4)]
    print("  workgroup (0, 0): kernel entry -> prologue start %d | chunk loop + prologue %d | (epilogue not stamped for the cell epilogue)" % (a[1] - a[0], a[2] - a[1]))
    rt = (ctypes.c_ulonglong * 4096)()
    raw.ssv_debug_nn_realtime(rt)
    nwg = 24 * 14
    ent = sorted(rt[2 * i] for i in range(nwg) if rt[2 * i])
    if ent:
        t0_ = ent[0]
        print("  (exit stamps are not taken in the cell epilogue; %d entries within %.2f us)" % (len(ent), (ent[-1] - t0_) / 100.0))
