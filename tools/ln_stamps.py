#!/usr/bin/env python3
"""Diagnostic (GPU box, library built with -DSSV_LN_STAMP via tools/build_variant.sh): where the workgroups of the highway LayerNorm / gate
BACKWARD kernel spend their time.  Per shape: when the workgroups enter and leave (s_memrealtime, 10 ns ticks, one clock for the device) and
the shader-clock length of a workgroup's three phases -- loads + first pass | cross-group sums (LDS, two barriers) | second pass + stores
issued | stores acknowledged."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spoofsv_amd import _lib
P = lambda t: ctypes.c_void_p(t.data_ptr())
dev = "cuda:0"
raw = ctypes.CDLL(_lib.LIBPATH)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for (B, C, L) in [(32, 256, 325), (32, 512, 186), (32, 256, 1300), (32, 512, 1300)]:
    nset = max(2, int(600e6 / (B * C * L * 28)))
    sets = []
    for _ in range(nset):
        sets.append(dict(dy=torch.randn(B, C, L, device=dev), x=torch.randn(B, C, L, device=dev), h=torch.randn(B, 2 * C, L, device=dev),
                         dh=torch.empty(B, 2 * C, L, device=dev), dx=torch.empty(B, C, L, device=dev)))
    g = torch.rand(C, device=dev) + 0.5; b = torch.randn(C, device=dev)
    stats = torch.rand(B, 4, L, device=dev) + 0.5; pg = torch.empty(6, C, device=dev)
    ng = _lib.query("ssv_highway_gate_bwd_workspace", B, C, L); wg = torch.empty(max(ng, 256), dtype=torch.uint8, device=dev)
    def run(s):
        _lib.call("ssv_highway_gate_bwd", P(s["dy"]), C * L, P(s["x"]), C * L, P(g), P(b), P(g), P(b), P(s["h"]), P(stats), P(s["dh"]), P(s["dx"]), C * L, P(pg), B, C, L, P(wg), ng, st)
    for s in sets: run(s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for s in sets: run(s)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / nset * 1e3
    buf = (ctypes.c_ulonglong * (4096 * 8))()
    rc = raw.ssv_debug_ln_stamps(buf)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8).astype(np.int64)
    n = min(4096, B * ((L + 15) // 16))
    a = a[:n]
    t_in = (a[:, 0] - a[:, 0].min()) * 0.01            # us
    t_out = (a[:, 6] - a[:, 0].min()) * 0.01
    ph = np.stack([a[:, 2] - a[:, 1], a[:, 3] - a[:, 2], a[:, 4] - a[:, 3], a[:, 5] - a[:, 4]], 1)
    q = lambda v: "%7.0f / %7.0f / %7.0f" % tuple(np.percentile(v, [10, 50, 90]))
    print("B%d C%d L%d: %d workgroups (%d stamped), %.1f us per launch (cold operands), rc=%d" % (B, C, L, B * ((L + 15) // 16), n, us, rc))
    print("  entry (us after the first): 10/50/90 %% = %s, last %.2f" % (q(t_in), t_in.max()))
    print("  exit  (us after the first entry): 10/50/90 %% = %s, last %.2f" % (q(t_out), t_out.max()))
    print("  residence (us): %s" % q(t_out - t_in))
    for name, col in zip(("loads + first pass", "cross-group sums", "second pass + stores issued", "stores acknowledged"), range(4)):
        print("  %-28s cycles 10/50/90 %% = %s" % (name, q(ph[:, col])))
    del sets
    torch.cuda.empty_cache()
