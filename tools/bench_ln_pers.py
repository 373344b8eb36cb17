#!/usr/bin/env python3
"""Tuning aid (GPU box): the highway LayerNorm / gate backward on cold operands (rotating tensor sets larger than the Infinity Cache) at the
step's C = 256 shapes, tile kernels (SSV_LN_PERSIST=0) against the persistent kernel with the given slot counts."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spoofsv_amd import _lib
P = lambda t: ctypes.c_void_p(t.data_ptr())
dev = "cuda:0"
raw = ctypes.CDLL(_lib.LIBPATH)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
modes = sys.argv[1:] or ["0", "256"]
for (B, C, L) in [(32, 256, 325), (32, 256, 650), (32, 256, 1300), (8, 256, 325)]:
    nset = max(3, int(700e6 / (B * C * L * 28)))
    sets = [dict(dy=torch.randn(B, C, L, device=dev), x=torch.randn(B, C, L, device=dev), h=torch.randn(B, 2 * C, L, device=dev),
                 dh=torch.empty(B, 2 * C, L, device=dev), dx=torch.empty(B, C, L, device=dev)) for _ in range(nset)]
    g = torch.rand(C, device=dev) + 0.5; b = torch.randn(C, device=dev)
    stats = torch.rand(B, 4, L, device=dev) + 0.5; pg = torch.empty(6, C, device=dev)
    ng = _lib.query("ssv_highway_gate_bwd_workspace", B, C, L); wg = torch.empty(max(ng, 256), dtype=torch.uint8, device=dev)
    amax = torch.empty(B * 4 * ((L + 63) // 64), device=dev)
    line = "B%d C%d L%d (%.1f MB):" % (B, C, L, 28.0 * B * C * L / 1e6)
    for m in modes:
        os.environ["SSV_LN_PERSIST"] = m
        raw.ssv_reload_tuning()
        def run(s):
            _lib.call("ssv_highway_gate_bwd", P(s["dy"]), C * L, P(s["x"]), C * L, P(g), P(b), P(g), P(b), P(s["h"]), P(stats), P(s["dh"]), P(s["dx"]), C * L, None, B, C, L, P(wg), ng, st)
        for s in sets: run(s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            for s in sets: run(s)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / (3 * nset) * 1e3
        line += "  [%s] %.1f us %.2f TB/s" % (m, us, 28.0 * B * C * L / us / 1e6)
    print(line, flush=True)
    del sets
    torch.cuda.empty_cache()
