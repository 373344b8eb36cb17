#!/usr/bin/env python3
"""Tuning aid: time the channel-LayerNorm entry points (ln_act and the highway gate) at the hot shapes for each group count
(SSV_LN_GROUPS) and report effective HBM bandwidth."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spoofsv_amd import _lib
P = lambda t: ctypes.c_void_p(t.data_ptr())
dev = "cuda:0"
def timeit(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (B, C, L) in [(32, 512, 186), (32, 256, 325), (32, 256, 650), (32, 256, 1300), (32, 512, 1300), (32, 513, 1300), (32, 80, 325)]:
    x = torch.randn(B, C, L, device=dev); y = torch.empty_like(x); dy = torch.randn_like(x); dx = torch.empty_like(x)
    h = torch.randn(B, 2 * C, L, device=dev); dh = torch.empty_like(h)
    g = torch.rand(C, device=dev) + 0.5; b = torch.randn(C, device=dev)
    stats = torch.empty(B, 4, L, device=dev); pg = torch.empty(6, C, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    nb = _lib.query("ssv_channel_ln_act_bwd_workspace", B, C, L); ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    ng = _lib.query("ssv_highway_gate_bwd_workspace", B, C, L) if C <= 512 else 0; wg = torch.empty(max(ng, 256), dtype=torch.uint8, device=dev)
    n = B * C * L * 4
    line = "B%d C%d L%d:" % (B, C, L)
    for G in os.environ.get("BENCH_LN_GROUPS", "16,32,64").split(","):      # "auto" = the library's own choice
        if G == "auto": os.environ.pop("SSV_LN_GROUPS", None)
        else: os.environ["SSV_LN_GROUPS"] = G
        ctypes.CDLL(_lib.LIBPATH).ssv_reload_tuning()        # exported for tuning scripts, not part of include/ssv_hip.h
        f = lambda: _lib.call("ssv_channel_ln_act_fwd", P(x), C * L, P(g), P(b), P(y), C * L, None, P(stats), B, C, L, 1, None, 0, st)
        bw = lambda: _lib.call("ssv_channel_ln_act_bwd", P(dy), C * L, P(x), C * L, P(stats), P(g), P(b), P(dx), C * L, P(pg), B, C, L, 1, P(ws), nb, st)
        tf, tb = timeit(f), timeit(bw)
        line += "  G%s act f %.1f (%.2f) b %.1f (%.2f)" % (G, tf, 2 * n / tf / 1e6, tb, 3 * n / tb / 1e6)
        if C <= 512:
            gb = lambda: _lib.call("ssv_highway_gate_bwd", P(dy), C * L, P(x), C * L, P(g), P(b), P(g), P(b), P(h), P(stats), P(dh), P(dx), C * L, P(pg), B, C, L, P(wg), ng, st)
            gf = lambda: _lib.call("ssv_highway_gate_fwd", P(h), P(x), C * L, P(g), P(b), P(g), P(b), P(stats), P(y), C * L, None, B, C, L, st)
            tg, tgf = timeit(gb), timeit(gf)
            line += " gate-f %.1f (%.2f) gate-b %.1f (%.2f)" % (tgf, 4 * n / tgf / 1e6, tg, 7 * n / tg / 1e6)
    print(line, flush=True)
