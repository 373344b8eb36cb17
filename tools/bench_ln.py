#!/usr/bin/env python3
"""Tuning aid: time the channel-LayerNorm entry points at the hot shapes and report effective HBM bandwidth."""
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
for (B, C, L) in [(32, 513, 1300), (32, 512, 186), (32, 256, 325), (32, 256, 1300), (32, 80, 325)]:
    x = torch.randn(B, C, L, device=dev); y = torch.empty_like(x); dy = torch.randn_like(x); dx = torch.empty_like(x)
    g = torch.rand(C, device=dev) + 0.5; b = torch.randn(C, device=dev)
    stats = torch.empty(B, 2, L, device=dev); pg = torch.empty(3, C, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    nb = _lib.query("ssv_channel_ln_act_bwd_workspace", B, C, L); ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    nbf = _lib.query("ssv_channel_ln_act_fwd_workspace", B, C, L); wsf = torch.empty(nbf, dtype=torch.uint8, device=dev)
    f = lambda: _lib.call("ssv_channel_ln_act_fwd", P(x), C * L, P(g), P(b), P(y), C * L, P(stats), B, C, L, 1, P(wsf), nbf, st)
    bw = lambda: _lib.call("ssv_channel_ln_act_bwd", P(dy), C * L, P(x), C * L, P(stats), P(g), P(b), P(dx), C * L, P(pg), B, C, L, 1, P(ws), nb, st)
    tf, tb = timeit(f), timeit(bw)
    n = B * C * L * 4
    print("B%d C%d L%d: fwd %.1fus %.2f TB/s | bwd(+reduce) %.1fus %.2f TB/s" % (B, C, L, tf, 2 * n / tf / 1e6, tb, 3 * n / tb / 1e6), flush=True)
