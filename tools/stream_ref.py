#!/usr/bin/env python3
"""Reference point (GPU box): what a trivial streaming kernel (torch.addcmul: 3 reads + 1 write per element, the byte count of
the highway gate forward) reaches at the LayerNorm kernels' tensor sizes, next to the gate forward itself -- same rotation
of operand sets, HIP events."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spoofsv_amd import _lib
P = lambda t: ctypes.c_void_p(t.data_ptr())
dev = "cuda:0"
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(fn, nset, reps=4):
    for i in range(nset): fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps * nset): fn(i % nset)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * nset) * 1e3
for (B, C, L) in [(32, 256, 325), (32, 512, 186), (32, 256, 1300), (32, 512, 1300)]:
    for nset, label in ((1, "hot (one operand set)"), (max(2, int(700e6 / (B * C * L * 16))), "cold (rotating sets > Infinity Cache)")):
        x = [torch.randn(B, C, L, device=dev) for _ in range(nset)]
        h = [torch.randn(B, 2 * C, L, device=dev) for _ in range(nset)]
        y = [torch.empty(B, C, L, device=dev) for _ in range(nset)]
        stats = torch.empty(B, 4, L, device=dev)
        g = torch.rand(C, device=dev) + 0.5; b = torch.randn(C, device=dev)
        n = B * C * L * 4
        t_ref = timeit(lambda i: torch.addcmul(x[i], h[i][:, :C], h[i][:, C:], out=y[i]), nset)
        t_gate = timeit(lambda i: _lib.call("ssv_highway_gate_fwd", P(h[i]), P(x[i]), C * L, P(g), P(b), P(g), P(b), P(stats), P(y[i]), C * L, None, B, C, L, st), nset)
        print("B%d C%d L%d %-40s addcmul %6.1f us = %.2f TB/s | highway gate fwd %6.1f us = %.2f TB/s" % (B, C, L, label, t_ref, 4 * n / t_ref / 1e6, t_gate, 4 * n / t_gate / 1e6), flush=True)
        del x, h, y
        torch.cuda.empty_cache()
