#!/usr/bin/env python3
"""Tuning aid (GPU box): conv forward only, a few shapes, for the library named by SSV_HIP_LIB (cold operands, resident weights)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spoofsv_amd import _lib, resident
P = lambda t: ctypes.c_void_p(t.data_ptr())
dev = "cuda:0"
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
out = []
for (B, C, L, k) in [(32, 256, 325, 3), (32, 512, 186, 3), (32, 256, 1300, 3), (32, 512, 1300, 3)]:
    nset = 12 if L <= 400 else 3
    xs = [torch.randn(B, C, L, device=dev) for _ in range(nset)]
    hs = [torch.empty(B, 2 * C, L, device=dev) for _ in range(nset)]
    w = torch.randn(2 * C, C, k, device=dev) * 0.03
    bias = torch.randn(2 * C, device=dev)
    rw = resident.ResidentWeights([w]); rw.refresh(st); wp = resident.lookup(w)
    nb = _lib.query("ssv_conv1d_fwd_workspace", C, 2 * C, k); ws = torch.empty(max(nb, 256), dtype=torch.uint8, device=dev)
    run = lambda i: _lib.call("ssv_conv1d_fwd", P(xs[i]), C * L, P(w), wp, P(bias), None, P(hs[i]), 2 * C * L, B, C, 2 * C, L, k, 1, 1, P(ws), nb, st)
    for i in range(nset): run(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(4 * nset): run(i % nset)
    e1.record(); torch.cuda.synchronize()
    out.append("C%d L%d %.1fus" % (C, L, e0.elapsed_time(e1) / (4 * nset) * 1e3))
    resident.invalidate([w])
print("%-10s" % os.path.basename(os.environ.get("SSV_HIP_LIB", "in-tree")), "  ".join(out), flush=True)
