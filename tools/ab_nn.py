#!/usr/bin/env python3
"""Timing of the k = 3 conv forward on three step shapes (GPU box), for ablation / variant builds (SSV_HIP_LIB): us per call, resident weights."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spoofsv_amd
from spoofsv_amd import ops, _lib, resident
B = 32
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
out = []
for (C, L, d) in ((256, 325, 1), (512, 186, 3), (512, 1300, 1), (256, 325, 27)):
    nset = 6
    xs = [torch.randn(B, C, L, device="cuda") for _ in range(nset)]
    ys = [torch.empty(B, 2 * C, L, device="cuda") for _ in range(nset)]
    xa = [ops.amax_of(t) for t in xs]
    w = torch.randn(2 * C, C, 3, device="cuda") * 0.05
    bias = torch.randn(2 * C, device="cuda")
    rw = resident.ResidentWeights([w]); rw.refresh(st); wp = resident.lookup(w)
    nb = _lib.query("ssv_conv1d_fwd_workspace", C, 2 * C, 3)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    cs = torch.empty(B * (2 * C // 64) * L * 2, device="cuda")
    run = lambda i: _lib.call("ssv_conv1d_fwd", P(xs[i]), C * L, P(xa[i]), ops._AMAX_PIECES, P(w), wp, P(bias), None, P(ys[i]), 2 * C * L, P(cs), B, C, 2 * C, L, 3, d, 1, P(ws), nb, st)
    for i in range(nset): run(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(5 * nset): run(i % nset)
    e1.record(); torch.cuda.synchronize()
    out.append("C%d L%d d%d: %.1f us" % (C, L, d, e0.elapsed_time(e1) * 1000 / (5 * nset)))
    resident.invalidate([w])
print("%-10s %s" % (os.environ.get("AB_TAG", "default"), " | ".join(out)), flush=True)
