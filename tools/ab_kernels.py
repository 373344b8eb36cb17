#!/usr/bin/env python3
"""Tuning aid (GPU box): time the hot operators of the training step for the library named by SSV_HIP_LIB (default: the
in-tree build), cold operands (rotating tensor sets larger than the Infinity Cache), HIP events.  One line of microseconds
per operator and shape, so two builds can be compared in one gpurun call:

    SSV_HIP_LIB=spoofsv_amd/csrc/build/ab/base.so python tools/ab_kernels.py; python tools/ab_kernels.py
"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spoofsv_amd import _lib, resident
P = lambda t: ctypes.c_void_p(t.data_ptr())
dev = "cuda:0"
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(run, nset, reps=3):
    for i in range(nset): run(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps * nset): run(i % nset)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * nset) * 1e3


tag = os.path.basename(os.environ.get("SSV_HIP_LIB", "in-tree"))
only = os.environ.get("AB_ONLY", "")
for (B, C, L, k, d) in [(32, 256, 325, 3, 1), (32, 256, 325, 3, 27), (32, 512, 186, 3, 1), (32, 256, 1300, 3, 3), (32, 512, 1300, 3, 1)]:
    nset = 12 if L <= 400 else 3
    xs = [torch.randn(B, C, L, device=dev) for _ in range(nset)]
    hs = [torch.randn(B, 2 * C, L, device=dev) for _ in range(nset)]
    ys = [torch.empty(B, C, L, device=dev) for _ in range(nset)]
    dxs = [torch.empty(B, C, L, device=dev) for _ in range(nset)]
    w = torch.randn(2 * C, C, k, device=dev) * 0.03
    bias = torch.randn(2 * C, device=dev)
    g = [torch.rand(C, device=dev) + 0.5 for _ in range(2)]
    bb = [torch.randn(C, device=dev) for _ in range(2)]
    stats = torch.empty(B, 4, L, device=dev)
    dw = torch.empty_like(w); pg = torch.empty(6, C, device=dev)
    rw = resident.ResidentWeights([w]); rw.refresh(st); wp = resident.lookup(w)
    res = {}
    nb = _lib.query("ssv_conv1d_fwd_workspace", C, 2 * C, k); ws = torch.empty(max(nb, 256), dtype=torch.uint8, device=dev)
    res["conv_fwd"] = timeit(lambda i: _lib.call("ssv_conv1d_fwd", P(xs[i]), C * L, P(w), wp, P(bias), None, P(hs[i]), 2 * C * L, B, C, 2 * C, L, k, d, 1, P(ws), nb, st), nset)
    nb2 = _lib.query("ssv_conv1d_bwd_data_workspace", C, 2 * C, k); ws2 = torch.empty(max(nb2, 256), dtype=torch.uint8, device=dev)
    res["conv_dgrad"] = timeit(lambda i: _lib.call("ssv_conv1d_bwd_data", P(hs[i]), 2 * C * L, P(w), wp, None, P(dxs[i]), C * L, B, C, 2 * C, L, k, d, 1, P(ws2), nb2, st), nset)
    nb3 = _lib.query("ssv_conv1d_bwd_weight_workspace", B, C, 2 * C, k); ws3 = torch.empty(max(nb3, 256), dtype=torch.uint8, device=dev)
    res["conv_wgrad"] = timeit(lambda i: _lib.call("ssv_conv1d_bwd_weight", P(hs[i]), 2 * C * L, P(xs[i]), C * L, P(dw), B, C, 2 * C, L, k, d, 1, P(ws3), nb3, st), nset)
    res["gate_fwd"] = timeit(lambda i: _lib.call("ssv_highway_gate_fwd", P(hs[i]), P(xs[i]), C * L, P(g[0]), P(bb[0]), P(g[1]), P(bb[1]), P(stats), P(ys[i]), C * L, B, C, L, st), nset)
    nb4 = _lib.query("ssv_highway_conv1d_bwd_workspace", B, C, L, k); ws4 = torch.empty(max(nb4, 256), dtype=torch.uint8, device=dev)
    res["highway_bwd"] = timeit(lambda i: _lib.call("ssv_highway_conv1d_bwd", P(ys[i]), C * L, P(xs[i]), C * L, P(w), wp, P(g[0]), P(bb[0]), P(g[1]), P(bb[1]), P(hs[i]), P(stats),
                                                    P(dxs[i]), C * L, P(dw), P(pg), B, C, L, k, d, 1, P(ws4), nb4, st), nset)
    nb5 = _lib.query("ssv_highway_conv1d_fwd_workspace", B, C, L, k); ws5 = torch.empty(max(nb5, 256), dtype=torch.uint8, device=dev)
    hh = torch.empty(B, 2 * C, L, device=dev)
    res["highway_fwd"] = timeit(lambda i: _lib.call("ssv_highway_conv1d_fwd", P(xs[i]), C * L, P(w), wp, P(bias), P(g[0]), P(bb[0]), P(g[1]), P(bb[1]), P(hh), P(stats), P(ys[i]), C * L,
                                                    B, C, L, k, d, 1, P(ws5), nb5, st), nset)
    gf = 2.0 * B * L * 2 * C * C * k / 1e6
    print("%-10s B%d C%d L%d k%d d%-2d  " % (tag, B, C, L, k, d) + "  ".join("%s %.1f" % (n, v) for n, v in res.items()) +
          "   [fwd %.0f, wgrad %.0f TFLOP/s]" % (gf / res["conv_fwd"], gf / res["conv_wgrad"]), flush=True)
    resident.invalidate([w])
    del xs, hs, ys, dxs
    torch.cuda.empty_cache()
