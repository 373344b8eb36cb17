#!/usr/bin/env python3
"""Tuning aid: time ssv_conv1d_fwd (resident weights) for kernel-size-3 shapes over 4-wave tiles (t) and wide workgroups (w)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spoofsv_amd import _lib, resident
SHAPES = [(32, 256, 512, 325, 3, 1), (32, 256, 512, 325, 3, 27), (32, 256, 256, 325, 3, 1), (32, 512, 1024, 186, 3, 3), (32, 256, 512, 1300, 3, 1), (32, 512, 256, 325, 3, 1)]
P = lambda t: ctypes.c_void_p(t.data_ptr())
dev = "cuda:0"
for (B, Cin, Cout, L, k, d) in SHAPES:
    xs = [torch.randn(B, Cin, L, device=dev) for _ in range(8)]
    w = torch.randn(Cout, Cin, k, device=dev) * 0.05
    ys = [torch.empty(B, Cout, L, device=dev) for _ in range(8)]
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rw = resident.ResidentWeights([w]); rw.refresh(st)
    nb = _lib.query("ssv_conv1d_fwd_workspace", Cin, Cout, k); ws = torch.empty(max(nb, 256), dtype=torch.uint8, device=dev)
    flops = 2.0 * B * L * Cout * Cin * k
    res = []
    for cfg in ["auto", "w2,6,2", "w2,7,2", "w1,6,2", "w2,4,4", "w2,7,4", "w2,7,3", "t2,7", "t2,6", "t2,4", "t1,7", "t1,6", "t1,4"]:
        os.environ.pop("SSV_NNB_WIDE", None); os.environ.pop("SSV_NNB_TILE", None)
        if cfg[0] == "w": os.environ["SSV_NNB_WIDE"] = cfg[1:]
        if cfg[0] == "t": os.environ["SSV_NNB_TILE"] = cfg[1:]; os.environ["SSV_NNB_WIDE"] = "0,0,0"
        _lib.lib().ssv_reload_tuning()
        run = lambda i: _lib.call("ssv_conv1d_fwd", P(xs[i % 8]), Cin * L, P(w), resident.lookup(w), None, None, P(ys[i % 8]), Cout * L, B, Cin, Cout, L, k, d, 0, P(ws), nb, st)
        for i in range(3): run(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(16): run(i)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 16
        res.append((cfg, ms * 1e3, flops / ms / 1e9))
    print("B%d Cin%d Cout%d L%d k%d d%d: " % (B, Cin, Cout, L, k, d) + " ".join("%s:%.0fus/%.0f" % r for r in res), flush=True)
