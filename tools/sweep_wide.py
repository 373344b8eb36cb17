#!/usr/bin/env python3
"""Tuning aid: time ssv_conv1d_fwd for the k=1 shapes over the wide-workgroup kernel variants (SSV_NNB_WIDE=wm,nt,nwn)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spoofsv_amd import _lib
SHAPES = [(32, 513, 513, 1300, 1), (32, 512, 513, 1300, 1), (32, 513, 512, 1300, 1), (32, 256, 512, 1300, 1), (32, 512, 512, 186, 1), (32, 256, 256, 650, 1), (32, 256, 256, 325, 1), (32, 80, 256, 325, 1), (32, 256, 80, 325, 1), (32, 512, 256, 325, 1)]
P = lambda t: ctypes.c_void_p(t.data_ptr())
dev = "cuda:0"
for (B, Cin, Cout, L, k) in SHAPES:
    x = torch.randn(B, Cin, L, device=dev); w = torch.randn(Cout, Cin, k, device=dev) * 0.05; y = torch.empty(B, Cout, L, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    nb = _lib.query("ssv_conv1d_fwd_workspace", Cin, Cout, k); ws = torch.empty(max(nb, 256), dtype=torch.uint8, device=dev)
    flops = 2.0 * B * L * Cout * Cin * k
    res = []
    for cfg in ["auto", "w2,7,3", "w1,7,3", "w2,7,2", "w2,6,2", "w1,6,2", "w2,4,4", "w2,7,4", "t2,7", "t2,6", "t2,4", "t2,2", "t1,7", "t1,4"]:
        os.environ.pop("SSV_NNB_WIDE", None); os.environ.pop("SSV_NNB_TILE", None)
        if cfg[0] == "w": os.environ["SSV_NNB_WIDE"] = cfg[1:]
        if cfg[0] == "t": os.environ["SSV_NNB_TILE"] = cfg[1:]; os.environ["SSV_NNB_WIDE"] = "0,0,0"
        _lib.lib().ssv_reload_tuning()
        run = lambda: _lib.call("ssv_conv1d_fwd", P(x), Cin * L, P(w), None, None, None, P(y), Cout * L, B, Cin, Cout, L, k, 1, 0, P(ws), nb, st)
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        res.append((cfg, ms * 1e3, flops / ms / 1e9))
    print("B%d Cin%d Cout%d L%d k%d: " % (B, Cin, Cout, L, k) + " ".join("%s:%.0fus/%.0f" % r for r in res), flush=True)
