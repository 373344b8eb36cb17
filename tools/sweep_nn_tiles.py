#!/usr/bin/env python3
"""Tuning aid: time ssv_conv1d_fwd over the (WM, NT) tile choices for the hot launch shapes (GPU box)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spoofsv_amd import _lib

SHAPES = [  # B, Cin, Cout, L, k
    (32, 256, 512, 325, 3), (32, 512, 1024, 186, 3), (32, 256, 512, 650, 3), (32, 256, 512, 1300, 3),
    (32, 512, 1024, 1300, 3), (32, 512, 256, 325, 3), (32, 1024, 512, 186, 3), (32, 1024, 512, 1300, 3),
    (32, 513, 513, 1300, 1), (32, 256, 256, 325, 1), (32, 512, 513, 1300, 1), (1, 768, 3072, 880, 1),
]
P = lambda t: ctypes.c_void_p(t.data_ptr())
dev = "cuda:0"
_lib.lib().ssv_set_precision(0 if os.environ.get("SWEEP_FP32") else 1)
for (B, Cin, Cout, L, k) in SHAPES:
    x = torch.randn(B, Cin, L, device=dev); w = torch.randn(Cout, Cin, k, device=dev) * 0.05
    y = torch.empty(B, Cout, L, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    nb = _lib.query("ssv_conv1d_fwd_workspace", Cin, Cout, k)
    ws = torch.empty(max(nb, 256), dtype=torch.uint8, device=dev)
    flops = 2.0 * B * L * Cout * Cin * k
    res = []
    for cfg in ["auto"] + ["%d,%d" % (wm, nt) for wm in (1, 2) for nt in (2, 4, 6, 7, 8)]:
        if cfg == "auto": os.environ.pop("SSV_NN_TILE", None); os.environ.pop("SSV_NNB_TILE", None)
        else: os.environ["SSV_NN_TILE"] = cfg; os.environ["SSV_NNB_TILE"] = cfg
        _lib.lib().ssv_reload_tuning()
        run = lambda: _lib.call("ssv_conv1d_fwd", P(x), Cin * L, P(w), None, None, None, P(y), Cout * L, B, Cin, Cout, L, k, 1, 0, P(ws), nb, st)
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        res.append((cfg, ms, flops / ms / 1e9))
    best = min(res[1:], key=lambda r: r[1])
    print("B%d Cin%d Cout%d L%d k%d: auto %.1fus %.1fTF | best %s %.1fus %.1fTF | " % (B, Cin, Cout, L, k, res[0][1]*1e3, res[0][2], best[0], best[1]*1e3, best[2])
          + " ".join("%s:%.0f" % (c, tf) for c, _, tf in res[1:]), flush=True)
