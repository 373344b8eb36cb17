#!/usr/bin/env python3
"""Tuning aid: time ssv_conv1d_bwd_weight (split-K weight gradient + slab reduction) at the hot launch shapes (GPU box)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spoofsv_amd import _lib

SHAPES = [  # B, Cin, Cout, L, k
    (32, 256, 512, 325, 3), (32, 512, 1024, 186, 3), (32, 256, 512, 650, 3), (32, 256, 512, 1300, 3),
    (32, 512, 1024, 1300, 3), (32, 513, 513, 1300, 1), (32, 256, 256, 325, 1), (32, 512, 512, 186, 1), (32, 512, 513, 1300, 1),
]
P = lambda t: ctypes.c_void_p(t.data_ptr())
dev = "cuda:0"
def cdiv(a, b): return (a + b - 1) // b
for (B, Cin, Cout, L, k) in SHAPES:
    x = torch.randn(B, Cin, L, device=dev); dy = torch.randn(B, Cout, L, device=dev)
    dw = torch.empty(Cout, Cin, k, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    flops = 2.0 * B * L * Cout * Cin * k
    res = []
    cfgs = [("auto", None, None)]
    for wm in (1, 2):
        for ntc in ((2, 4) if k == 3 else (2, 4, 6)):
            tiles = cdiv(Cout, 64 * wm) * cdiv(Cin, 16 * ntc)
            for target in (256, 512, 1024):
                z = max(1, min(B, cdiv(target, tiles)))
                cfgs.append(("%d,%d,z%d" % (wm, ntc, z), "%d,%d" % (wm, ntc), str(z)))
    seen = set()
    for name, plan, z in cfgs:
        if name in seen: continue
        seen.add(name)
        for kk, v in (("SSV_NT_PLAN", plan), ("SSV_NT_Z", z)):
            if v is None: os.environ.pop(kk, None)
            else: os.environ[kk] = v
        _lib.lib().ssv_reload_tuning()            # the library reads its knobs once; tell it they changed
        nb = _lib.query("ssv_conv1d_bwd_weight_workspace", B, Cin, Cout, k)
        ws = torch.empty(max(nb, 256), dtype=torch.uint8, device=dev)
        run = lambda: _lib.call("ssv_conv1d_bwd_weight", P(dy), Cout * L, P(x), Cin * L, P(dw), B, Cin, Cout, L, k, 1, 0, P(ws), nb, st)
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        res.append((name, ms * 1e3, flops / ms / 1e9))
    best = min(res[1:], key=lambda r: r[1])
    print("B%d Cin%d Cout%d L%d k%d: auto %.1fus %.0fTF | best %s %.1fus %.0fTF | " % (B, Cin, Cout, L, k, res[0][1], res[0][2], best[0], best[1], best[2])
          + " ".join("%s:%.0f" % (n, tf) for n, _, tf in res[1:]), flush=True)
