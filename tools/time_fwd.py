#!/usr/bin/env python3
"""Tuning aid: time ssv_conv1d_fwd at a few shapes with the current env (SSV_HIP_LIB, SSV_NNB_TILE)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spoofsv_amd import _lib
P = lambda t: ctypes.c_void_p(t.data_ptr())
dev = "cuda:0"
out = []
for (B, Cin, Cout, L, k) in [(32, 256, 512, 325, 3), (32, 512, 256, 325, 3), (32, 512, 1024, 186, 3), (32, 512, 1024, 1300, 3), (32, 512, 513, 1300, 1)]:
    x = torch.randn(B, Cin, L, device=dev); w = torch.randn(Cout, Cin, k, device=dev) * 0.05; y = torch.empty(B, Cout, L, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    nb = _lib.query("ssv_conv1d_fwd_workspace", Cin, Cout, k); ws = torch.empty(max(nb, 256), dtype=torch.uint8, device=dev)
    run = lambda: _lib.call("ssv_conv1d_fwd", P(x), Cin * L, P(w), None, None, None, P(y), Cout * L, B, Cin, Cout, L, k, 1, 0, P(ws), nb, st)
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    out.append("%.1fus" % (e0.elapsed_time(e1) / 10 * 1e3))
print(os.environ.get("SSV_HIP_LIB", "default"), " ".join(out), flush=True)
