#!/usr/bin/env python3
"""Experiment (GPU box): the backward chain of 16 highway layers WITHOUT weight gradients ([LayerNorm/gate backward, data gradient]
per layer: short, latency-bound kernels) on the main stream, and the 16 layers' batched weight gradient (one long launch) either
behind it on the same stream or beside it on a second stream.  Does the long launch hide under the chain?"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spoofsv_amd import _lib, resident
P = lambda t: ctypes.c_void_p(t.data_ptr())
dev = "cuda:0"
S = lambda s: ctypes.c_void_p(s.cuda_stream)
B, C, L, k, nl = 32, 256, 325, 3, 16
ws_ = [torch.randn(2 * C, C, k, device=dev) * 0.03 for _ in range(nl)]
rw = resident.ResidentWeights(ws_); rw.refresh(S(torch.cuda.current_stream()))
g = torch.rand(C, device=dev) + 0.5; bb = torch.randn(C, device=dev)
acts = [torch.randn(B, C, L, device=dev) for _ in range(nl + 1)]
hs = [torch.randn(B, 2 * C, L, device=dev) for _ in range(nl)]
stats = [torch.rand(B, 4, L, device=dev) for _ in range(nl)]
grads = [torch.randn(B, C, L, device=dev) for _ in range(nl + 1)]
dhs = [torch.randn(B, 2 * C, L, device=dev) for _ in range(nl)]
nblk = _lib.query("ssv_ln_partial_rows", B, L)
parts = [torch.empty(nblk, 6 * C, device=dev) for _ in range(nl)]
dws = [torch.empty_like(ws_[0]) for _ in range(nl)]
pgs = [torch.empty(6, C, device=dev) for _ in range(nl)]
nbd = _lib.query("ssv_highway_conv1d_bwd_data_workspace", B, C, L, k); wsd = torch.empty(max(nbd, 256), dtype=torch.uint8, device=dev)
table = (_lib.WgradJob * nl)()
sh = (ctypes.c_int * 3)(); _lib.call("ssv_conv_shifts", k, 1, 1, sh)
for i, t in enumerate(table):
    t.dy, t.x, t.dw, t.part, t.pgrads = dhs[i].data_ptr(), acts[i].data_ptr(), dws[i].data_ptr(), parts[i].data_ptr(), pgs[i].data_ptr()
    t.shift[0], t.shift[1], t.shift[2] = sh[0], sh[1], sh[2]
tdev = torch.frombuffer(bytearray(bytes(table)), dtype=torch.uint8).to(dev)
nbm = _lib.query("ssv_conv1d_bwd_weight_multi_workspace", nl, B, C, 2 * C, k); wsm = torch.empty(nbm, dtype=torch.uint8, device=dev)
side = torch.cuda.Stream()


def chain(s):
    for i in reversed(range(nl)):
        _lib.call("ssv_highway_conv1d_bwd_data", P(grads[i + 1]), C * L, P(acts[i]), C * L, P(ws_[i]), resident.lookup(ws_[i]), P(g), P(bb), P(g), P(bb),
                  P(hs[i]), P(stats[i]), P(grads[i]), C * L, P(dhs[i]), P(parts[i]), B, C, L, k, 1, 1, P(wsd), nbd, s)


def wgrad(s):
    _lib.call("ssv_conv1d_bwd_weight_multi", P(tdev), nl, 2 * C * L, C * L, B, C, 2 * C, L, k, 6 * C, nblk, P(wsm), nbm, s)


def variant(mode):
    def step():
        cur = torch.cuda.current_stream()
        if mode == "chain": chain(S(cur))
        elif mode == "wgrad": wgrad(S(cur))
        elif mode == "serial": chain(S(cur)); wgrad(S(cur))
        else:                                  # the weight gradients (of the PREVIOUS segment's layers, say) beside the chain
            side.wait_stream(cur); wgrad(S(side)); chain(S(cur)); cur.wait_stream(side)
    s0 = torch.cuda.Stream(); s0.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s0): step()
    torch.cuda.current_stream().wait_stream(s0); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr): step()
    for _ in range(3): gr.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10


print("NT target %s, Z = %d: " % (os.environ.get("SSV_NT_TARGET", "512"), _lib.query("ssv_conv1d_bwd_weight_multi_splits", nl, B, C, 2 * C, k)) +
      "  ".join("%s %.3f ms" % (m, variant(m)) for m in ("chain", "wgrad", "serial", "beside")), flush=True)
