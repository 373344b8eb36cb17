#!/usr/bin/env python3
"""Experiment (GPU box): a chain of highway layers (forward, then backward) on the whole batch in one stream versus the batch cut
into 2 / 4 lanes, every lane a chain of its own on its own stream -- do the per-kernel ramp / drain phases of one lane hide under
the other lanes' main loops?  Each variant is captured in a hipGraph and replayed."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spoofsv_amd import _lib, resident
P = lambda t: ctypes.c_void_p(t.data_ptr())
dev = "cuda:0"
S = lambda s: ctypes.c_void_p(s.cuda_stream)


def run_variant(B, C, L, nl, lanes):
    k, d = 3, 1
    ws_ = [torch.randn(2 * C, C, k, device=dev) * 0.03 for _ in range(nl)]
    rw = resident.ResidentWeights(ws_); rw.refresh(S(torch.cuda.current_stream()))
    bias = torch.randn(2 * C, device=dev); g = torch.rand(C, device=dev) + 0.5; bb = torch.randn(C, device=dev)
    acts = [torch.randn(B, C, L, device=dev) for _ in range(nl + 1)]
    hs = [torch.empty(B, 2 * C, L, device=dev) for _ in range(nl)]
    stats = [torch.empty(B, 4, L, device=dev) for _ in range(nl)]
    grads = [torch.randn(B, C, L, device=dev) for _ in range(nl + 1)]
    Bl = B // lanes
    dws = [[torch.empty_like(ws_[0]) for _ in range(lanes)] for _ in range(nl)]
    pgs = [[torch.empty(6, C, device=dev) for _ in range(lanes)] for _ in range(nl)]
    nbf = _lib.query("ssv_highway_conv1d_fwd_workspace", Bl, C, L, k)
    nbb = _lib.query("ssv_highway_conv1d_bwd_workspace", Bl, C, L, k)
    wsf = [torch.empty(max(nbf, 256), dtype=torch.uint8, device=dev) for _ in range(lanes)]
    wsb = [[torch.empty(max(nbb, 256), dtype=torch.uint8, device=dev) for _ in range(lanes)] for _ in range(nl)]
    streams = [torch.cuda.Stream() for _ in range(lanes)]
    off = lambda t, lane: ctypes.c_void_p(t.data_ptr() + 4 * lane * Bl * t.stride(0))

    def step():
        cur = torch.cuda.current_stream()
        for ln, st in enumerate(streams):
            st.wait_stream(cur)
            s = S(st)
            for i in range(nl):
                _lib.call("ssv_highway_conv1d_fwd", off(acts[i], ln), C * L, P(ws_[i]), resident.lookup(ws_[i]), P(bias), P(g), P(bb), P(g), P(bb),
                          off(hs[i], ln), off(stats[i], ln), off(acts[i + 1], ln), C * L, Bl, C, L, k, d, 1, P(wsf[ln]), nbf, s)
            for i in reversed(range(nl)):
                _lib.call("ssv_highway_conv1d_bwd", off(grads[i + 1], ln), C * L, off(acts[i], ln), C * L, P(ws_[i]), resident.lookup(ws_[i]), P(g), P(bb), P(g), P(bb),
                          off(hs[i], ln), off(stats[i], ln), off(grads[i], ln), C * L, P(dws[i][ln]), P(pgs[i][ln]), Bl, C, L, k, d, 1, P(wsb[i][ln]), nbb, s)
        for st in streams:
            cur.wait_stream(st)
        if lanes > 1:                      # the lanes' partial parameter gradients are summed (one small kernel per layer)
            for i in range(nl):
                torch.add(dws[i][0], dws[i][1], out=dws[i][0])
    s0 = torch.cuda.Stream(); s0.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s0):
        step()
    torch.cuda.current_stream().wait_stream(s0); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        step()
    for _ in range(3): gr.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): gr.replay()
    e1.record(); torch.cuda.synchronize()
    resident.invalidate(ws_)
    return e0.elapsed_time(e1) / 10


for (B, C, L, nl) in [(32, 256, 325, 8), (32, 512, 186, 6), (32, 256, 1300, 4)]:
    res = {lanes: run_variant(B, C, L, nl, lanes) for lanes in (1, 2, 4)}
    print("B%d C%d L%d, %d highway layers fwd+bwd: " % (B, C, L, nl) + "  ".join("%d lane(s) %.3f ms" % (k, v) for k, v in res.items()), flush=True)
    torch.cuda.empty_cache()
