"""GPU accuracy tests of the split-fp16 arithmetic (the default mode) against float64, next to the exact-fp32 MFMA
kernels on the same operands: the claim "fp32-grade" is held here, per GEMM, on operands that exercise the power-of-two scales --
unit scale, rows spread over 2^12, an overall scale of 1e-7, outliers -- and on the documented worst case, one element 2^20
above everything else in its batch item.  The reference computes these products in fp32 (nn.Conv1d, models/TTSModel.py:59,78;
autograd's data and weight gradients behind train/ordinary.py:237)."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

SHAPES = [(4, 256, 325, 3, 27, True), (4, 512, 186, 3, 3, False), (4, 256, 325, 1, 1, False)]


def _rl2(a, b):
    return float((a.detach().double().cpu() - b).norm() / b.norm())


def _reference(x, w, dy, k, d, causal):
    pad = d * (k - 1)
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    xin = F.pad(xd, (pad, 0)) if causal else F.pad(xd, (pad // 2, pad // 2))
    yd = F.conv1d(xin, wd, None, dilation=d)
    yd.backward(dy.double())
    return yd.detach(), xd.grad, wd.grad


def _hip(x, w, dy, k, d, causal, prec):
    import spoofsv_amd
    from spoofsv_amd import ops
    prev = spoofsv_amd.set_precision(prec)
    try:
        xg, wg = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
        y = ops.conv1d(xg, wg, None, k, d, causal)
        y.backward(dy.to(DEV))
        torch.cuda.synchronize()
        return y.detach(), xg.grad, wg.grad
    finally:
        spoofsv_amd.set_precision(prev)


def _operands(kind, B, C, L, k, gen):
    x = torch.randn(B, C, L, generator=gen)
    w = torch.randn(2 * C, C, k, generator=gen) * 0.03
    dy = torch.randn(B, 2 * C, L, generator=gen)
    if kind == "spread":                   # per-row magnitudes over 2^12 (what a gradient through LayerNorm scales looks like)
        dy = dy * torch.exp2(torch.randint(-12, 1, (B, 2 * C, 1), generator=gen).float())
        x = x * torch.exp2(torch.randint(-8, 5, (B, C, 1), generator=gen).float())
    elif kind == "tiny":                   # overall scale 1e-7: raw fp16 would flush every element
        dy, x = dy * 1e-7, x * 3e-6
    elif kind == "outliers":               # a few entries 300x the rest
        dy[:, 0, :3] = 300.0
        x[:, 1, 5:8] = -300.0
        w[3, 2, 0] = 9.0
    elif kind == "gradient":               # the tool's original wide case: all three at once
        dy = dy * 1e-7 * torch.exp2(torch.randint(-12, 1, (B, 2 * C, 1), generator=gen).float())
        dy[0, 0, :3] = 3e-5
        x = x * torch.exp2(torch.randint(-8, 3, (B, C, 1), generator=gen).float())
    return x, w, dy


@pytest.mark.parametrize("kind", ["unit", "spread", "tiny", "outliers", "gradient"])
def test_split_fp16_gemms_are_as_close_to_float64_as_the_exact_fp32_kernels(kind):
    """Forward, data gradient and weight gradient of real layer shapes: relative L2 error against float64 of the split-fp16 kernels
    <= 1.5 x the exact-fp32 MFMA kernels' on the same operands (both are bounded by the fp32 accumulation they share; the
    representation error 2^-22 must not show), and <= 2e-6 absolutely."""
    gen = torch.Generator().manual_seed(11)
    worst = {}
    for (B, C, L, k, d, causal) in SHAPES:
        x, w, dy = _operands(kind, B, C, L, k, gen)
        ref = _reference(x, w, dy, k, d, causal)
        e16 = [_rl2(a, b) for a, b in zip(_hip(x, w, dy, k, d, causal, "f16x2"), ref)]
        e32 = [_rl2(a, b) for a, b in zip(_hip(x, w, dy, k, d, causal, "fp32"), ref)]
        for name, a, b in zip(("fwd", "dgrad", "wgrad"), e16, e32):
            print("%-8s B%d C%d L%d k%d d%-2d %-5s  f16x2 %.2e  fp32 %.2e  ratio %.2f" % (kind, B, C, L, k, d, name, a, b, a / b))
            assert a <= 1.5 * b and a <= 2e-6, (kind, (B, C, L, k, d), name, a, b)
            worst[name] = max(worst.get(name, 0.0), a / b)
    print("worst f16x2 / fp32 error ratio:", worst)


def test_split_fp16_one_element_2_to_20_above_the_rest_meets_the_documented_bound():
    """an element more than 2^17 below its batch item's maximum keeps an ABSOLUTE error of 2^-40 of that maximum
    (its lo half is a subnormal fp16 number) instead of 22 significand bits.  One input element 2^20 above everything else in
    item 0: every output of that item must stay within  2^-39 amax sum_k |w_mk|  (the representation bound, amax rounded up to
    the power of two the scale uses) plus the fp32-accumulation level of the exact kernels; the other items are untouched."""
    gen = torch.Generator().manual_seed(12)
    B, C, L, k, d, causal = 4, 256, 325, 3, 3, False
    x, w, dy = _operands("unit", B, C, L, k, gen)
    big = float(x.abs().max()) * 2.0 ** 20
    x[0, 7, 100] = big
    yd, _, _ = _reference(x, w, dy, k, d, causal)
    y16, _, _ = _hip(x, w, dy, k, d, causal, "f16x2")
    y32, _, _ = _hip(x, w, dy, k, d, causal, "fp32")
    err16 = (y16.double().cpu() - yd).abs()
    err32 = (y32.double().cpu() - yd).abs()
    # columns the outlier reaches (taps at -d, 0, +d) carry products of size |w| * big: compared relatively below
    touched = torch.zeros(L, dtype=torch.bool)
    touched[[100 - d, 100, 100 + d]] = True
    rows_l1 = w.double().abs().sum(dim=(1, 2))                                  # sum_k |w_mk| per output row
    bound = 2.0 ** -39 * big * rows_l1[:, None] + 4.0 * err32[0][:, ~touched].max()     # (the exact kernels' level on the same columns)
    off = err16[0][:, ~touched]
    assert bool((off <= bound.expand(-1, L)[:, ~touched]).all()), (float(off.max()), float(bound.min()))
    print("item 0, untouched columns: max abs error %.3e (bound %.3e, exact fp32 %.3e; rms of y %.3e)"
          % (float(off.max()), float(bound.min()), float(err32[0][:, ~touched].max()), float(yd[0].pow(2).mean().sqrt())))
    # the three columns that contain the outlier's products: relative to those products
    on = err16[0][:, touched] / yd[0][:, touched].abs().clamp_min(1e-30)
    assert float(on.median()) < 1e-6, float(on.median())
    # items 1..3 have scales of their own (per batch item): same accuracy as without the outlier
    for b in range(1, B):
        assert _rl2(y16[b], yd[b]) <= 1.5 * _rl2(y32[b], yd[b]) + 1e-9, b


def _gate_ref(h, x, g1, b1, g2, b2):
    """models/TTSModel.py:79-83 on (B, C, L) tensors in float64."""
    C = x.shape[1]
    ln = lambda t, g, b: F.layer_norm(t.permute(0, 2, 1), (C,), g, b, 1e-5).permute(0, 2, 1)
    s = torch.sigmoid(ln(h[:, :C], g1, b1))
    return s * ln(h[:, C:], g2, b2) + (1 - s) * x


@pytest.mark.parametrize("B,C,L", [(3, 320, 77), (2, 512, 70), (2, 192, 1030), (1, 384, 64), (2, 256, 1089), (2, 448, 33),
                                   (32, 256, 325), (32, 256, 1300), (40, 256, 100), (3, 256, 650), (64, 256, 40), (256, 256, 19), (300, 256, 17)])
def test_highway_gate_backward_wide_tile_kernels_at_ragged_shapes(B, C, L):
    """The LayerNorm / gate backward on its wide-tile kernels (round 4: 1024 threads, 32- or 64-column tiles; chosen for C > 256 or
    L >= 1024) at shapes that are multiples of nothing: ragged last column tile, last channel step partly empty, and one shape below
    the rule (16-column kernel) for comparison -- every gradient against float64 autograd of the reference's expression.
    C = 256 (round 5) runs the PERSISTENT kernel: one workgroup per CU owns a column range of one item and walks it in double-buffered
    16-column sub-tiles -- the timed step's own launches (32 x 325: 40 / 41 columns = 3 sub-tiles per workgroup; 32 x 1300: 11), ranges
    shorter than a sub-tile, fewer ranges than CUs, one workgroup per item, and more items than CUs (back on the tile kernel)."""
    from spoofsv_amd import ops
    gen = torch.Generator().manual_seed(5)
    h = torch.randn(B, 2 * C, L, generator=gen)
    x = torch.randn(B, C, L, generator=gen)
    ps = [torch.rand(C, generator=gen) + 0.5, torch.randn(C, generator=gen) * 0.3, torch.rand(C, generator=gen) + 0.5, torch.randn(C, generator=gen) * 0.3]
    gy = torch.randn(B, C, L, generator=gen)
    ref_in = [t.double().requires_grad_(True) for t in (h, x, *ps)]
    _gate_ref(*ref_in).backward(gy.double())
    dev_in = [t.to(DEV).requires_grad_(True) for t in (h, x, *ps)]
    y = ops.highway_gate_dd(*dev_in)
    y.backward(gy.to(DEV))
    torch.cuda.synchronize()
    for name, a, b in zip(("h", "x", "g1", "b1", "g2", "b2"), dev_in, ref_in):
        e = _rl2(a.grad, b.grad)
        assert e < 3e-6, (name, (B, C, L), e)


@pytest.mark.parametrize("B,C,L,act", [(2, 513, 129, 1), (2, 576, 65, 2), (3, 320, 1030, 0), (2, 272, 40, 1)])
def test_channel_layernorm_act_backward_wide_tile_kernels_at_ragged_shapes(B, C, L, act):
    """Same for LayerNorm + activation after a 1x1 convolution (models/TTSModel.py:128-131, :353-361): the conv + LN + act operator's
    backward runs ln_act_bwd (wide tiles for 256 < C <= 576) -- input, weight, bias and LayerNorm parameter gradients vs float64."""
    from spoofsv_amd import ops
    gen = torch.Generator().manual_seed(6)
    Cin = 48
    x = torch.randn(B, Cin, L, generator=gen)
    w = torch.randn(C, Cin, 1, generator=gen) * 0.2
    bias = torch.randn(C, generator=gen) * 0.1
    gam, bet = torch.rand(C, generator=gen) + 0.5, torch.randn(C, generator=gen) * 0.3
    gy = torch.randn(B, C, L, generator=gen)

    def ref(x, w, bias, gam, bet):
        n = F.layer_norm(F.conv1d(x, w, bias).permute(0, 2, 1), (C,), gam, bet, 1e-5).permute(0, 2, 1)
        return torch.relu(n) if act == 1 else torch.sigmoid(n) if act == 2 else n
    ref_in = [t.double().requires_grad_(True) for t in (x, w, bias, gam, bet)]
    ref(*ref_in).backward(gy.double())
    import spoofsv_amd
    prev = spoofsv_amd.set_precision("fp32")           # exact products: what is tested here is the LayerNorm backward
    try:
        dev_in = [t.to(DEV).requires_grad_(True) for t in (x, w, bias, gam, bet)]
        ops.pointwise_conv_ln_act(*dev_in, None, act).backward(gy.to(DEV))
        torch.cuda.synchronize()
    finally:
        spoofsv_amd.set_precision(prev)
    for name, a, b in zip(("x", "w", "bias", "gamma", "beta"), dev_in, ref_in):
        e = _rl2(a.grad, b.grad)
        assert e < 5e-6, (name, (B, C, L, act), e)


RING_SHAPES = [
    # B, Cin, Cout, L, d, causal   (k = 3 everywhere: the weight gradient's ring kernel, gemm_nt3r_kernel)
    (2, 256, 512, 325, 1, False), (2, 256, 512, 325, 27, True), (3, 64, 128, 186, 3, False), (3, 64, 128, 186, 27, False),
    (1, 64, 128, 128, 9, True), (1, 64, 128, 64, 1, False), (5, 64, 128, 65, 3, True), (2, 64, 128, 63, 27, True),
    (2, 80, 100, 70, 9, False), (4, 24, 40, 33, 1, True), (2, 200, 130, 129, 3, False), (7, 72, 200, 191, 27, False),
    (2, 64, 64, 1300, 1, False), (32, 64, 128, 8, 1, False), (3, 64, 128, 193, 27, True), (2, 64, 128, 100, 13, False),
]


@pytest.mark.parametrize("prec,tol", [("f16x2", 2e-6), ("bf16x3", 2e-5)])
def test_weight_gradient_ring_kernel_ragged_shapes_dilations_and_item_boundaries(prec, tol):
    """The k = 3 weight gradient stages its input once per 64-step chunk into a ring of time-major LDS rows and reads every tap's operand by
    transposed reads at shifted rows (round 4).  What can go wrong there is all about edges: rows before an item's first and past its last
    time step (zeros of the convolution's padding), the virtual time line running from one batch item into the next, ring wrap-around
    (the mirror rows), row lengths that are multiples of nothing, channel and row tiles partly empty, the longest shifts the entry takes
    (54: dilation 27, causal) -- each against float64 autograd of nn.Conv1d's expression (models/TTSModel.py:59-61, :78)."""
    gen = torch.Generator().manual_seed(21)
    for (B, Cin, Cout, L, d, causal) in RING_SHAPES:
        x = torch.randn(B, Cin, L, generator=gen)
        w = torch.randn(Cout, Cin, 3, generator=gen) * 0.05
        dy = torch.randn(B, Cout, L, generator=gen)
        _, _, wref = _reference(x, w, dy, 3, d, causal)
        _, _, wg = _hip(x, w, dy, 3, d, causal, prec)
        assert bool(torch.isfinite(wg).all()), (B, Cin, Cout, L, d, causal)
        # per tap: a wrong edge shows in one tap only and would hide in the norm over all three
        for j in range(3):
            e = _rl2(wg[:, :, j], wref[:, :, j])
            print("%-6s B%d Cin%d Cout%d L%d d%d %s tap %d: %.2e" % (prec, B, Cin, Cout, L, d, "causal" if causal else "same", j, e))
            assert e <= tol, ((B, Cin, Cout, L, d, causal), j, e)
        worst = float((wg.double().cpu() - wref).abs().max() / wref.abs().max())
        assert worst <= 20 * tol, ((B, Cin, Cout, L, d, causal), worst)


def _multi_wgrad(xs, dys, specs, k, max_shift, prec):
    """ssv_conv1d_bwd_weight_multi on equal-shaped jobs with their own (dilation, causal); returns the dw tensors."""
    import ctypes
    import spoofsv_amd
    from spoofsv_amd import ops, _lib
    prev = spoofsv_amd.set_precision(prec)
    try:
        B, Cin, L = xs[0].shape
        Cout = dys[0].shape[1]
        xg, dyg = [t.to(DEV) for t in xs], [t.to(DEV) for t in dys]
        dws = [torch.full((Cout, Cin, k), 7.0, device=DEV) for _ in xs]
        xa, dya = [ops.amax_of(t) for t in xg], [ops.amax_of(t) for t in dyg]
        table = (_lib.WgradJob * len(xs))()
        for t, x_, dy_, dw_, xa_, dya_, (d, causal) in zip(table, xg, dyg, dws, xa, dya, specs):
            sh = (ctypes.c_int * 3)()
            _lib.call("ssv_conv_shifts", k, d, int(causal), sh)
            t.dy, t.x, t.dw, t.part, t.pgrads = dy_.data_ptr(), x_.data_ptr(), dw_.data_ptr(), None, None
            t.shift[0], t.shift[1], t.shift[2] = sh[0], sh[1], sh[2]
            t.dy_amax, t.x_amax, t.dy_namax, t.x_namax = dya_.data_ptr(), xa_.data_ptr(), dya_.numel(), xa_.numel()
        tdev = torch.frombuffer(bytearray(bytes(table)), dtype=torch.uint8).to(DEV)
        nb = _lib.query("ssv_conv1d_bwd_weight_multi_workspace", len(xs), B, Cin, Cout, L, k)
        ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
        P = lambda t: ctypes.c_void_p(t.data_ptr())
        _lib.call("ssv_conv1d_bwd_weight_multi", P(tdev), len(xs), Cout * L, Cin * L, B, Cin, Cout, L, k, max_shift, 0, 0, P(ws), nb,
                  ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        return dws
    finally:
        spoofsv_amd.set_precision(prev)


def test_batched_weight_gradient_with_mixed_dilations_and_the_shift_bound():
    """ssv_conv1d_bwd_weight_multi: jobs of one shape but different dilations in ONE launch (what the training step does with its
    highway stacks: dilations 1, 3, 9, 27 share a launch).  The job table lives on the device, so the caller states a bound on the
    shifts: with the true bound the ring kernel runs, with -1 the general kernel -- both within the split-fp16 bar of float64 -- and a
    bound that a job exceeds must poison THAT job's gradient (NaN), not return a wrong one."""
    gen = torch.Generator().manual_seed(31)
    B, Cin, Cout, L, k = 6, 128, 256, 200, 3
    specs = [(1, False), (27, True), (9, False), (3, True), (27, False)]
    xs = [torch.randn(B, Cin, L, generator=gen) for _ in specs]
    dys = [torch.randn(B, Cout, L, generator=gen) for _ in specs]
    ws = [torch.zeros(Cout, Cin, k) for _ in specs]
    refs = [_reference(x, w, dy, k, d, causal)[2] for x, w, dy, (d, causal) in zip(xs, ws, dys, specs)]
    for max_shift in (54, 64, -1):
        got = _multi_wgrad(xs, dys, specs, k, max_shift, "f16x2")
        for (d, causal), g, r in zip(specs, got, refs):
            e = _rl2(g, r)
            print("max_shift %3d  d%-2d %-6s: %.2e" % (max_shift, d, "causal" if causal else "same", e))
            assert e <= 2e-6, (max_shift, d, causal, e)
    got = _multi_wgrad(xs, dys, specs, k, 27, "f16x2")           # the causal dilation-27 job shifts by 54
    for (d, causal), g, r in zip(specs, got, refs):
        if d == 27 and causal:
            assert bool(torch.isnan(g).all()), "a job beyond the stated shift bound must come back as NaN"
        else:
            assert _rl2(g, r) <= 2e-6, (d, causal)


@pytest.mark.parametrize("d,causal", [(7, False), (8, False), (8, True), (9, False), (9, True), (1, True)])
def test_conv_forward_and_data_gradient_on_both_sides_of_the_narrow_halo_rule(d, causal):
    """The k = 3 forward / data-gradient kernels come in two instantiations: a 16-column halo for layers whose taps span at most 16
    columns, 54 otherwise (csrc/bf3_tuning.h, SSV_NN_HALO_SMALL).  Dilations 7 and 8 are the last ones on the narrow side (span 14 / 16),
    9 the first on the wide side: forward, data gradient and weight gradient against float64 at a row length that is a multiple of
    nothing, so that the last column tile is partly empty and its halo runs off the row (nn.Conv1d with the reference's padding,
    models/TTSModel.py:59-61)."""
    gen = torch.Generator().manual_seed(41)
    B, Cin, Cout, L, k = 3, 128, 256, 333, 3
    x = torch.randn(B, Cin, L, generator=gen)
    w = torch.randn(Cout, Cin, k, generator=gen) * 0.05
    dy = torch.randn(B, Cout, L, generator=gen)
    ref = _reference(x, w, dy, k, d, causal)
    for prec, tol in (("f16x2", 2e-6), ("bf16x3", 3e-5)):
        got = _hip(x, w, dy, k, d, causal, prec)
        for name, a, b in zip(("fwd", "dgrad", "wgrad"), got, ref):
            e = _rl2(a, b)
            print("%-6s d%d %-6s %-5s %.2e" % (prec, d, "causal" if causal else "same", name, e))
            assert e <= tol, (prec, d, causal, name, e)
            # the columns next to the row's ends are where a wrong halo shows: compare them on their own
            if name != "wgrad":
                edge = torch.cat((a[..., : 2 * d + 2].double().cpu() - b[..., : 2 * d + 2], a[..., -(2 * d + 2):].double().cpu() - b[..., -(2 * d + 2):]), -1)
                scale = b.abs().max()
                assert float(edge.abs().max() / scale) <= 40 * tol, (prec, d, causal, name)


@pytest.mark.parametrize("B,Cin,Cout,L", [(3, 520, 513, 333), (2, 513, 513, 1300), (5, 1030, 257, 70), (2, 512, 513, 64), (3, 2100, 129, 200)])
def test_one_by_one_convolutions_with_128_j_plus_1_output_rows(B, Cin, Cout, L):
    """128 j + 1 output channels (SSRN's 513 frequency bins, models/TTSModel.py:353-361): the wide k = 1 forward / data-gradient kernel and
    the k = 1 weight-gradient kernel keep the LAST row out of their MFMA tiles and add it as fp32 dot products beside the staging
    (csrc/conv_nn.hip, wgrad_nt.hip: gemm_nn_bf3w_kernel<.., XR = 1> for long sequences, gemm_nt_bf3_kernel<.., XR = 1> with RANGE slabs -- a workgroup's
    chunk range starts and ends inside batch items).  Forward, data gradient (whose output rows are the INPUT channels: 513 -> the extra
    row there too) and weight gradient against float64, at ragged lengths and channel counts (more slabs than chunks at L = 64, item
    boundaries inside every range), and the last row on its own (where a wrong row would hide in an L2 norm over 513 rows)."""
    gen = torch.Generator().manual_seed(B * 1000 + Cin + Cout + L)
    x = torch.randn(B, Cin, L, generator=gen)
    w = torch.randn(Cout, Cin, 1, generator=gen) * 0.05
    dy = torch.randn(B, Cout, L, generator=gen)
    ref = _reference(x, w, dy, 1, 1, False)
    for prec, tol in (("f16x2", 2e-6), ("bf16x3", 3e-5)):
        got = _hip(x, w, dy, 1, 1, False, prec)
        for name, a, b in zip(("fwd", "dgrad", "wgrad"), got, ref):
            e = _rl2(a, b)
            print("%-6s %4d -> %4d L %4d %-5s %.2e" % (prec, Cin, Cout, L, name, e))
            assert e <= tol, (prec, name, e)
        # the last output row of the forward and of the weight gradient, and the last input channel's row of the data gradient
        for name, a, b in (("fwd row", got[0][:, -1], ref[0][:, -1]), ("wgrad row", got[2][-1], ref[2][-1]), ("dgrad row", got[1][:, -1], ref[1][:, -1])):
            e = _rl2(a, b)
            assert e <= 2 * tol, (prec, name, e)


@pytest.mark.parametrize("B,Cin,Cout,L", [(5, 1024, 2048, 100), (3, 1152, 1920, 333), (7, 1024, 2304, 64)])
def test_one_by_one_weight_gradient_on_the_large_output_tiles(B, Cin, Cout, L):
    """k = 1 weight gradients whose output is 2^21 elements or more with whole 128-row / 128-column tiles -- the shape class of the GE2E embedder's LSTM
    weight gradients (GE2E/speech_embedder_net.py:19: 3072 x 768 over 120 frames x 880 utterances), reached here through the convolution entry: the
    256 x 128 tile on 8 waves (Cout % 256 == 0) and the 128 x 128 tile (Cout % 256 == 128), with the column tiles of a row tile adjacent
    (csrc/wgrad_nt.hip, ssv_nt_bf3_tile and the tile order).  Ragged lengths, more tiles than one XCD holds, against float64."""
    gen = torch.Generator().manual_seed(B * 1000 + Cin + Cout + L)
    x = torch.randn(B, Cin, L, generator=gen)
    w = torch.randn(Cout, Cin, 1, generator=gen) * 0.05
    dy = torch.randn(B, Cout, L, generator=gen)
    ref = _reference(x, w, dy, 1, 1, False)
    for prec, tol in (("f16x2", 2e-6), ("bf16x3", 3e-5)):
        got = _hip(x, w, dy, 1, 1, False, prec)
        e = _rl2(got[2], ref[2])
        print("%-6s %4d -> %4d L %4d wgrad %.2e" % (prec, Cin, Cout, L, e))
        assert e <= tol, (prec, e)
        # every 128 x 128 block of the output on its own: a misplaced tile would hide in the norm over the whole matrix
        d = (got[2].double().cpu() - ref[2]).squeeze(-1)
        r = ref[2].squeeze(-1)
        for i in range(0, Cout, 128):
            for j in range(0, Cin, 128):
                assert float(d[i:i + 128, j:j + 128].norm() / r[i:i + 128, j:j + 128].norm()) <= 3 * tol, (prec, i, j)


@pytest.mark.parametrize("gate,B,C,L", [(0, 2, 513, 129), (0, 3, 320, 1030), (0, 2, 256, 100), (1, 2, 512, 70), (1, 2, 256, 1089), (1, 3, 192, 77)])
def test_layernorm_backward_sums_only_the_partial_rows_it_wrote(gate, B, C, L):
    """The backward kernels leave one partial row of parameter-gradient sums per column tile in the caller's workspace -- compact, one
    per tile of whichever kernel the shape picks (ABI version 4: ssv_ln_bwd_partial_rows) -- and a reduction sums them.  The workspace is
    handed over FULL OF NaN: a reduction that read a row nobody wrote (the 16-column numbering of the wide kernels' rows, as before
    version 4) would return NaN parameter gradients.  Through the C ABI, wide-tile and 16-column shapes, against float64."""
    from spoofsv_amd import _lib
    gen = torch.Generator().manual_seed(17 + C + L)
    q = _lib.query
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    gy = torch.randn(B, C, L, generator=gen)
    x = torch.randn(B, C, L, generator=gen)
    if gate:
        h = torch.randn(B, 2 * C, L, generator=gen)
        ps = [torch.rand(C, generator=gen) + 0.5, torch.randn(C, generator=gen) * 0.3, torch.rand(C, generator=gen) + 0.5, torch.randn(C, generator=gen) * 0.3]
        ref_in = [t.double().requires_grad_(True) for t in (h, x, *ps)]
        _gate_ref(*ref_in).backward(gy.double())
        hd, xd, gyd = h.to(DEV), x.to(DEV), gy.to(DEV)
        psd = [p.to(DEV) for p in ps]
        y, stats = torch.empty(B, C, L, device=DEV), torch.empty(B, 4, L, device=DEV)
        _lib.call("ssv_highway_gate_fwd", P(hd), P(xd), C * L, *[P(p) for p in psd], P(stats), P(y), C * L, None, B, C, L, st)
        nb = q("ssv_highway_gate_bwd_workspace", B, C, L)
        ws = torch.full((nb // 4 + 64,), float("nan"), device=DEV)
        dh, dx, pg = torch.empty(B, 2 * C, L, device=DEV), torch.empty(B, C, L, device=DEV), torch.empty(6, C, device=DEV)
        _lib.call("ssv_highway_gate_bwd", P(gyd), C * L, P(xd), C * L, *[P(p) for p in psd], P(hd), P(stats), P(dh), P(dx), C * L, P(pg), B, C, L, P(ws), nb, st)
        torch.cuda.synchronize()
        got = {"g1": pg[0], "b1": pg[1], "g2": pg[2], "b2": pg[3]}
        want = dict(zip(("g1", "b1", "g2", "b2"), (t.grad for t in ref_in[2:])))
    else:
        gam, bet = torch.rand(C, generator=gen) + 0.5, torch.randn(C, generator=gen) * 0.3
        ref_in = [t.double().requires_grad_(True) for t in (x, gam, bet)]
        F.layer_norm(ref_in[0].permute(0, 2, 1), (C,), ref_in[1], ref_in[2], 1e-5).permute(0, 2, 1).backward(gy.double())
        xd, gyd, gd, bd = x.to(DEV), gy.to(DEV), gam.to(DEV), bet.to(DEV)
        y, stats = torch.empty(B, C, L, device=DEV), torch.empty(B, 2, L, device=DEV)
        nbf = q("ssv_channel_ln_act_fwd_workspace", B, C, L)
        wsf = torch.empty(max(nbf, 256), dtype=torch.uint8, device=DEV)
        _lib.call("ssv_channel_ln_act_fwd", P(xd), C * L, P(gd), P(bd), P(y), C * L, None, P(stats), B, C, L, 0, P(wsf), nbf, st)
        nb = q("ssv_channel_ln_act_bwd_workspace", B, C, L)
        ws = torch.full((nb // 4 + 64,), float("nan"), device=DEV)
        dx, pg = torch.empty(B, C, L, device=DEV), torch.empty(3, C, device=DEV)
        _lib.call("ssv_channel_ln_act_bwd", P(gyd), C * L, P(xd), C * L, P(stats), P(gd), P(bd), P(dx), C * L, P(pg), B, C, L, 0, P(ws), nb, st)
        torch.cuda.synchronize()
        got = {"gamma": pg[0], "beta": pg[1]}
        want = {"gamma": ref_in[1].grad, "beta": ref_in[2].grad}
    rows = _lib.lib().ssv_ln_bwd_partial_rows(gate, B, C, L, 0)
    assert 0 < rows <= _lib.lib().ssv_ln_partial_rows(B, L)
    for name in got:
        assert bool(torch.isfinite(got[name]).all()), (name, "a row nobody wrote was summed", rows)
        assert _rl2(got[name], want[name]) < 3e-6, (name, _rl2(got[name], want[name]))


@pytest.mark.parametrize("B,d,N,T", [(32, 256, 186, 325), (3, 256, 43, 97), (2, 128, 192, 64), (2, 64, 17, 200), (1, 192, 100, 1)])
def test_fused_attention_forward_and_backward_vs_float64(B, d, N, T):
    """models/TTSModel.py:266-270 in one launch per direction (csrc/attn_fused.hip: scores, column softmax and V A from MFMA accumulators;
    backward dA, dS and dQ likewise, dK / dV on the weight-gradient kernel): A, cat(R, Q) and the gradients of K|V and Q -- with an external
    gradient on A, as the guided-attention loss supplies -- against float64 at the configured size (d = 256, N = 186, T = 325, B = 32),
    ragged text / frame counts, fewer channels, a single frame."""
    from spoofsv_amd import ops
    gen = torch.Generator().manual_seed(21 + N)
    kv = torch.randn(B, 2 * d, N, generator=gen)
    q = torch.randn(B, d, T, generator=gen)
    g_rq = torch.randn(B, 2 * d, T, generator=gen)
    g_a = torch.randn(B, N, T, generator=gen) * 0.1
    kvr, qr = kv.double().requires_grad_(True), q.double().requires_grad_(True)
    a_ref = torch.softmax(torch.matmul(kvr[:, :d].transpose(1, 2), qr) / d ** 0.5, dim=1)
    rq_ref = torch.cat((torch.matmul(kvr[:, d:], a_ref), qr), dim=1)
    (rq_ref * g_rq.double()).sum().add((a_ref * g_a.double()).sum()).backward()
    kvg, qg = kv.to(DEV).requires_grad_(True), q.to(DEV).requires_grad_(True)
    rq, a = ops.attention_train(kvg, qg)
    ((rq * g_rq.to(DEV)).sum() + (a * g_a.to(DEV)).sum()).backward()
    torch.cuda.synchronize()
    assert float((a.double().cpu() - a_ref).abs().max() / a_ref.abs().max()) < 2e-5
    assert _rl2(rq, rq_ref.detach()) < 2e-6
    assert _rl2(kvg.grad, kvr.grad) < 5e-6 and _rl2(qg.grad, qr.grad) < 5e-6, (_rl2(kvg.grad, kvr.grad), _rl2(qg.grad, qr.grad))


def _link_backward(B, Cin, Cout, L, act, mode, seed=5):
    """Gradients of sum(gy * act(LN(conv1x1(x)))) on the HIP path with the link-backward form ``mode`` (SSV_PWLN_BWD: 0 two launches, 1 the
    one-launch kernel for up to 256 LayerNorm rows, 2 for every shape), weights resident, deferred weight gradient -- the trainers' path."""
    import os
    from spoofsv_amd import _lib, ops, resident
    os.environ["SSV_PWLN_BWD"] = str(mode)
    _lib.lib().ssv_reload_tuning()
    try:
        gen = torch.Generator().manual_seed(seed)
        x = torch.randn(B, Cin, L, generator=gen).to(DEV).requires_grad_(True)
        w = (torch.randn(Cout, Cin, 1, generator=gen) / Cin ** 0.5).to(DEV).requires_grad_(True)
        bias = torch.randn(Cout, generator=gen).to(DEV).requires_grad_(True)
        gamma = (1 + 0.1 * torch.randn(Cout, generator=gen)).to(DEV).requires_grad_(True)
        beta = (0.1 * torch.randn(Cout, generator=gen)).to(DEV).requires_grad_(True)
        gy = torch.randn(B, Cout, L, generator=gen).to(DEV)
        rw = resident.ResidentWeights([w])
        rw.refresh(torch.cuda.current_stream().cuda_stream)
        dfr = ops.DeferredWgrad()
        dfr.begin_step()
        y = ops.pointwise_conv_ln_act(x, w, bias, gamma, beta, None, act)
        with dfr:
            y.backward(gy)
        dfr.flush()
        torch.cuda.synchronize()
        return (x, w, bias, gamma, beta, gy), [t.grad.double().cpu() for t in (x, w, bias, gamma, beta)]
    finally:
        os.environ.pop("SSV_PWLN_BWD", None)
        _lib.lib().ssv_reload_tuning()
        resident.invalidate()


@pytest.mark.parametrize("B,Cin,Cout,L,act", [(2, 256, 256, 325, 1), (3, 128, 256, 70, 2), (2, 512, 256, 186, 0), (2, 96, 136, 333, 1),
                                               (1, 32, 32, 129, 1), (8, 40, 256, 17, 2), (2, 256, 80, 64, 0), (5, 200, 248, 191, 1),
                                               (2, 256, 512, 650, 1), (2, 512, 513, 1300, 1), (2, 513, 513, 1299, 2)])
def test_link_backward_in_one_launch_vs_float64_and_the_two_launch_form(B, Cin, Cout, L, act):
    """models/TTSModel.py:128-131, :173-180, :218-231, :343-361 backward: LayerNorm / activation backward and the 1x1 data gradient in ONE launch
    (pwln_bwd_kernel; round 5) -- every instantiation (row blocks per wave 1 / 2 / 4, one or two row-group units per thread, the 513th LayerNorm
    row and the 513th output row beside the tiles, ragged last column group, a channel count that is not a multiple of 32) against a float64
    autograd evaluation of the same expression, at the bar the two-launch form meets."""
    ins, g2 = _link_backward(B, Cin, Cout, L, act, 2)
    _, g0 = _link_backward(B, Cin, Cout, L, act, 0)
    x, w, bias, gamma, beta, gy = [t.detach().double().cpu() for t in ins]
    for t in (x, w, bias, gamma, beta):
        t.requires_grad_(True)
    pre = F.conv1d(x, w, bias)
    n = F.layer_norm(pre.transpose(1, 2), (Cout,), gamma, beta, 1e-5).transpose(1, 2)
    y = F.relu(n) if act == 1 else (torch.sigmoid(n) if act == 2 else n)
    y.backward(gy)
    for name, t, a2, a0 in zip(("dx", "dw", "dbias", "dgamma", "dbeta"), (x, w, bias, gamma, beta), g2, g0):
        ref = t.grad
        e2, e0 = float((a2 - ref).norm() / ref.norm()), float((a0 - ref).norm() / ref.norm())
        assert e2 < 3e-6 and e0 < 3e-6, (name, e2, e0)
        assert e2 < 2.0 * e0 + 2e-7, (name, e2, e0)


@pytest.mark.parametrize("B,C,T", [(32, 80, 325), (3, 5, 7), (2, 1, 1), (1, 80, 1301)])
def test_shift_right_kernel_matches_torch_cat_and_tags_its_scale_list(B, C, T):
    """ops.shift_right (ssv_shift_right_amax) = torch.cat((zeros, mel[:, :, :-1]), -1) of train/ordinary.py:226, bit for bit, on dense and on
    batch-strided inputs; the scale list it leaves has max |y| of every item as its maximum."""
    from spoofsv_amd import ops, train
    torch.manual_seed(B + T)
    big = torch.randn(B, 2 * C, T, device="cuda")
    for x in (big[:, :C].contiguous(), big[:, C:]):                        # dense, and items 2 C T apart
        want = torch.cat((torch.zeros_like(x[:, :, :1]), x[:, :, :-1]), dim=-1)
        got = train.shift_right(x)
        assert torch.equal(got, want)
        am = ops.amax_of(got)
        assert am is got._ssv_amax[0] and torch.equal(am.max(dim=1).values, want.abs().amax(dim=(1, 2)))
    cpu = train.shift_right(big[:, :C].cpu())                              # host tensors keep the torch expression
    assert torch.equal(cpu, torch.cat((torch.zeros(B, C, 1), big[:, :C, :-1].cpu()), dim=-1))


@pytest.mark.parametrize("B,n", [(32, 256 * 650), (3, 37), (2, 1), (5, 4096 + 3)])
def test_deinterleave_kernel_matches_torch_and_delivers_the_scale_list(B, n):
    """ssv_deinterleave2_amax: out[j][b][i] = x[b][2 i + j] and the partial maxima of |x| per item, against torch (dense and strided items,
    odd pair counts, pieces that are not multiples of the vector width)."""
    import ctypes
    from spoofsv_amd import _lib
    torch.manual_seed(n % 97)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    big = torch.randn(B, 2 * n + 6, device="cuda")
    for x, bs in ((big[:, :2 * n].contiguous(), 2 * n), (big[:, 4:4 + 2 * n], 2 * n + 6), (big[:, 1:1 + 2 * n], 2 * n + 6)):      # the last: 4-byte aligned only
        out = torch.full((2, B, n), float("nan"), device="cuda")
        am = torch.full((B, 64), float("nan"), device="cuda")
        _lib.call("ssv_deinterleave2_amax", P(x), bs, P(out), B, n, P(am), 64, st)
        torch.cuda.synchronize()
        assert torch.equal(out[0], x[:, 0::2]) and torch.equal(out[1], x[:, 1::2])
        assert torch.equal(am.max(dim=1).values, x.abs().amax(dim=1))
        _lib.call("ssv_deinterleave2_amax", P(x), bs, P(out), B, n, None, 64, st)           # without a list
        torch.cuda.synchronize()
        assert torch.equal(out[0], x[:, 0::2])


@pytest.mark.parametrize("shape", [(32, 513, 1300), (3, 7, 37), (2, 80, 325), (1, 1, 3)])
def test_spectrogram_losses_forward_and_backward_in_one_pass_vs_float64_and_the_two_kernels(shape):
    """ssv_spec_losses_fwd_bwd (round 6; train/ordinary.py:230-231, :249-250 and the dy of loss.backward(), :237,253): loss values and dy against
    float64 and against the separate forward / backward kernels, at the SSRN loss's full size, on an element count that is no multiple of 4
    (the kernel's 16-byte vectors have a tail), and with a backward that is NOT seeded with the promised tensor (must take the separate kernel)."""
    from spoofsv_amd import ops
    torch.manual_seed(5)
    y = (torch.rand(shape) * 0.98 + 0.01)
    gt = torch.rand(shape)
    gt[..., 0] = y[..., 0]                                  # exact hits: the sign of (y - gt) is 0 there
    seed = torch.tensor([0.7, 1.3], device=DEV)
    yd, gd = y.double().requires_grad_(True), gt.double()
    l1 = (gd - yd).abs().mean()
    bd = (-gd * torch.log(yd + 1e-8) - (1.0 - gd) * torch.log(1.0 - yd + 1e-8)).mean()
    (0.7 * l1 + 1.3 * bd).backward()
    outs = {}
    for form in ("fused", "separate", "other_seed"):
        yg = y.to(DEV).requires_grad_(True)
        lv = ops.spec_losses_vec(yg, gt.to(DEV), None if form == "separate" else seed)
        lv.backward(seed.clone() if form == "other_seed" else seed)
        torch.cuda.synchronize()
        outs[form] = (lv.detach().double().cpu(), yg.grad.double().cpu())
    for form, (lv, dy) in outs.items():
        assert abs(float(lv[0]) - float(l1)) <= 2e-6 * float(l1), (form, float(lv[0]), float(l1))
        assert abs(float(lv[1]) - float(bd)) <= 2e-6 * float(bd), (form, float(lv[1]), float(bd))
        assert float((dy - yd.grad).abs().max() / yd.grad.abs().max()) <= 2e-6, form
    # the same per-element formulas (hipcc contracts them differently in the two kernels: equal to rounding, not bit for bit)
    assert float((outs["fused"][1] - outs["separate"][1]).abs().max()) <= 5e-7 * float(outs["separate"][1].abs().max())
    assert torch.equal(outs["other_seed"][1], outs["separate"][1])       # not the promised seed tensor: the separate backward kernel ran


@pytest.mark.parametrize("B,rows,L", [(32, 256, 650), (3, 5, 37), (2, 1, 1), (2, 7, 325)])
def test_row_wise_deinterleave_kernel_matches_torch_and_delivers_the_scale_list(B, rows, L):
    """ssv_deinterleave2_rows_amax: out(b, 2 r + j, t) = x(b, r, 2 t + j) and the partial maxima of |x| per item, against torch (dense and strided
    items, odd row lengths, 4-byte aligned items)."""
    import ctypes
    from spoofsv_amd import _lib
    torch.manual_seed(L)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    n = rows * 2 * L
    big = torch.randn(B, n + 6, device="cuda")
    for x, bs in ((big[:, :n].contiguous(), n), (big[:, 4:4 + n], n + 6), (big[:, 1:1 + n], n + 6)):
        out = torch.full((B, 2 * rows, L), float("nan"), device="cuda")
        am = torch.full((B, 64), float("nan"), device="cuda")
        _lib.call("ssv_deinterleave2_rows_amax", P(x), bs, P(out), B, rows, L, P(am), 64, st)
        torch.cuda.synchronize()
        ref = x.reshape(B, rows, L, 2).permute(0, 1, 3, 2).reshape(B, 2 * rows, L)
        assert torch.equal(out, ref)
        assert torch.equal(am.max(dim=1).values, x.abs().amax(dim=1))
        _lib.call("ssv_deinterleave2_rows_amax", P(x), bs, P(out), B, rows, L, None, 64, st)           # without a list
        torch.cuda.synchronize()
        assert torch.equal(out, ref)


@pytest.mark.parametrize("B,C,L", [(32, 513, 1300), (32, 80, 325), (3, 5, 7), (2, 1, 1), (5, 33, 1030)])
def test_bias_gradient_kernel_matches_float64_for_dense_and_strided_items(B, C, L):
    """ssv_bias_grad: out(c) = sum over (b, t) of x(b, c, t) in one launch (the bias gradient of nn.Conv1d / nn.ConvTranspose1d under
    loss.backward(), train/ordinary.py:237), items a stride apart, against float64."""
    import ctypes
    from spoofsv_amd import _lib
    torch.manual_seed(C + L)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    big = torch.randn(B, C * L + 5, device="cuda")
    for x, bs in ((big[:, :C * L].contiguous(), C * L), (big[:, 3:3 + C * L], C * L + 5)):
        out = torch.full((C,), float("nan"), device="cuda")
        _lib.call("ssv_bias_grad", P(x), bs, P(out), B, C, L, st)
        torch.cuda.synchronize()
        ref = x.double().reshape(B, C, L).sum(dim=(0, 2))
        scale = x.double().reshape(B, C, L).abs().sum(dim=(0, 2))
        assert float(((out.double() - ref).abs() / scale).max()) < 2e-7
