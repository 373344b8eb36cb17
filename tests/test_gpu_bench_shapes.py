"""The launch configurations bench.py times, value-checked at ITS batch (B = 32 utterances per GPU): every distinct convolution of
the Text2Mel + SSRN training step -- forward, data gradient, weight gradient (alone and in the job-table launches that batch
equal-shaped layers) -- against float64 on the same operands, plus the two edge cases the reference's zero-padding collate and any
diverging run can hand to the operand-scale path: an all-zero batch item and a non-finite element.

Slab counts, tiles and the job-table Z all depend on B (csrc/api.hip dw_splits / nt_slabs, conv_nn.hip pick_nnb), so a test at
B = 8 launches other configurations than the timed step; the rows below are the (B, Cin, Cout, L, k) of
profiles/round4_shapes.tsv.  Reference semantics: nn.Conv1d in models/TTSModel.py:59,78 and autograd's two gradients behind
train/ordinary.py:237,253; the collate that produces all-zero tails is data/dataset.py:187-258.  The float64 reference is a sum of
per-tap matrix products on the GPU (a checker, not the product path)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
B = 32

# (Cin, Cout, L, k, [(dilation, causal), ...], highway)   highway: the forward also writes the column statistics the LayerNorm/gate kernel reads
LAYERS = [
    # Text2Mel, text encoder (models/TTSModel.py:106-140): C = 512, N = 186
    (512, 1024, 186, 3, [(1, 0), (3, 0), (9, 0), (27, 0)], True),
    (512, 1024, 186, 1, [(1, 0)], True),
    (128, 512, 186, 1, [(1, 0)], False),
    (512, 512, 186, 1, [(1, 0)], False),
    # audio encoder / decoder (:142-232): C = 256, T = 325, causal
    (256, 512, 325, 3, [(1, 1), (3, 1), (9, 1), (27, 1)], True),
    (80, 256, 325, 1, [(1, 0)], False),
    (256, 256, 325, 1, [(1, 0)], False),
    (512, 256, 325, 1, [(1, 0)], False),
    (256, 80, 325, 1, [(1, 0)], False),
    # SSRN (:319-362): 325 -> 650 -> 1300 frames, 513 linear bins
    (256, 512, 325, 3, [(1, 0), (3, 0)], True),
    (256, 512, 650, 3, [(1, 0), (3, 0)], True),
    (256, 512, 1300, 3, [(1, 0), (3, 0)], True),
    (512, 1024, 1300, 3, [(1, 0)], True),
    (256, 256, 650, 1, [(1, 0)], False),
    (256, 512, 1300, 1, [(1, 0)], False),
    (512, 513, 1300, 1, [(1, 0)], False),
    (513, 513, 1300, 1, [(1, 0)], False),
]
# Relative L2 against float64 per product (tests/test_gpu_accuracy.py holds split-fp16 to <= 2e-6 and 1.5x the exact kernels at B = 4;
# at B = 32 the weight gradient sums 32 x L terms, so its fp32 accumulation level is what is allowed here) and the largest entry
# error against the tensor's rms.
TOL = {"fwd": (1e-6, 1.5e-5), "dgrad": (1.5e-6, 2e-5), "wgrad": (2e-6, 1.5e-5)}      # measured on MI355X (profiles/round5_b32_shapes.txt): <= 3.6e-7/5.4e-6, 6.2e-7/8.5e-6, 8.7e-7/5.6e-6


def _P(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _shifts(k, d, causal):
    from spoofsv_amd import _lib
    sh = (ctypes.c_int * 3)()
    _lib.call("ssv_conv_shifts", k, d, int(causal), sh)
    return [int(v) for v in sh][:k] if k == 3 else [0]


def _shifted(x, s):
    """x(b, c, t + s) with zeros outside [0, L)."""
    if s == 0:
        return x
    out = torch.zeros_like(x)
    L = x.shape[2]
    if s > 0:
        out[:, :, :L - s] = x[:, :, s:]
    else:
        out[:, :, -s:] = x[:, :, :L + s]
    return out


def _ref64(x, w, dy, shifts):
    """float64 forward, data gradient and weight gradient (per tap) of y(b,o,t) = sum_j sum_c w(o,c,j) x(b,c,t + shift_j)."""
    xd, wd, dyd = x.double(), w.double(), dy.double()
    y = torch.zeros(x.shape[0], w.shape[0], x.shape[2], dtype=torch.float64, device=x.device)
    dx = torch.zeros_like(xd)
    dw = torch.zeros_like(wd)
    for j, s in enumerate(shifts):
        xs = _shifted(xd, s)
        y += torch.matmul(wd[:, :, j], xs)
        dx += _shifted(torch.matmul(wd[:, :, j].t(), dyd), -s)
        dw[:, :, j] = torch.einsum("bot,bct->oc", dyd, xs)
    return y, dx, dw


def _errs(a, ref):
    a = a.double()
    return float((a - ref).norm() / ref.norm()), float((a - ref).abs().max() / ref.pow(2).mean().sqrt())


def _check(what, name, a, ref):
    l2, ent = _errs(a, ref)
    t = TOL[name]
    assert l2 <= t[0] and ent <= t[1], (what, name, l2, ent)
    return l2, ent


@pytest.mark.parametrize("layer", LAYERS, ids=lambda l: "%dto%d_L%d_k%d" % l[:4])
def test_every_conv_launch_of_the_timed_step_at_its_own_batch_vs_float64(layer):
    """Forward (resident pre-split weights, with the column statistics where the step asks for them), data gradient and weight
    gradient (one layer: the slab count dw_splits picks at B = 32, incl. the 21 RANGE slabs of the 513-row tail) of one layer
    shape of the step, default arithmetic (split-fp16), B = 32."""
    import spoofsv_amd
    from spoofsv_amd import _lib, ops, resident
    assert spoofsv_amd.get_precision() == "f16x2"
    Cin, Cout, L, k, variants, highway = layer
    st = ops._stream()
    gen = torch.Generator().manual_seed(1000 + Cin + Cout + L + k)
    x = torch.randn(B, Cin, L, generator=gen).to(DEV)
    dy = torch.randn(B, Cout, L, generator=gen).to(DEV)
    w = (torch.randn(Cout, Cin, k, generator=gen) * (2.0 / (Cin * k)) ** 0.5).to(DEV)
    bias = torch.randn(Cout, generator=gen).to(DEV)
    rw = resident.ResidentWeights([w])
    rw.refresh(st)
    try:
        for d, causal in variants:
            what = "B=%d %d->%d L=%d k=%d d=%d causal=%d" % (B, Cin, Cout, L, k, d, causal)
            y64, dx64, dw64 = _ref64(x, w, dy, _shifts(k, d, causal))
            y64 += bias.double().view(1, -1, 1)
            xa, dya = ops.amax_of(x), ops.amax_of(dy)
            y = torch.full((B, Cout, L), float("nan"), device=DEV)
            cs = torch.empty(B * (Cout // 64) * L * 2, device=DEV) if (highway and Cout % 64 == 0) else None
            nb = _lib.query("ssv_conv1d_fwd_workspace", Cin, Cout, k)
            ws = torch.empty(max(nb, 256), dtype=torch.uint8, device=DEV)
            _lib.call("ssv_conv1d_fwd", _P(x), Cin * L, _P(xa), xa.shape[1], _P(w), resident.lookup(w), _P(bias), None, _P(y), Cout * L, _P(cs),
                      B, Cin, Cout, L, k, d, causal, _P(ws), nb, st)
            e = [_check(what, "fwd", y, y64)]
            if cs is not None:
                # the epilogue's by-product: per 64-row group and column, (mean, sum of squared deviations from that mean) of the output
                st64 = y64.view(B, Cout // 64, 64, L)
                got = cs.view(B, Cout // 64, L, 2).double()
                mean64 = st64.mean(2)
                m264 = (st64 - mean64.unsqueeze(2)).pow(2).sum(2)
                assert float((got[..., 0] - mean64).abs().max()) <= 2e-5 * float(mean64.abs().max()), what
                assert float((got[..., 1] - m264).abs().max()) <= 2e-5 * float(m264.abs().max()), what
            dx = ops._conv_bwd_data(dy, Cout * L, w, Cin, L, k, d, causal, dya)
            e.append(_check(what, "dgrad", dx, dx64))
            dw = ops._conv_bwd_weight(dy, Cout * L, x, Cin * L, (Cout, Cin, k), k, d, causal, dy_amax=dya, x_amax=xa)
            for j in range(k):                                   # per tap: a wrong slab or shift of one tap must not hide in the norm over three
                e.append(_check(what + " tap %d" % j, "wgrad", dw[:, :, j], dw64[:, :, j]))
            print(what, " ".join("%.1e/%.1e" % p for p in e))
    finally:
        resident.invalidate([w])


# (njobs, Cin, Cout, L, k, causal, Z the timed step's launch uses: profiles/round4_shapes.tsv)
JOB_TABLES = [
    (16, 256, 512, 325, 3, 1, 2),       # the 16 causal highway layers of audio encoder / decoder
    (10, 512, 1024, 186, 3, 0, 4),      # text encoder, k = 3
    (2, 512, 1024, 186, 1, 0, 5),       # text encoder, k = 1 highway layers
    (2, 256, 512, 325, 3, 0, 16),       # SSRN, first pair
    (5, 256, 256, 325, 1, 0, 11),       # 1x1 links of audio encoder / decoder
]


@pytest.mark.parametrize("njobs,Cin,Cout,L,k,causal,Z", JOB_TABLES, ids=lambda v: str(v))
def test_job_table_weight_gradients_at_the_timed_steps_job_counts_vs_float64(njobs, Cin, Cout, L, k, causal, Z):
    """ssv_conv1d_bwd_weight_multi as ops.DeferredWgrad.flush calls it in the timed step: `njobs` equal-shaped layers (dilations cycling
    1, 3, 9, 27 as the stacks do), one kernel launch + one reduction launch that also sums each job's partial LayerNorm / bias rows.
    Every job's dW per tap and its summed rows against float64; the slab count must be the one the timed step ran with."""
    import spoofsv_amd
    from spoofsv_amd import _lib, ops
    assert spoofsv_amd.get_precision() == "f16x2"
    assert int(_lib.lib().ssv_conv1d_bwd_weight_multi_splits(njobs, B, Cin, Cout, L, k)) == Z
    st = ops._stream()
    gen = torch.Generator().manual_seed(77 + njobs)
    n2 = 6 * Cin if Cout == 2 * Cin else 3 * Cout
    nblk = _lib.query("ssv_ln_bwd_partial_rows", 1 if Cout == 2 * Cin else 0, B, Cin if Cout == 2 * Cin else Cout, L, 1)
    assert 0 < nblk <= 768
    dils = [1, 3, 9, 27] if k == 3 else [1]
    jobs = []
    table = (_lib.WgradJob * njobs)()
    max_shift = 0
    for i, t in enumerate(table):
        x = torch.randn(B, Cin, L, generator=gen).to(DEV)
        dy = (torch.randn(B, Cout, L, generator=gen) * 10.0 ** (-(i % 4))).to(DEV)       # gradients of different magnitude per layer
        dw = torch.full((Cout, Cin, k), float("nan"), device=DEV)
        part = torch.randn(nblk, n2, generator=gen).to(DEV)
        pg = torch.full((n2,), float("nan"), device=DEV)
        xa, dya = ops.amax_of(x), ops.amax_of(dy)
        sh = _shifts(k, dils[i % len(dils)], causal)
        max_shift = max(max_shift, max(abs(v) for v in sh))
        t.dy, t.x, t.dw, t.part, t.pgrads = dy.data_ptr(), x.data_ptr(), dw.data_ptr(), part.data_ptr(), pg.data_ptr()
        for j in range(3):
            t.shift[j] = sh[j] if j < len(sh) else 0
        t.dy_amax, t.x_amax, t.dy_namax, t.x_namax = dya.data_ptr(), xa.data_ptr(), dya.numel(), xa.numel()
        jobs.append((x, dy, dw, part, pg, sh, xa, dya))
    tdev = torch.frombuffer(bytearray(bytes(table)), dtype=torch.uint8).to(DEV)
    nb = _lib.query("ssv_conv1d_bwd_weight_multi_workspace", njobs, B, Cin, Cout, L, k)
    ws = torch.empty(max(nb, 256), dtype=torch.uint8, device=DEV)
    _lib.call("ssv_conv1d_bwd_weight_multi", _P(tdev), njobs, Cout * L, Cin * L, B, Cin, Cout, L, k, max_shift, n2, nblk, _P(ws), nb, st)
    torch.cuda.synchronize()
    worst = (0.0, 0.0)
    for i, (x, dy, dw, part, pg, sh, _, _) in enumerate(jobs):
        xd, dyd = x.double(), dy.double()
        for j, s in enumerate(sh):
            ref = torch.einsum("bot,bct->oc", dyd, _shifted(xd, s))
            e = _check("job %d tap %d of %d jobs %d->%d L=%d k=%d" % (i, j, njobs, Cin, Cout, L, k), "wgrad", dw[:, :, j], ref)
            worst = (max(worst[0], e[0]), max(worst[1], e[1]))
        rows = part.double().sum(0)
        assert float((pg.double() - rows).abs().max()) <= 1e-5 * float(rows.abs().max()), ("partial rows of job", i)
    print("jobs=%d %d->%d L=%d k=%d Z=%d: worst rel L2 %.1e, worst entry/rms %.1e" % (njobs, Cin, Cout, L, k, Z, worst[0], worst[1]))


def _highway_ref64(x, w, bias, g1, b1, g2, b2, k, d, causal):
    """models/TTSModel.py:63-84 in float64 on the GPU."""
    import torch.nn.functional as F
    C = x.shape[1]
    y = _ref64(x, w, torch.zeros(x.shape[0], 2 * C, x.shape[2], device=x.device), _shifts(k, d, causal))[0] + bias.double().view(1, -1, 1)
    ln = lambda t, g, b: F.layer_norm(t.permute(0, 2, 1), (C,), g.double(), b.double(), 1e-5).permute(0, 2, 1)
    s = torch.sigmoid(ln(y[:, :C], g1, b1))
    return s * ln(y[:, C:], g2, b2) + (1 - s) * x.double()


def _highway_params(C, k, gen):
    w = (torch.randn(2 * C, C, k, generator=gen) * (2.0 / (C * k)) ** 0.5).to(DEV).requires_grad_(True)
    bias = (0.1 * torch.randn(2 * C, generator=gen)).to(DEV).requires_grad_(True)
    g1, g2 = [(1 + 0.2 * torch.randn(C, generator=gen)).to(DEV).requires_grad_(True) for _ in range(2)]
    b1, b2 = [(0.2 * torch.randn(C, generator=gen)).to(DEV).requires_grad_(True) for _ in range(2)]
    return w, bias, g1, b1, g2, b2


def test_all_zero_batch_item_through_the_operand_scale_path_on_a_real_layer_shape():
    """data/dataset.py:187-258 zero-pads every utterance to the batch maximum; a batch item (or, after the teacher-forcing shift, a
    whole input) can be all zeros.  Its operand scale comes from max |x| = 0 (csrc/ssv_common.h ssv_pow2_scale clamps the exponent):
    forward and every gradient of a real highway layer (C = 256, L = 325, k = 3, d = 3, causal) must equal float64's, for the zero
    item and for its neighbours -- in all three arithmetic modes."""
    import spoofsv_amd
    from spoofsv_amd import ops
    gen = torch.Generator().manual_seed(5)
    Bz, C, L, k, d, causal = 4, 256, 325, 3, 3, 1
    p = _highway_params(C, k, gen)
    x0 = torch.randn(Bz, C, L, generator=gen)
    x0[1] = 0.0                                           # all-zero item
    x0[2, :, 200:] = 0.0                                  # zero-padded tail
    dy = torch.randn(Bz, C, L, generator=gen).to(DEV)
    dy[3] = 0.0                                           # and an all-zero gradient item (a masked loss)
    xr = x0.to(DEV).double().requires_grad_(True)
    pr = [q.detach().double().requires_grad_(True) for q in p]
    yr = _highway_ref64(xr, *pr, k, d, causal)
    yr.backward(dy.double())
    for mode in ("f16x2", "fp32", "bf16x3"):
        prev = spoofsv_amd.set_precision(mode)
        try:
            x = x0.to(DEV).requires_grad_(True)
            for q in p:
                q.grad = None
            y = ops.highway_conv1d(x, *p, k, d, bool(causal))
            y.backward(dy)
            torch.cuda.synchronize()
            tol = 2e-4 if mode == "bf16x3" else 2e-5
            assert bool(torch.isfinite(y).all()) and bool(torch.isfinite(x.grad).all())
            for b in range(Bz):
                e = float((y[b].double() - yr[b]).abs().max() / yr[b].abs().max())
                assert e < tol, (mode, "y", b, e)
            assert float(x.grad[3].abs().max()) == 0.0                       # zero gradient in, zero gradient out: exactly
            for b in range(3):
                e = float((x.grad[b].double() - xr.grad[b]).norm() / xr.grad[b].norm())
                assert e < 10 * tol, (mode, "dx", b, e)
            for q, r, name in zip(p, pr, ("w", "bias", "g1", "b1", "g2", "b2")):
                e = float((q.grad.double() - r.grad).norm() / r.grad.norm())
                assert e < 10 * tol, (mode, name, e)
        finally:
            spoofsv_amd.set_precision(prev)


@pytest.mark.parametrize("bad", [float("nan"), float("inf"), float("-inf")])
def test_non_finite_element_propagates_through_the_operand_scale_path(bad):
    """One NaN / Inf in the input of a real highway layer: the reference's fp32 conv makes every output channel of the columns the
    element reaches non-finite, LayerNorm over channels keeps those columns non-finite, and so does the gate.  The scaled split-fp16
    path must not lose it (a scale computed from max |x| that ignores NaN, or an Inf that saturates to a finite fp16 value, would):
    every column the element reaches is non-finite in the output; every OTHER batch item is untouched and still correct."""
    import spoofsv_amd
    from spoofsv_amd import ops
    gen = torch.Generator().manual_seed(6)
    Bz, C, L, k, d, causal = 4, 256, 325, 3, 3, 1
    p = [q.detach() for q in _highway_params(C, k, gen)]
    x0 = torch.randn(Bz, C, L, generator=gen)
    clean = _highway_ref64(x0.to(DEV), *p, k, d, causal)
    x0[2, 7, 100] = bad
    touched = [100 - s for s in _shifts(k, d, causal)]            # y(t) reads x(t + shift): columns t = 100 - shift
    for mode in ("f16x2", "fp32"):
        prev = spoofsv_amd.set_precision(mode)
        try:
            with torch.no_grad():
                y = ops.highway_conv1d(x0.to(DEV), *p, k, d, bool(causal))
            torch.cuda.synchronize()
            for t in touched:
                assert not bool(torch.isfinite(y[2, :, t]).any()), (mode, bad, "column", t, "lost the non-finite value")
            for b in (0, 1, 3):
                assert bool(torch.isfinite(y[b]).all())
                assert float((y[b].double() - clean[b]).abs().max() / clean[b].abs().max()) < 2e-5, (mode, b)
            rest = torch.ones(L, dtype=torch.bool)
            rest[touched] = False
            print("%s %s: %d of %d untouched columns of the same item stay finite" % (mode, bad, int(torch.isfinite(y[2][:, rest]).all(0).sum()), int(rest.sum())))
        finally:
            spoofsv_amd.set_precision(prev)
