#!/usr/bin/env python3
"""Benchmark of the hot path: mel-frames/s of Text2Mel + SSRN training on synthetic VCTK-shaped batches.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A "step" is one optimizer iteration of train_text2mel AND one of train_ssrn (train/ordinary.py:221-254 of
the reference) on a batch of 32 utterances per GPU (N=186 characters, T=325 mel frames, 513x1300 linear
frames): forward, the reference's losses, backward, Adam -- nothing skipped.  value = mel frames consumed
by all ranks per second of that combined step ("Text2Mel+SSRN").  Inputs are resident in
HBM before the timed region.  Weak scaling: the per-GPU batch is fixed, gradients are averaged with a
flat RCCL all-reduce per step.

Rank 0 prints ONE JSON line with the driver's contract plus:
  roofline            the dominant kernel (dilated Conv1d implicit GEMM, split-fp16 MFMA) timed with HIP events here
  cpu_baseline        the CPU oracle (same stock-op sequence the reference runs) on a bounded sample
  stock_gpu_baseline  the same op sequence on torch's stock GPU kernels (MIOpen / rocBLAS), fp32, eager: the path the reference
                      itself would take on this GPU (SURVEY.md "facts"); config.eager_ms_per_step = the HIP path without hipGraphs
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_PER_GPU, N_TEXT, T_MEL = 32, 186, 325
PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0        # MI355X_MICROARCH.md: bf16 MFMA dense peak (not the 2:1-sparse figure)
SUSTAINED_F16_MFMA_TFLOPS = 2100.0    # what a kernel of nothing but v_mfma_f32_16x16x32_f16 holds on this chip (tools/probe/mfma_rate.hip: 2,070-2,156 at 1-8 waves per SIMD)
CPU_WARMUP, CPU_TIMED = 3, 10         # the CPU arm's protocol


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-adversarial", action="store_true", help="skip the extra WGAN-GP cycle timing")
    ap.add_argument("--no-fp32", action="store_true", help="skip the exact-fp32 timing of the same step")
    ap.add_argument("--no-ge2e", action="store_true", help="skip the GE2E (config 5) figures on the line")
    ap.add_argument("--no-stock", action="store_true", help="skip the stock-op (MIOpen / rocBLAS) arm on the same GPU and the eager timing of the HIP path")
    ap.add_argument("--no-roofline", action="store_true", help="skip the isolated-kernel roofline measurements (profiling runs)")
    ap.add_argument("--ge2e", action="store_true", help="measure BASELINE config 5 (GE2E speaker embedder) instead and print its JSON line")
    ap.add_argument("--batch", type=int, default=B_PER_GPU)
    ap.add_argument("--check-replicas", action="store_true",
                    help="N > 1: after the timed steps (and the adversarial cycles) all-gather a checksum of every model's parameters and fail unless all ranks hold the same replica")
    ap.add_argument("--launch-check", action="store_true",
                    help="rehearse ONLY the launch plumbing of --gpus N, without any kernel launch or device memory (device COUNTING may initialise the HIP runtime): N ranks, gloo rendezvous on 127.0.0.1, local-rank folding, "
                         "one all-gather, ONE JSON line from rank 0, non-zero exit if any rank dies (runs on a GPU-less host)")
    ap.add_argument("--precision", choices=["f16x2", "bf16x3", "fp32"], default="f16x2",
                    help="conv GEMM arithmetic: split-fp16 MFMA with power-of-two operand scales (default, ~2^-22 per product = fp32-grade), "
                         "split-bf16 MFMA (~2^-16, narrower than the reference: opt-in) or exact fp32 MFMA")
    return ap.parse_args()


class Trainer:
    """Model + optimizer + static batch behind ``train.TrainStep``: forward, losses, backward, Adam -- one hipGraph on a
    single GPU; with N > 1 the backward runs in the segments of ``tts.ddp_plan`` (separate hipGraphs) and every gradient
    bucket's RCCL all-reduce is launched between the replays, overlapping the remaining backward."""

    def __init__(self, kind, batch, dev, rank, world, use_graph):
        from spoofsv_amd import train
        from spoofsv_amd.tts import SSRN, melSyn
        self.kind, self.world = kind, world
        torch.manual_seed(1234)
        gaw = None
        if kind == "text2mel":
            self.model = melSyn(34, True, 200, textemb_dim=128, freq_bins=80, hidden_dim=256)
            data = train.synthetic_text2mel_batch(batch, N_TEXT, T_MEL, seed=rank, device=dev)
            gaw = train.guided_attention_mat(186, 325, device=dev)
        else:
            self.model = SSRN(80, 513, 256)
            data = train.synthetic_ssrn_batch(batch, T_MEL, seed=rank, device=dev)
        self.model.apply(train.init_weights)
        self.model.to(dev).train()
        self.opt = train.FusedAdam(self.model.parameters(), 2e-4, (0.5, 0.9), 1e-6, capturable=True)
        self.ddp = None
        if world > 1 or os.environ.get("SSV_FORCE_SEGMENTED") == "1" or os.environ.get("SSV_FORCE_COLLECTIVES") == "1":
            # the second and third form rehearse the N > 1 launch structure on one GPU (the third with RCCL really called)
            self.ddp = train.DataParallelRanks(model=self.model)
            self.ddp.broadcast_parameters(0)
        self.opt.refresh_resident_weights()
        self.stepper = train.TrainStep(kind, self.model, self.opt, list(data), gaw, self.ddp, graph=use_graph,
                                       defer_wgrad=os.environ.get("SSV_DEFER_WGRAD", "1") != "0")
        self.stepper.loss_log = []         # loss terms of the eager warm-up iterations 0, 1: checked against the oracle's

    def prepare(self):
        self.stepper.prepare()

    def step(self):
        self.stepper()

    @property
    def loss(self):
        return float(sum(self.stepper.out))

    def first_losses(self):
        """Loss terms of optimizer iterations 0 and 1 (initial weights; after one Adam step) as lists of floats."""
        return [[float(v) for v in t.cpu()] for t in self.stepper.loss_log[:2]]


def adversarial_cycle_ms(kind, batch, dev, world=1, cycles=3, check_replicas=False):
    """One full WGAN-GP cycle of the reference (1 generator + RATIO=5 critic iterations,
    train/adversarial_wasserstein_gp.py:261-322): generator on the HIP path, critic on the twice-differentiable HIP conv / LayerNorm / gate ops
    with gradient penalty, both iterations replayed from captured hipGraphs (train.AdversarialGraphStep); with N > 1 data
    parallel (BASELINE config 4): global adaptive weight, bucketed generator all-reduce between the backward segments, one packed
    critic all-reduce.  Returns ms per ITERATION averaged over the cycle (max over ranks)."""
    from spoofsv_amd import train
    from spoofsv_amd.critic import linDisc, melDisc
    from spoofsv_amd.tts import SSRN, melSyn
    rank = dist.get_rank() if world > 1 else 0
    torch.manual_seed(1234)
    gaw = None
    if kind == "text2mel":
        model, disc = melSyn(34, True, 200, 128, 80, 256), melDisc(80, 128)
        data = train.synthetic_text2mel_batch(batch, N_TEXT, T_MEL, seed=rank, device=dev)
        gaw = train.guided_attention_mat(186, 325, device=dev)
    else:
        model, disc = SSRN(80, 513, 256), linDisc(513, 128)
        data = train.synthetic_ssrn_batch(batch, T_MEL, seed=rank, device=dev)
    model.apply(train.init_weights); disc.apply(train.init_weights)
    model.to(dev).train(); disc.to(dev).train()
    og = train.FusedAdam(model.parameters(), 2e-4, (0.5, 0.9), 1e-6, capturable=True)
    od = train.FusedAdam(disc.parameters(), 2e-4, (0.5, 0.9), 1e-6, capturable=True)
    ddp_syn = ddp_disc = None
    if world > 1:
        ddp_syn, ddp_disc = train.DataParallelRanks(model=model), train.DataParallelRanks(list(disc.parameters()))
        ddp_syn.broadcast_parameters(0); ddp_disc.broadcast_parameters(0)
        torch.cuda.manual_seed(1234 + rank)                   # dropout masks: independent draws per rank, as per sample in one big batch
    stepper = train.AdversarialGraphStep(kind, model, disc, og, od, data, gaw, 10.0, ddp_syn, ddp_disc)
    def cycle():
        stepper.g_step()
        for _ in range(5):
            stepper.d_step()
    cycle()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(cycles):
        cycle()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = stepper.g_step()
    if not all(float(v) == float(v) for v in out):
        raise SystemExit("non-finite loss in the adversarial cycle")
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax[0])
    if check_replicas:
        ok, rows = replica_checksums(list(model.parameters()) + list(disc.parameters()), dev, world)
        if not ok:
            raise SystemExit("bench: %s generator / critic replicas differ across ranks after the adversarial cycles: %r" % (kind, rows))
    return dt / (6 * cycles) * 1e3


def replica_checksums(params, dev, world):
    """(equal, [per-rank checksum pairs]): every rank's (sum, sum of |.|) over all parameters in float64, all-gathered.  Data-parallel
    replicas apply the SAME averaged gradient to the SAME weights, so the checksums must be equal BITWISE on every rank -- a rank that
    missed a bucket's all-reduce, or replayed a graph over stale gradients, shows here."""
    with torch.no_grad():
        flat = torch.cat([p.detach().reshape(-1).double() for p in params])
        mine = torch.stack([flat.sum(), flat.abs().sum()])
    if world == 1:
        return True, [mine.tolist()]
    got = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(got, mine)
    rows = [g.tolist() for g in got]
    return all(r == rows[0] for r in rows), rows


ADV_MASK_SEED = 99


def _adv_build(kind):
    """Generator + critic of the adversarial parity check, on the CPU: seed 1234, generator first (the construction order of
    ``adversarial_cycle_ms``), ``init_weights`` on both."""
    from spoofsv_amd import train
    from spoofsv_amd.critic import linDisc, melDisc
    from spoofsv_amd.tts import SSRN, melSyn
    torch.manual_seed(1234)
    if kind == "text2mel":
        model, disc = melSyn(34, True, 200, 128, 80, 256), melDisc(80, 128)
    else:
        model, disc = SSRN(80, 513, 256), linDisc(513, 128)
    model.apply(train.init_weights); disc.apply(train.init_weights)
    return model, disc


def _adv_masks(kind, batch, n_calls):
    """The keep masks (already scaled by 1 / 0.95, what nn.Dropout multiplies by) of ``n_calls`` critic calls: three per call, at the
    critic's three dropout sites (models/discriminator.py:26,31 and the dropout highwayConv), drawn from one seeded CPU generator."""
    g = torch.Generator().manual_seed(ADV_MASK_SEED)
    T, pool = (T_MEL, 4) if kind == "text2mel" else (4 * T_MEL, 8)
    shapes = [(batch, 128, T), (batch, 128, T), (batch, 64, T // pool)]
    return [(torch.rand(sh, generator=g) >= 0.05).float() / 0.95 for _ in range(n_calls) for sh in shapes]


def adversarial_first_g_iterations(kind, batch, dev):
    """Loss terms (l1, bin-div, [att,] disc) of generator iterations 0 and 1 of the WGAN-GP trainer at the workload's batch, eagerly, with
    the critic's dropout masks INJECTED (``critic.injected_dropout_masks``) so the CPU oracle can apply the same ones: iteration 0 checks
    generator + critic forward, iteration 1 the generator's gradient THROUGH the critic, the adaptive weight and Adam
    (train/adversarial_wasserstein_gp.py:278-297, :329-343)."""
    from spoofsv_amd import critic, train
    model, disc = _adv_build(kind)
    gaw = None
    if kind == "text2mel":
        data = train.synthetic_text2mel_batch(batch, N_TEXT, T_MEL, seed=0, device=dev)
        gaw = train.guided_attention_mat(186, 325, device=dev)
    else:
        data = train.synthetic_ssrn_batch(batch, T_MEL, seed=0, device=dev)
    model.to(dev).train(); disc.to(dev).train()
    og = train.FusedAdam(model.parameters(), 2e-4, (0.5, 0.9), 1e-6, capturable=True)
    od = train.FusedAdam(disc.parameters(), 2e-4, (0.5, 0.9), 1e-6, capturable=True)
    og.refresh_resident_weights()
    stepper = train.AdversarialGraphStep(kind, model, disc, og, od, data, gaw, 10.0, None, None, graph=False)
    masks = [m.to(dev) for m in _adv_masks(kind, batch, 2)]
    out = []
    with critic.injected_dropout_masks(masks):
        for _ in range(2):
            l1, bd, la, ld, _tot = stepper.g_step()
            out.append([float(l1), float(bd)] + ([float(la)] if kind == "text2mel" else []) + [float(ld)])
    del stepper, og, od, model, disc
    torch.cuda.empty_cache()
    return out


def adversarial_first_g_iterations_oracle(kind, batch):
    """The same two generator iterations on the CPU oracles (oracle/tts_oracle.py + oracle/critic_oracle.py, same masks, torch's Adam)."""
    from oracle import critic_oracle as CO
    from oracle import tts_oracle as TO
    from spoofsv_amd import train
    model, disc = _adv_build(kind)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    dsd = {k: v.detach().clone() for k, v in disc.state_dict().items()}
    opt = torch.optim.Adam(list(sd.values()), 2e-4, (0.5, 0.9), 1e-6)
    if kind == "text2mel":
        mel, text, spk = train.synthetic_text2mel_batch(batch, N_TEXT, T_MEL, seed=0)
        gaw = train.guided_attention_mat(186, 325)
    else:
        mel, lin = train.synthetic_ssrn_batch(batch, T_MEL, seed=0)
    masks = _adv_masks(kind, batch, 2)
    out = []
    for it in range(2):
        opt.zero_grad()
        if kind == "text2mel":
            Y, A = TO.melsyn_train(train.shift_right(mel), text, spk, sd)
            terms = list(TO.text2mel_losses(Y, A, mel, gaw))
        else:
            Y = TO.ssrn(mel, sd)
            terms = list(TO.ssrn_losses(Y, lin))
        ld = torch.mean(-CO.critic(Y, dsd, "mel" if kind == "text2mel" else "lin", masks=masks[3 * it:3 * it + 3]))
        base = sum(terms)
        (base + (float(base.detach()) / abs(float(ld.detach()))) * ld).backward()        # adversarial_wasserstein_gp.py:290 / :338
        opt.step()
        out.append([float(t.detach()) for t in terms] + [float(ld.detach())])
    return out


def kernel_roofline(dev):
    """Average duration of the dominant kernel -- the k=3 dilated Conv1d implicit GEMM (gemm_nn_bf3_kernel / gemm_nn_kernel)
    at its most frequent launch shape in the step: highwayConv C=256 (M=2C=512), L=325, B=32 -- timed with HIP events
    on the launch stream, launched the way the training step launches it: weights resident in pre-split form (so the
    timed region holds this kernel only) and every launch on its own activation tensors (20 rotating input/output sets,
    640 MB > the 256 MB Infinity Cache), because in the step each layer reads and writes tensors of its own.
    Algorithmic FLOPs per launch = 2*B*L*(2C)*C*k."""
    import ctypes
    from spoofsv_amd import _lib, ops, resident
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
    f16 = _lib.precision() == 2
    # split-fp16: the operands' scale lists exist before the launch in the training step (the producing LayerNorm kernel
    # writes them), so they are prepared outside the timed region here as well
    amax = lambda ts: [ops.amax_of(t) if f16 else None for t in ts]
    na = ops._AMAX_PIECES if f16 else 0

    def time_conv_fwd(B, C, L, k, nset):
        """ms per launch of the forward conv C -> 2C, cold operands, resident weights."""
        xs = [torch.randn(B, C, L, device=dev) for _ in range(nset)]
        ys = [torch.empty(B, 2 * C, L, device=dev) for _ in range(nset)]
        xa = amax(xs)
        w = torch.randn(2 * C, C, k, device=dev) * 0.05
        bias = torch.randn(2 * C, device=dev)
        rw = resident.ResidentWeights([w])
        rw.refresh(st)
        wp = resident.lookup(w)
        nb = _lib.query("ssv_conv1d_fwd_workspace", C, 2 * C, k)
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        # exactly the launch the training step makes for this layer: with the column statistics of the output the streaming
        # LayerNorm / gate kernel reads (highwayConv forward, C % 64 == 0), outside the exact-fp32 mode
        cs = torch.empty(B * (2 * C // 64) * L * 2, device=dev) if _lib.precision() >= 1 else None
        run = lambda i: _lib.call("ssv_conv1d_fwd", P(xs[i]), C * L, P(xa[i]), na, P(w), wp, P(bias), None, P(ys[i]), 2 * C * L, P(cs), B, C, 2 * C, L, k, 1, 1,
                                  P(ws), nb, st)
        for i in range(nset):
            run(i)
        reps = 3 * nset
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps):
            run(i % nset)
        e1.record()
        torch.cuda.synchronize()
        resident.invalidate([w])
        return e0.elapsed_time(e1) / reps

    def time_conv_dw(B, C, L, k, nset):
        """ms per call of the weight gradient of the same conv (gemm_nt3r_kernel + its slab reduction), cold operands."""
        xs = [torch.randn(B, C, L, device=dev) for _ in range(nset)]
        dys = [torch.randn(B, 2 * C, L, device=dev) for _ in range(nset)]
        xa, dya = amax(xs), amax(dys)
        dw = torch.empty(2 * C, C, k, device=dev)
        nb = _lib.query("ssv_conv1d_bwd_weight_workspace", B, C, 2 * C, L, k)
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        run = lambda i: _lib.call("ssv_conv1d_bwd_weight", P(dys[i]), 2 * C * L, P(dya[i]), na, P(xs[i]), C * L, P(xa[i]), na, P(dw), B, C, 2 * C, L, k, 1, 1,
                                  P(ws), nb, st)
        for i in range(nset):
            run(i)
        reps = 3 * nset
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps):
            run(i % nset)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    def time_conv_dw_multi(B, C, L, k, njobs):
        """ms per LAYER of the weight gradients of `njobs` equal-shaped layers computed by one launch + one reduction launch
        (ssv_conv1d_bwd_weight_multi: what the training step does with its 16 C=256 layers), cold operands."""
        from spoofsv_amd import _lib as L_
        xs = [torch.randn(B, C, L, device=dev) for _ in range(njobs)]
        dys = [torch.randn(B, 2 * C, L, device=dev) for _ in range(njobs)]
        dws = [torch.empty(2 * C, C, k, device=dev) for _ in range(njobs)]
        xa, dya = amax(xs), amax(dys)
        table = (L_.WgradJob * njobs)()
        sh = (ctypes.c_int * 3)()
        L_.call("ssv_conv_shifts", k, 1, 1, sh)
        max_shift = max(abs(v) for v in sh)
        for t, x_, dy_, dw_, xa_, dya_ in zip(table, xs, dys, dws, xa, dya):
            t.dy, t.x, t.dw, t.part, t.pgrads = dy_.data_ptr(), x_.data_ptr(), dw_.data_ptr(), None, None
            t.shift[0], t.shift[1], t.shift[2] = sh[0], sh[1], sh[2]
            if f16:
                t.dy_amax, t.x_amax, t.dy_namax, t.x_namax = dya_.data_ptr(), xa_.data_ptr(), dya_.numel(), xa_.numel()
        tdev = torch.frombuffer(bytearray(bytes(table)), dtype=torch.uint8).to(dev)
        nb = L_.query("ssv_conv1d_bwd_weight_multi_workspace", njobs, B, C, 2 * C, L, k)
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        run = lambda: L_.call("ssv_conv1d_bwd_weight_multi", P(tdev), njobs, 2 * C * L, C * L, B, C, 2 * C, L, k, max_shift, 0, 0, P(ws), nb, st)
        run(); run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            run()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 5 / njobs

    B, C, L, k = 32, 256, 325, 3
    ms = time_conv_fwd(B, C, L, k, 20)
    flops = 2.0 * B * L * (2 * C) * C * k
    bytes_alg = 4.0 * (B * C * L + B * 2 * C * L + 2 * C * C * k + 2 * C)
    ach = flops / (ms * 1e-3) / 1e12
    mode = _lib.precision()                           # 0 exact fp32, 1 split-bf16, 2 split-fp16
    split = mode >= 1
    # both split modes execute 3 sixteen-bit MFMAs (bf16 or fp16: same dense peak) per algorithmic fp32 product: the roof for
    # ALGORITHMIC flops is peak/3
    peak = PEAK_BF16_MFMA_TFLOPS / 3.0 if split else PEAK_F32_MFMA_TFLOPS
    name = {2: "gemm_nn_bf3_kernel<3,1,7,0,1,16> (split-fp16 MFMA: fp16 hi+lo of power-of-two scaled operands, 3 fp16 MFMAs per fp32 product)",
            1: "gemm_nn_bf3_kernel<3,1,7,0,0,16> (split-bf16 MFMA, 3 bf16 MFMAs per fp32 product)", 0: "gemm_nn_kernel<KT=3> (fp32 MFMA)"}[mode]
    tmpl = "1" if mode == 2 else "0"
    # the same measurement for the step's most expensive single launch (SSRN highwayConv C=512, L=1300: 130.9 GFLOP)
    # and for the weight gradient of the headline shape (kernel + slab reduction), so the line shows the range, not one point
    others = []
    for label, fn, (b_, c_, l_) in (("Conv1d fwd C=512->1024 L=1300 (gemm_nn_bf3_kernel<3,2,7,0,%s,16>)" % tmpl, time_conv_fwd, (32, 512, 1300)),
                                     ("Conv1d weight gradient C=256->512 L=325 (gemm_nt3r_kernel<2,%s> + reduce_slabs_perm)" % tmpl, time_conv_dw, (32, 256, 325)),
                                     ("Conv1d weight gradient C=512->1024 L=1300 (gemm_nt3r_kernel<2,%s> + reduce_slabs_perm)" % tmpl, time_conv_dw, (32, 512, 1300))):
        m_ = fn(b_, c_, l_, k, 4 if l_ > 1000 else 20)
        f_ = 2.0 * b_ * l_ * (2 * c_) * c_ * k
        a_ = f_ / (m_ * 1e-3) / 1e12
        others.append({"kernel": label, "us_per_launch": round(m_ * 1e3, 2), "achieved": round(a_, 2), "frac": round(a_ / peak, 4)})
    if split:
        m_ = time_conv_dw_multi(32, 256, 325, 3, 16)
        a_ = flops / (m_ * 1e-3) / 1e12
        others.append({"kernel": "Conv1d weight gradient C=256->512 L=325, 16 layers in one launch (gemm_nt3r_kernel<2> with a job table, "
                                 "2 slabs per layer, + reduce_pair_multi): per layer", "us_per_launch": round(m_ * 1e3, 2), "achieved": round(a_, 2),
                       "frac": round(a_ / peak, 4)})
    lib_ref = None
    if split:
        # reference point, not a baseline of the path: ONE plain bf16 product of the headline shape (operands already bf16 and
        # resident, no split, no bias) through the vendor library (hipBLASLt behind torch.bmm).  The split-bf16 arithmetic needs
        # three such products per fp32 product, so a library-based implementation of the same arithmetic costs >= 3x this time.
        a16 = torch.randn(B, 2 * C, C * k, device=dev).bfloat16()
        x16 = torch.randn(B, C * k, L, device=dev).bfloat16()
        o16 = torch.empty(B, 2 * C, L, device=dev, dtype=torch.bfloat16)
        for _ in range(5):
            torch.bmm(a16, x16, out=o16)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            torch.bmm(a16, x16, out=o16)
        e1.record()
        torch.cuda.synchronize()
        lus = e0.elapsed_time(e1) / 50 * 1e3
        lib_ref = {"what": "torch.bmm (hipBLASLt) bf16 x bf16 -> bf16, same M x K x N x batch, ONE product", "us_per_launch": round(lus, 2),
                   "tflops_one_product": round(flops / lus / 1e6, 1), "us_for_three_products": round(3 * lus, 2),
                   "algorithmic_tflops_if_three_products": round(flops / (3 * lus) / 1e6, 1)}
    return {"bound": "mfma", "kernel": name + ", dilated Conv1d fwd B=32 C=256->512 L=325",
            "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(ach / peak, 4),
            # beside the nominal-peak fraction: against what the matrix pipes sustain under a pure-MFMA load (the split modes only)
            **({"frac_of_sustained": round(ach / (SUSTAINED_F16_MFMA_TFLOPS / 3.0), 4), "sustained_peak": round(SUSTAINED_F16_MFMA_TFLOPS / 3.0, 1),
                "sustained_note": "tools/probe/mfma_rate.hip: 2,070-2,156 TFLOP/s of fp16 MFMAs sustained (0.84 of nominal); / 3 products"} if split else {}),
            "peak_note": ("fp16 dense 2500 TFLOP/s / 3 (three fp16 MFMAs per fp32 product)" if mode == 2 else "bf16 dense 2500 TFLOP/s / 3") if split else "fp32-input MFMA dense",
            "us_per_launch": round(ms * 1e3, 2), "flops_per_launch": flops,
            "hbm_alg_bytes_per_launch": bytes_alg, "hbm_frac_of_8TBs": round(bytes_alg / (ms * 1e-3) / 8e12, 4),
            **pmc_traffic(mode),
            "others": others, "library_reference": lib_ref}


def pmc_traffic(mode):
    """``roofline.traffic``: HBM bytes per launch of the headline kernel from the PMC counters (FETCH_SIZE + WRITE_SIZE, separate
    rocprofv3 --pmc passes, gfx950 corrections of MI355X_MICROARCH.md applied).  Counters cannot be collected by the run that
    prints the line, so the figure is read from ``profiles/traffic.json``, which records the sha256 of the kernel sources it was
    measured on; if the sources have changed since, the figure is withheld (null) instead of going stale silently."""
    import hashlib
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        rec = json.load(open(path))[{0: "fp32", 1: "bf16x3", 2: "f16x2"}[mode]]
    except Exception as e:
        return {"traffic": None, "traffic_source": "no PMC record (%s)" % type(e).__name__}
    h = hashlib.sha256()
    try:
        for f in rec["sources"]:
            h.update(open(os.path.join(ROOT, f), "rb").read())
    except OSError as e:                       # a recorded source no longer exists (renamed / split): the record is stale by definition
        return {"traffic": None, "traffic_source": "stale: %s (%s) since the PMC pass %s" % (os.path.basename(str(e.filename)), type(e).__name__, rec["profile"])}
    if h.hexdigest()[:16] != rec["sources_sha16"]:
        return {"traffic": None, "traffic_source": "stale: %s changed since the PMC pass %s" % (", ".join(rec["sources"]), rec["profile"])}
    return {"traffic": rec["traffic_bytes"], "traffic_corrected": rec.get("traffic_bytes_corrected"), "traffic_source": "%s (sources sha %s); %s" % (rec["profile"], rec["sources_sha16"], rec["note"])}


def cpu_baseline():
    """The CPU oracle (oracle/tts_oracle.py: the stock torch-CPU op sequence the reference itself runs),
    Text2Mel and SSRN train steps at the workload's batch under BASELINE.md section 3's protocol (3 warm-up + 10 timed iterations of each
    model, ~50 s on 16 cores), all host threads torch gives us; then the 8-thread arm on 2 timed iterations."""
    from oracle import tts_oracle as TO
    from spoofsv_amd import train
    from spoofsv_amd.tts import SSRN, melSyn
    torch.manual_seed(1234)
    # use the cores this process may actually run on (the box's share), not every core of the host
    cores = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = max(1, min(cores, int(int(q) / int(per))))
    except Exception:
        pass
    torch.set_num_threads(cores)
    out = {"host_cpus": os.cpu_count(), "affinity": len(os.sched_getaffinity(0))}
    times, times8, first = {}, {}, {}
    for kind, b in (("text2mel", B_PER_GPU), ("ssrn", B_PER_GPU)):
        torch.manual_seed(1234)                    # the weights of Trainer(kind, ...): same seed, same construction order
        if kind == "text2mel":
            m = melSyn(34, True, 200, 128, 80, 256)
            mel, text, spk = train.synthetic_text2mel_batch(b, N_TEXT, T_MEL, seed=0)
            gaw = train.guided_attention_mat(186, 325)
        else:
            m = SSRN(80, 513, 256)
            mel, lin = train.synthetic_ssrn_batch(b, T_MEL, seed=0)
        m.apply(train.init_weights)
        sd = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
        plist = list(sd.values())
        opt = torch.optim.Adam(plist, 2e-4, (0.5, 0.9), 1e-6)

        log = first.setdefault(kind, [])

        def one():
            opt.zero_grad()
            if kind == "text2mel":
                Y, A = TO.melsyn_train(train.shift_right(mel), text, spk, sd)
                ls = TO.text2mel_losses(Y, A, mel, gaw)
            else:
                ls = TO.ssrn_losses(TO.ssrn(mel, sd), lin)
            sum(ls).backward()
            opt.step()
            log.append([float(v.detach()) for v in ls])
        for _ in range(CPU_WARMUP):
            one()
        t0 = time.time()
        reps = CPU_TIMED
        for _ in range(reps):
            one()
        times[kind] = (time.time() - t0) / reps / (b * T_MEL)     # seconds per mel frame
        if cores > 8:
            # SURVEY.md 8d's second arm: the same steps on 8 threads (the survey's own probe ran on 8 cores), 2 timed steps
            torch.set_num_threads(8)
            n_before = len(log)
            t0 = time.time()
            for _ in range(2):
                one()
            times8[kind] = (time.time() - t0) / 2 / (b * T_MEL)
            del log[n_before:]
            torch.set_num_threads(cores)
    fps = 1.0 / (times["text2mel"] + times["ssrn"])
    out.update({"value": round(fps, 1), "unit": "mel-frames/s", "cores": torch.get_num_threads(), "kind": "port",
                "sample": "%d warm-up + %d timed train steps each of Text2Mel and SSRN (BASELINE.md section 3) at the workload's own batch (B=%d), N=186, T=325, fp32, torch CPU ops; "
                          "value_8_threads: 2 timed steps" % (CPU_WARMUP, CPU_TIMED, B_PER_GPU),
                "text2mel_fps": round(1.0 / times["text2mel"], 1), "ssrn_fps": round(1.0 / times["ssrn"], 1),
                "first_losses": {k: v[:2] for k, v in first.items()}})       # popped by main() after the parity check
    if len(times8) == 2:
        out["value_8_threads"] = round(1.0 / (times8["text2mel"] + times8["ssrn"]), 1)
    return out


def stock_gpu_baseline(dev, batch):
    """The yardstick SURVEY.md names ("the stock path the new build must beat"): the reference itself on a ROCm box runs every op through
    torch's stock kernels (MIOpen / rocBLAS / ATen).  Here: the oracle's op sequence (oracle/tts_oracle.py -- nn.functional conv1d,
    layer_norm, softmax ..., autograd, torch.optim.Adam) with tensors on the GPU, fp32, eager, 1 warm-up + 4 timed training iterations
    of each model at the workload's batch.  Baseline leg only: nothing of this runs in the timed region of `value`."""
    from oracle import tts_oracle as TO
    from spoofsv_amd import train
    from spoofsv_amd.tts import SSRN, melSyn
    prev = (torch.backends.cudnn.allow_tf32, torch.backends.cuda.matmul.allow_tf32, torch.backends.cudnn.benchmark)
    torch.backends.cudnn.allow_tf32 = torch.backends.cuda.matmul.allow_tf32 = False
    # torch's default is MIOpen's immediate mode (no solver search), which is what a user of the reference gets; SSV_STOCK_BENCHMARK=1
    # lets MIOpen search (minutes of warm-up) -- DESIGN.md records both
    searched = torch.backends.cudnn.benchmark = os.environ.get("SSV_STOCK_BENCHMARK") == "1"
    ms, losses = {}, {}
    try:
        for kind in ("text2mel", "ssrn"):
            torch.manual_seed(1234)
            if kind == "text2mel":
                m = melSyn(34, True, 200, 128, 80, 256)
                mel, text, spk = train.synthetic_text2mel_batch(batch, N_TEXT, T_MEL, seed=0, device=dev)
                gaw = train.guided_attention_mat(186, 325, device=dev)
            else:
                m = SSRN(80, 513, 256)
                mel, lin = train.synthetic_ssrn_batch(batch, T_MEL, seed=0, device=dev)
            m.apply(train.init_weights)
            sd = {k: v.detach().clone().to(dev).requires_grad_(True) for k, v in m.state_dict().items()}
            opt = torch.optim.Adam(list(sd.values()), 2e-4, (0.5, 0.9), 1e-6)

            def one():
                opt.zero_grad(set_to_none=True)
                if kind == "text2mel":
                    Y, A = TO.melsyn_train(train.shift_right(mel), text, spk, sd)
                    ls = TO.text2mel_losses(Y, A, mel, gaw)
                else:
                    ls = TO.ssrn_losses(TO.ssrn(mel, sd), lin)
                sum(ls).backward()
                opt.step()
                return ls
            ls = one()
            losses[kind] = [float(v.detach()) for v in ls]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                one()
            torch.cuda.synchronize()
            ms[kind] = (time.perf_counter() - t0) / 4 * 1e3
            del sd, opt, m
            torch.cuda.empty_cache()
    finally:
        torch.backends.cudnn.allow_tf32, torch.backends.cuda.matmul.allow_tf32, torch.backends.cudnn.benchmark = prev
    return {"value": round(batch * T_MEL / ((ms["text2mel"] + ms["ssrn"]) * 1e-3), 1), "unit": "mel-frames/s",
            "text2mel_ms": round(ms["text2mel"], 3), "ssrn_ms": round(ms["ssrn"], 3),
            "note": "torch stock ops (MIOpen / rocBLAS / ATen) on the same GPU, fp32, eager, autograd + torch.optim.Adam, 1 warm-up + 4 timed iterations per model, B=%d, cudnn.benchmark=%s" % (batch, searched),
            "first_losses": losses}


def ge2e_config5():
    """BASELINE config 5 (`python bench.py --ge2e`): GE2E d-vector extraction, 88 speakers x 10 utterances x 120 frames x 40 mels --
    LSTM forward + projection + GE2E loss on one MI355X in utterances/s, one full training iteration, and the CPU oracle on
    a bounded sample (its cpu_baseline leg, which is also the parity check of the measured output)."""
    from spoofsv_amd.ge2e import SpeechEmbedder, GE2ELoss
    from oracle import ge2e_oracle as GO
    dev = "cuda:0"
    torch.manual_seed(0)
    m = SpeechEmbedder()
    x = torch.randn(880, 120, 40)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.to(dev).eval()
    L = GE2ELoss(dev)
    xg = x.to(dev)
    def step():
        with torch.no_grad():
            e = m(xg)
            return e, L(e.view(88, 10, 256))
    from spoofsv_amd import ge2e as _ge2e
    for _ in range(2): step()
    torch.cuda.synchronize()
    # a call that has to split the weights first (the first call on a model, or the first after a weight update): timed with the kept workspace dropped
    dt_cold = []
    for _ in range(3):
        _ge2e._FWD_CACHE.clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter(); step(); torch.cuda.synchronize()
        dt_cold.append(time.perf_counter() - t0)
    step(); torch.cuda.synchronize()
    t0 = time.perf_counter(); reps = 5
    for _ in range(reps): e, loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    # one training iteration (GE2E/train_speech_embedder.py:70-86): forward keeping every frame, loss, backward, clip, SGD
    m.train()
    opt = torch.optim.SGD([{"params": m.parameters()}, {"params": L.parameters()}], lr=0.01)
    def train_step():
        opt.zero_grad(set_to_none=True)
        ls = L(m(xg).reshape(88, 10, 256))
        ls.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 3.0)
        torch.nn.utils.clip_grad_norm_(L.parameters(), 1.0)
        opt.step()
        return ls
    train_step(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    for _ in range(3): tl = train_step()
    torch.cuda.synchronize()
    dtt = (time.perf_counter() - t2) / 3
    m.eval()
    # CPU oracle on 44 utterances (bounded), all cores of the box share
    cores = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max": cores = max(1, min(cores, int(int(q) / int(per))))
    except Exception: pass
    torch.set_num_threads(cores)
    xs = x[:44]
    with torch.no_grad():
        t1 = time.perf_counter(); eo = GO.speech_embedder(xs, sd); tc = time.perf_counter() - t1
        lo, _ = GO.ge2e_loss(GO.speech_embedder(x[:40], sd).view(4, 10, 256), torch.tensor(10.0), torch.tensor(-5.0))
    err = float((e[:44].cpu() - eo).abs().max() / eo.abs().max())
    flops = 880 * 120 * 2 * (4 * 768 * (40 + 768) + 2 * 4 * 768 * (768 + 768)) + 880 * 2 * 768 * 256
    return {"metric": "GE2E utterances/s (LSTM fwd + projection + loss)", "value": round(880 / dt, 1), "ms": round(dt * 1e3, 2),
            "tflops": round(flops / dt / 1e12, 1), "loss": round(float(loss), 4), "rel_err_vs_cpu_oracle": err,
            "ms_with_weight_split": round(min(dt_cold) * 1e3, 2),
            "note": "d-vector extraction on fixed weights: the fp16 hi / lo weight planes and their scale are split ONCE and kept between calls (ssv_lstm_fwd_cached; "
                    "rebuilt when a weight's version changes); `ms` is a call on kept planes, `ms_with_weight_split` a call that splits first (the round-5 figure's form)",
            "train_iteration": {"ms": round(dtt * 1e3, 2), "utterances_per_s": round(880 / dtt, 1), "tflops": round(3 * flops / dtt / 1e12, 1),
                                "loss_after": round(float(tl.detach()), 4)},
            "cpu_baseline": {"value": round(44 / tc, 1), "unit": "utterances/s", "cores": cores, "sample": "44 utterances x 120 frames"}}


def launch_ranks(n):
    """``python bench.py --gpus N`` without a launcher: become N ranks.  The parent starts N children of this very script with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set (what ``torch.distributed.run`` would set), relays rank 0's stdout (the ONE JSON
    line), sends the other ranks' stdout to stderr, and exits non-zero as soon as any child does (ending the others by their exact
    PIDs).  It makes NO GPU call itself -- nothing here initialises HIP -- so it is a plain process manager; the reference turns
    multi-GPU on from inside one process (`MULTI_GPU`, train/adversarial_wasserstein_gp.py:183-196), here it is one process per GPU."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if port is None:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"), MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    live = list(procs)
    try:
        while live and rc == 0:
            time.sleep(0.2)
            for p in list(live):
                c = p.poll()
                if c is not None:
                    live.remove(p)
                    if c != 0:
                        rc = c if c > 0 else 1
    except KeyboardInterrupt:           # the launcher is being stopped: take the ranks (exactly these PIDs) down with it
        rc = 130
    # a rank failed: the others would wait in a collective for ever.  They get a few seconds to report an error of their own
    # (the usual case: every rank stops at the same check), then exactly these PIDs are ended.
    grace = time.time() + 10.0
    while live and rc != 130 and time.time() < grace:
        time.sleep(0.2)
        live = [p for p in live if p.poll() is None]
    for p in live:
        p.terminate()
    for p in live:
        try:
            p.wait(timeout=20)
        except subprocess.TimeoutExpired:
            p.kill()
    return rc


class _stdout_on_stderr:
    """The contract is ONE JSON line on stdout; librccl prints a version banner there when its communicator comes up and gloo a
    "connected to N peer ranks" line: while a process group is being created, file descriptor 1 points at stderr."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def launch_check(args):
    """`--launch-check`: everything `--gpus N` does BEFORE the first GPU call, and the exit path after the last one, with gloo and no
    device: rank / world from the environment, local-rank folding onto the visible devices (here: onto one), rendezvous on
    127.0.0.1, one all-gather (each rank's (rank, pid)), rank 0's ONE JSON line, barrier, teardown.  SSV_LAUNCH_CHECK_DIE=<rank>
    makes that rank exit 3 after the rendezvous (the parent must then end the others and return non-zero)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    ndev = max(torch.cuda.device_count(), 1)           # (with amdsmi this only counts; without it torch falls back to hipGetDeviceCount, which runs hipInit -- harmless: no exec follows in the ranks)
    folded = local % ndev
    with _stdout_on_stderr():
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dist.barrier()
    if os.environ.get("SSV_LAUNCH_CHECK_DIE") == str(rank):
        sys.exit(3)
    mine = torch.tensor([rank, os.getpid(), folded], dtype=torch.int64)
    got = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(got, mine)
    dist.barrier()
    if rank == 0:
        ranks = sorted(int(g[0]) for g in got)
        print(json.dumps({"launch_check": True, "n_gpus": world, "ranks": ranks, "distinct_pids": len({int(g[1]) for g in got}),
                          "devices_visible": ndev, "local_ranks_folded_onto": sorted({int(g[2]) for g in got}), "dist_backend": "gloo"}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    args = parse()
    if args.ge2e:
        import spoofsv_amd
        spoofsv_amd.set_precision(args.precision)
        print(json.dumps(dict(ge2e_config5(), arithmetic=args.precision)))
        return
    if args.gpus < 1:
        raise SystemExit("bench: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))
    if args.launch_check:
        return launch_check(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        # the line says "n_gpus": the number of ranks that really ran must be the number that was asked for
        raise SystemExit("bench: --gpus %d but WORLD_SIZE=%d: launch %d ranks (python bench.py --gpus %d starts them itself, or "
                         "python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d)" % (args.gpus, world, args.gpus, args.gpus, args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the HIP hot path has no CPU fallback")
    ndev = torch.cuda.device_count()
    if world > ndev and os.environ.get("SSV_DIST_BACKEND", "nccl") == "nccl":
        raise SystemExit("bench: %d ranks but %d GPU(s) visible: RCCL needs one GPU per rank (SSV_DIST_BACKEND=gloo rehearses the "
                         "launch structure with several ranks on one GPU)" % (world, ndev))
    local = local % max(ndev, 1)          # rehearsal on fewer GPUs than ranks (SSV_DIST_BACKEND=gloo); identity on a full node
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    force_coll = os.environ.get("SSV_FORCE_COLLECTIVES") == "1"     # one rank, every collective really issued (RCCL rehearsal, see train.DataParallelRanks)
    if world > 1 or force_coll:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        backend = os.environ.get("SSV_DIST_BACKEND", "nccl")     # "nccl" is RCCL on ROCm
        # (the communicator is created eagerly -- device_id + a first collective -- while stdout points at stderr)
        with _stdout_on_stderr():
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
            dist.all_reduce(torch.zeros(1, device=dev))
            torch.cuda.synchronize()
    if world > 1 and dist.get_world_size() != args.gpus:
        raise SystemExit("bench: process group of %d ranks for --gpus %d" % (dist.get_world_size(), args.gpus))
    from spoofsv_amd import _lib
    _lib.lib()

    use_graph = not args.no_graph
    import spoofsv_amd
    spoofsv_amd.set_precision(args.precision)
    dtype_note = {"f16x2": "f32 (conv GEMMs: fp32 operands scaled by a power of two and split into fp16 hi+lo = 22 significand bits, three exact "
                           "fp16 MFMAs per product, fp32 accumulate: ~2^-22 per product, gradients within 3e-6 of float64 = the exact-fp32 path's level)",
                  "bf16x3": "f32 (conv GEMMs: split-bf16 hi+lo operands, fp32 accumulate: ~2^-16 per product, NARROWER than the reference's fp32)",
                  "fp32": "f32"}[args.precision]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(trainers, steps, warmup):
        """ms per step of one iteration of EVERY trainer in ``trainers``: barrier + synchronize on both sides, max over ranks."""
        for _ in range(warmup):
            for tr in trainers:
                tr.step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            for tr in trainers:
                tr.step()
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax[0])
        return dt / steps

    t2m = Trainer("text2mel", args.batch, dev, rank, world, use_graph)
    ssr = Trainer("ssrn", args.batch, dev, rank, world, use_graph)
    t2m.prepare()
    ssr.prepare()
    # one "step" of the benchmark: a Text2Mel iteration and an SSRN iteration (independent models), back to back
    per_step = timed([t2m, ssr], args.steps, args.warmup)
    split = {"text2mel": timed([t2m], args.steps, 0), "ssrn": timed([ssr], args.steps, 0)}      # per model, not part of `value`
    frames_per_step = args.batch * T_MEL * world
    loss_t2m, loss_ssrn = t2m.loss, ssr.loss
    replicas_equal = None
    if args.check_replicas:
        for tr in (t2m, ssr):
            ok, rows = replica_checksums(list(tr.model.parameters()), dev, world)
            if not ok:
                raise SystemExit("bench: %s replicas differ across ranks after %d steps: %r" % (tr.kind, args.steps + args.warmup, rows))
        replicas_equal = True
    first = {"text2mel": t2m.first_losses(), "ssrn": ssr.first_losses()}      # iterations 0 and 1, for the oracle check below
    if not (loss_t2m == loss_t2m and loss_ssrn == loss_ssrn):
        raise SystemExit("non-finite loss in the benchmark step")
    ddp_note = "none"
    if t2m.ddp is not None and world == 1:
        ddp_note = ("single-GPU rehearsal of the N > 1 step: gradient arena, backward in %d+%d segments (hipGraphs)%s"
                    % (t2m.ddp.n_buckets, ssr.ddp.n_buckets, ", one RCCL all-reduce per bucket (world size 1) launched between the replays" if t2m.ddp.collectives else ""))
    if world > 1:
        ddp_note = ("gradient arena, backward in %d+%d segments (hipGraphs), one RCCL all-reduce per bucket launched between the replays"
                    % (t2m.ddp.n_buckets, ssr.ddp.n_buckets))
    res = {"metric": "mel-frames/sec (Text2Mel+SSRN train)", "value": round(frames_per_step / per_step, 1), "unit": "mel-frames/s",
           "n_gpus": world, "rccl_ranks": (dist.get_world_size() if dist.is_initialized() and dist.get_backend() == "nccl" else 0),
           "dist_backend": (dist.get_backend() if dist.is_initialized() else "none"), "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(per_step * 1e3, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": dtype_note, "data": "synthetic",
           "config": {"workload": "train_text2mel + train_ssrn (train/ordinary.py step: fwd, l1+bin-div+guided-att losses, bwd, Adam), "
                                  "batch %d utterances/GPU, N=186, T=325, 80 mel -> 513x1300 linear, hidden 256, random-init" % args.batch,
                      "global_batch": args.batch * world, "parallelism": "dp%d" % world,
                      "launch": "hipGraph replay" if use_graph else "eager", "ddp": ddp_note,
                      "text2mel_ms": round(split["text2mel"] * 1e3, 3), "ssrn_ms": round(split["ssrn"] * 1e3, 3),
                      "text2mel_fps": round(frames_per_step / split["text2mel"], 1),
                      "ssrn_fps": round(frames_per_step / split["ssrn"], 1),
                      "final_loss_text2mel": round(loss_t2m, 5), "final_loss_ssrn": round(loss_ssrn, 5)}}
    cfg = res["config"]          # scalars only: the driver's parser keeps flat keys
    if replicas_equal is not None:
        cfg["replica_checksums_equal"] = True       # (a difference is a non-zero exit, never a false on the line); adversarial models are checked below
    del t2m, ssr
    torch.cuda.empty_cache()
    if not args.no_adversarial:
        # BASELINE config 3 (--adversarial; config 4 when N > 1): reported beside the headline, never inside `value`
        a1 = adversarial_cycle_ms("text2mel", args.batch, dev, world, check_replicas=args.check_replicas)
        a2 = adversarial_cycle_ms("ssrn", args.batch, dev, world, check_replicas=args.check_replicas)
        cfg.update({"adversarial_text2mel_ms": round(a1, 3), "adversarial_ssrn_ms": round(a2, 3),
                    "adversarial_text2mel_fps": round(frames_per_step / a1 * 1e3, 1), "adversarial_ssrn_fps": round(frames_per_step / a2 * 1e3, 1),
                    "adversarial_combined_fps": round(frames_per_step / (a1 + a2) * 1e3, 1),
                    "adversarial_note": "ms per iteration averaged over 1 G : 5 D cycles (RATIO=5), WGAN-GP critics on twice-differentiable HIP kernels, "
                                        "hipGraph replay" + ("; data parallel: global adaptive weight, bucketed generator all-reduce, packed critic all-reduce" if world > 1 else "")})
        torch.cuda.empty_cache()
    adv_first = None
    if world == 1 and not args.no_adversarial and not args.no_cpu_baseline and args.batch == B_PER_GPU:
        adv_first = {k: adversarial_first_g_iterations(k, args.batch, dev) for k in ("text2mel", "ssrn")}
    if world == 1 and not args.no_fp32:
        # the same step in the other arithmetic modes, beside the headline and never inside `value`: exact fp32 MFMA
        # (v_mfma_f32_16x16x4_f32: what strict fp32 fma chains cost) and, opt-in for users who accept ~2^-16 products, split-bf16
        for other, key in (("fp32", "fp32_exact"), ("bf16x3", "fast_bf16x3"), ("f16x2", "f16x2")):
            if other == args.precision:
                continue
            spoofsv_amd.set_precision(other)
            f1, f2 = Trainer("text2mel", args.batch, dev, rank, world, use_graph), Trainer("ssrn", args.batch, dev, rank, world, use_graph)
            f1.prepare(); f2.prepare()
            fp = timed([f1, f2], 10, 2)
            cfg.update({key + "_ms_per_step": round(fp * 1e3, 3), key + "_value": round(frames_per_step / fp, 1)})
            del f1, f2
            torch.cuda.empty_cache()
        spoofsv_amd.set_precision(args.precision)
    if world == 1 and not args.no_stock:
        # the same HIP step launched EAGERLY (what a ragged real-data run gets: harness.ordinary_train captures only fixed shapes) ...
        if use_graph:
            e1, e2 = Trainer("text2mel", args.batch, dev, rank, world, False), Trainer("ssrn", args.batch, dev, rank, world, False)
            e1.prepare(); e2.prepare()
            ep = timed([e1, e2], 10, 2)
            cfg.update({"eager_ms_per_step": round(ep * 1e3, 3), "eager_value": round(frames_per_step / ep, 1)})
            del e1, e2
            torch.cuda.empty_cache()
        # ... and the stock-op path on the same GPU
        sg = stock_gpu_baseline(dev, args.batch)
        stock_first = sg.pop("first_losses")
        res["stock_gpu_baseline"] = sg
        cfg["speedup_vs_stock_gpu"] = round(res["value"] / sg["value"], 2)
        for kind in ("text2mel", "ssrn"):       # same seeds, same batch: the stock arm's first losses are the HIP path's (iteration 0)
            e = max(abs(a - b) / max(abs(b), 1e-12) for a, b in zip(first[kind][0], stock_first[kind]))
            cfg["loss_rel_err_vs_stock_gpu_%s_iter0" % kind] = float("%.3g" % e)
    if rank == 0:
        if not args.no_roofline:
            res["roofline"] = kernel_roofline(dev)
        if world == 1 and not args.no_ge2e:
            # BASELINE config 5 on the same line (flat scalars); `python bench.py --ge2e` prints the full record
            g = ge2e_config5()
            cfg.update({"ge2e_utt_per_s": g["value"], "ge2e_ms": g["ms"], "ge2e_tflops": g["tflops"], "ge2e_rel_err_vs_oracle": g["rel_err_vs_cpu_oracle"],
                        "ge2e_roofline_frac": round(g["tflops"] / (PEAK_BF16_MFMA_TFLOPS / 3.0), 4),       # split-fp16 (or split-bf16) LSTM products: 3 MFMAs per product
                        "ge2e_arithmetic": "LSTM products in the %s mode" % args.precision, "ge2e_ms_with_weight_split": g["ms_with_weight_split"],
                        "ge2e_train_iteration_ms": g["train_iteration"]["ms"], "ge2e_cpu_utt_per_s": g["cpu_baseline"]["value"],
                        "ge2e_cpu_cores": g["cpu_baseline"]["cores"]})
        bad = None
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline()
            cfg["speedup_vs_cpu_baseline"] = round(res["value"] / res["cpu_baseline"]["value"], 1)
            # Parity of the configuration that was just timed: the oracle ran the SAME model (seed 1234) on the SAME batch
            # (seed 0 = rank 0's); its loss terms at optimizer iterations 0 (initial weights) and 1 (after one forward +
            # backward + Adam step) against the ones the captured, batched-weight-gradient step produced.
            if args.batch == B_PER_GPU:
                ref = res["cpu_baseline"].pop("first_losses")
                for kind in ("text2mel", "ssrn"):
                    for it in range(2):
                        e = max(abs(a - b) / max(abs(b), 1e-12) for a, b in zip(first[kind][it], ref[kind][it]))
                        cfg["loss_rel_err_vs_oracle_%s_iter%d" % (kind, it)] = float("%.3g" % e)
                        if not e < 1e-4:
                            bad = (kind, it, first[kind][it], ref[kind][it])
            # ... and of the ADVERSARIAL trainer (BASELINE config 3 at its own batch): generator iterations 0 and 1 with the critic's
            # dropout masks injected on both sides -- forward of generator and critic, then the gradient through the critic with the
            # adaptive weight and one Adam step
            if adv_first is not None:
                for kind in ("text2mel", "ssrn"):
                    ref = adversarial_first_g_iterations_oracle(kind, args.batch)
                    worst = 0.0
                    for it in range(2):
                        e = max(abs(a - b) / max(abs(b), 1e-12) for a, b in zip(adv_first[kind][it], ref[it]))
                        cfg["loss_rel_err_vs_oracle_adv_%s_iter%d" % (kind, it)] = float("%.3g" % e)
                        worst = max(worst, e)
                    cfg["loss_rel_err_vs_oracle_adv_%s" % kind] = float("%.3g" % worst)
                    if not worst < 1e-4:
                        bad = ("adversarial " + kind, adv_first[kind], ref)
        print(json.dumps(res), flush=True)
        if bad is not None:
            raise SystemExit("bench: losses differ from the CPU oracle's by more than 1e-4: %r" % (bad,))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
