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
  roofline      the dominant kernel (dilated Conv1d implicit GEMM, fp32 MFMA) timed with HIP events here
  cpu_baseline  the CPU oracle (same stock-op sequence the reference runs) on a bounded sample
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-adversarial", action="store_true", help="skip the extra WGAN-GP cycle timing")
    ap.add_argument("--ge2e", action="store_true", help="measure BASELINE config 5 (GE2E speaker embedder) instead and print its JSON line")
    ap.add_argument("--batch", type=int, default=B_PER_GPU)
    ap.add_argument("--precision", choices=["bf16x3", "fp32"], default="bf16x3",
                    help="conv GEMM arithmetic: split-bf16 MFMA (default, ~1e-5 rel) or exact fp32 MFMA")
    return ap.parse_args()


class Trainer:
    """Holds model + optimizer + static batch; runs fwd+bwd (+Adam) eagerly or from a captured graph."""

    def __init__(self, kind, batch, dev, rank, world, use_graph):
        from spoofsv_amd import train
        from spoofsv_amd.tts import SSRN, melSyn
        self.kind, self.world, self.train = kind, world, train
        torch.manual_seed(1234)
        if kind == "text2mel":
            self.model = melSyn(34, True, 200, textemb_dim=128, freq_bins=80, hidden_dim=256)
            self.batch = train.synthetic_text2mel_batch(batch, N_TEXT, T_MEL, seed=rank, device=dev)
            self.gaw = train.guided_attention_mat(186, 325, device=dev)
        else:
            self.model = SSRN(80, 513, 256)
            self.batch = train.synthetic_ssrn_batch(batch, T_MEL, seed=rank, device=dev)
        self.model.apply(train.init_weights)
        self.model.to(dev).train()
        self.params = [p for p in self.model.parameters()]
        self.opt = train.FusedAdam(self.params, 2e-4, (0.5, 0.9), 1e-6, capturable=True)
        self.ddp = train.DataParallelRanks(self.params) if world > 1 else None
        if self.ddp:
            self.ddp.broadcast_parameters(0)
        self.graph = None
        self.use_graph = use_graph
        self.losses = None

    def _fwd_bwd(self):
        t = self.train
        self.opt.zero_grad(set_to_none=True)
        if self.kind == "text2mel":
            mel, text, spk = self.batch
            pred, att = self.model(t.shift_right(mel), text, spk)
            l1, bd, la = t.text2mel_losses(pred, att, mel, self.gaw)
            loss = l1 + bd + la
        else:
            mel, lin = self.batch
            l1, bd = t.ops.spec_losses(self.model(mel), lin)
            loss = l1 + bd
        loss.backward()
        self.losses = loss

    def _whole(self):
        self._fwd_bwd()
        if self.ddp:
            self.ddp.all_reduce_grads()
        self.opt.step()

    def prepare(self):
        """Warm the allocator on a side stream and capture the step.  Single GPU: forward, backward and
        Adam in ONE graph.  Multi GPU: forward+backward is captured, the gradient all-reduce and Adam
        are launched eagerly after the replay (RCCL collectives stay outside the capture)."""
        if not self.use_graph:
            return
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                self._whole()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            if self.ddp:
                self._fwd_bwd()
                self.graph_grads = [p.grad for p in self.params]      # the tensors every replay writes
            else:
                self._whole()

    def step(self):
        self.step_begin()
        self.step_end()

    # Two halves, so that with N > 1 the gradient all-reduce of one model overlaps the other model's compute:
    #   begin = forward + backward (+ start of the asynchronous all-reduce);  end = wait + Adam.
    def step_begin(self):
        self._pending = None
        if self.ddp is None:
            if self.graph is None:
                self._whole()
            else:
                self.graph.replay()
            return
        if self.graph is None:
            self._fwd_bwd()
            self._pending = self.ddp.all_reduce_grads_begin()
        else:
            self.graph.replay()
            self._pending = self.ddp.all_reduce_grads_begin(self.graph_grads)

    def step_end(self):
        if self.ddp is not None:
            self.ddp.all_reduce_grads_end(self._pending)
            self.opt.step()


def adversarial_cycle_ms(kind, batch, dev, cycles=3):
    """One full WGAN-GP cycle of the reference (1 generator + RATIO=5 critic iterations,
    train/adversarial_wasserstein_gp.py:261-322): generator on the HIP path, critic on the twice-differentiable HIP conv / LayerNorm / gate ops
    with gradient penalty, both iterations replayed from captured hipGraphs (train.AdversarialGraphStep).
    Returns ms per ITERATION averaged over the cycle."""
    from spoofsv_amd import train
    from spoofsv_amd.critic import linDisc, melDisc
    from spoofsv_amd.tts import SSRN, melSyn
    torch.manual_seed(1234)
    gaw = None
    if kind == "text2mel":
        model, disc = melSyn(34, True, 200, 128, 80, 256), melDisc(80, 128)
        data = train.synthetic_text2mel_batch(batch, N_TEXT, T_MEL, seed=0, device=dev)
        gaw = train.guided_attention_mat(186, 325, device=dev)
    else:
        model, disc = SSRN(80, 513, 256), linDisc(513, 128)
        data = train.synthetic_ssrn_batch(batch, T_MEL, seed=0, device=dev)
    model.apply(train.init_weights); disc.apply(train.init_weights)
    model.to(dev).train(); disc.to(dev).train()
    og = train.FusedAdam(model.parameters(), 2e-4, (0.5, 0.9), 1e-6, capturable=True)
    od = train.FusedAdam(disc.parameters(), 2e-4, (0.5, 0.9), 1e-6, capturable=True)
    stepper = train.AdversarialGraphStep(kind, model, disc, og, od, data, gaw)
    def cycle():
        stepper.g_step()
        for _ in range(5):
            stepper.d_step()
    cycle()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(cycles):
        cycle()
    torch.cuda.synchronize()
    out = stepper.g_step()
    if not all(float(v) == float(v) for v in out):
        raise SystemExit("non-finite loss in the adversarial cycle")
    return (time.perf_counter() - t0) / (6 * cycles) * 1e3


def kernel_roofline(dev):
    """Average duration of the dominant kernel -- the k=3 dilated Conv1d implicit GEMM (gemm_nn_bf3_kernel / gemm_nn_kernel)
    at its most frequent launch shape in the step: highwayConv C=256 (M=2C=512), L=325, B=32 -- timed with HIP events
    on the launch stream, launched the way the training step launches it: weights resident in pre-split form (so the
    timed region holds this kernel only) and every launch on its own activation tensors (20 rotating input/output sets,
    640 MB > the 256 MB Infinity Cache), because in the step each layer reads and writes tensors of its own.
    Algorithmic FLOPs per launch = 2*B*L*(2C)*C*k."""
    import ctypes
    from spoofsv_amd import _lib, resident
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())

    def time_conv_fwd(B, C, L, k, nset):
        """ms per launch of the forward conv C -> 2C, cold operands, resident weights."""
        xs = [torch.randn(B, C, L, device=dev) for _ in range(nset)]
        ys = [torch.empty(B, 2 * C, L, device=dev) for _ in range(nset)]
        w = torch.randn(2 * C, C, k, device=dev) * 0.05
        bias = torch.randn(2 * C, device=dev)
        rw = resident.ResidentWeights([w])
        rw.refresh(st)
        wp = resident.lookup(w)
        nb = _lib.query("ssv_conv1d_fwd_workspace", C, 2 * C, k)
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        run = lambda i: _lib.call("ssv_conv1d_fwd", P(xs[i]), C * L, P(w), wp, P(bias), None, P(ys[i]), 2 * C * L, B, C, 2 * C, L, k, 1, 1,
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
        """ms per call of the weight gradient of the same conv (gemm_nt_bf3_kernel + its slab reduction), cold operands."""
        xs = [torch.randn(B, C, L, device=dev) for _ in range(nset)]
        dys = [torch.randn(B, 2 * C, L, device=dev) for _ in range(nset)]
        dw = torch.empty(2 * C, C, k, device=dev)
        nb = _lib.query("ssv_conv1d_bwd_weight_workspace", B, C, 2 * C, k)
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        run = lambda i: _lib.call("ssv_conv1d_bwd_weight", P(dys[i]), 2 * C * L, P(xs[i]), C * L, P(dw), B, C, 2 * C, L, k, 1, 1, P(ws), nb, st)
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

    B, C, L, k = 32, 256, 325, 3
    ms = time_conv_fwd(B, C, L, k, 20)
    flops = 2.0 * B * L * (2 * C) * C * k
    bytes_alg = 4.0 * (B * C * L + B * 2 * C * L + 2 * C * C * k + 2 * C)
    ach = flops / (ms * 1e-3) / 1e12
    split = _lib.lib().ssv_set_precision(1) == 1      # read the mode (set_precision returns the previous one) ...
    _lib.lib().ssv_set_precision(1 if split else 0)   # ... and restore it
    # split-bf16 mode executes 3 bf16 MFMAs per algorithmic fp32 product: the roof for ALGORITHMIC flops is peak/3
    peak = PEAK_BF16_MFMA_TFLOPS / 3.0 if split else PEAK_F32_MFMA_TFLOPS
    name = "gemm_nn_bf3_kernel<3,1,7,0> (split-bf16 MFMA, 3 bf16 MFMAs per fp32 product)" if split else "gemm_nn_kernel<KT=3> (fp32 MFMA)"
    # the same measurement for the step's most expensive single launch (SSRN highwayConv C=512, L=1300: 130.9 GFLOP)
    # and for the weight gradient of the headline shape (kernel + slab reduction), so the line shows the range, not one point
    others = []
    for label, fn, (b_, c_, l_) in (("Conv1d fwd C=512->1024 L=1300 (gemm_nn_bf3_kernel<3,2,7,0>)", time_conv_fwd, (32, 512, 1300)),
                                     ("Conv1d weight gradient C=256->512 L=325 (gemm_nt_bf3_kernel<3,2,4> + reduce_slabs_perm)", time_conv_dw, (32, 256, 325)),
                                     ("Conv1d weight gradient C=512->1024 L=1300 (gemm_nt_bf3_kernel<3,2,4> + reduce_slabs_perm)", time_conv_dw, (32, 512, 1300))):
        m_ = fn(b_, c_, l_, k, 4 if l_ > 1000 else 20)
        f_ = 2.0 * b_ * l_ * (2 * c_) * c_ * k
        a_ = f_ / (m_ * 1e-3) / 1e12
        others.append({"kernel": label, "us_per_launch": round(m_ * 1e3, 2), "achieved": round(a_, 2), "frac": round(a_ / peak, 4)})
    return {"bound": "mfma", "kernel": name + ", dilated Conv1d fwd B=32 C=256->512 L=325",
            "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(ach / peak, 4),
            "peak_note": "bf16 dense 2500 TFLOP/s / 3" if split else "fp32-input MFMA dense",
            "us_per_launch": round(ms * 1e3, 2), "flops_per_launch": flops,
            "hbm_alg_bytes_per_launch": bytes_alg, "hbm_frac_of_8TBs": round(bytes_alg / (ms * 1e-3) / 8e12, 4),
            # PMC: FETCH_SIZE + WRITE_SIZE per launch of this kernel instantiation, separate rocprofv3 --pmc passes (not collected live)
            # gfx950 correction (MI355X_MICROARCH.md, FETCH_SIZE): 16-byte-per-lane loads are tallied at half their bytes -- here only the
            # pre-split weight fragments (2 planes x 512 x 768 bf16 = 1.5 MiB fetched once from HBM), so +768 KiB; the input tile
            # loads are 4-byte-per-lane and counted in full
            "traffic": (12900.7 + 768.0 + 20839.0) * 1024 if split else (45959.9 + 20878.8) * 1024,
            "traffic_source": "profiles/round1_bench_kernel_stats_v5_final.txt (PMC passes; kernel times of the final tree: ..._v6_final.txt)" if split else "profiles/round1_bench_kernel_stats_v1_fp32.txt",
            "others": others}


def cpu_baseline():
    """The CPU oracle (oracle/tts_oracle.py: the stock torch-CPU op sequence the reference itself runs),
    Text2Mel and SSRN train steps on a bounded sample (1 warm-up + 4 timed steps each at the workload's batch, ~10-20 s),
    all host threads torch gives us."""
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
    times = {}
    for kind, b in (("text2mel", B_PER_GPU), ("ssrn", B_PER_GPU)):
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

        def one():
            opt.zero_grad()
            if kind == "text2mel":
                Y, A = TO.melsyn_train(train.shift_right(mel), text, spk, sd)
                l1, bd, la = TO.text2mel_losses(Y, A, mel, gaw)
                (l1 + bd + la).backward()
            else:
                l1, bd = TO.ssrn_losses(TO.ssrn(mel, sd), lin)
                (l1 + bd).backward()
            opt.step()
        one()
        t0 = time.time()
        reps = 4
        for _ in range(reps):
            one()
        times[kind] = (time.time() - t0) / reps / (b * T_MEL)     # seconds per mel frame
    fps = 1.0 / (times["text2mel"] + times["ssrn"])
    out.update({"value": round(fps, 1), "unit": "mel-frames/s", "cores": torch.get_num_threads(), "kind": "port",
                "sample": "1 warm-up + 4 timed train steps each of Text2Mel and SSRN at the workload's own batch (B=%d), N=186, T=325, fp32, torch CPU ops" % B_PER_GPU,
                "text2mel_fps": round(1.0 / times["text2mel"], 1), "ssrn_fps": round(1.0 / times["ssrn"], 1)})
    return out


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
    for _ in range(2): step()
    torch.cuda.synchronize()
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
    print(json.dumps({"metric": "GE2E utterances/s (LSTM fwd + projection + loss)", "value": round(880 / dt, 1), "ms": round(dt * 1e3, 2),
                      "tflops": round(flops / dt / 1e12, 1), "loss": round(float(loss), 4), "rel_err_vs_cpu_oracle": err,
                      "train_iteration": {"ms": round(dtt * 1e3, 2), "utterances_per_s": round(880 / dtt, 1), "tflops": round(3 * flops / dtt / 1e12, 1),
                                          "loss_after": round(float(tl.detach()), 4)},
                      "cpu_baseline": {"value": round(44 / tc, 1), "unit": "utterances/s", "cores": cores, "sample": "44 utterances x 120 frames"}}))

def main():
    args = parse()
    if args.ge2e:
        return ge2e_config5()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the HIP hot path has no CPU fallback")
    ndev = torch.cuda.device_count()
    local = local % max(ndev, 1)          # rehearsal on fewer GPUs than ranks (SSV_DIST_BACKEND=gloo); identity on a full node
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("SSV_DIST_BACKEND", "nccl")     # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    from spoofsv_amd import _lib
    _lib.lib()

    use_graph = not args.no_graph
    import spoofsv_amd
    spoofsv_amd.set_precision(args.precision)
    use_bf3_mode = args.precision == "bf16x3"
    t2m = Trainer("text2mel", args.batch, dev, rank, world, use_graph)
    ssr = Trainer("ssrn", args.batch, dev, rank, world, use_graph)
    t2m.prepare()
    ssr.prepare()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Leave each model's gradient all-reduce in flight while the other model computes (RCCL runs collectives on its own
    # stream).  Only with the RCCL backend: gloo's CUDA path synchronises with the host and the overlap backfires.
    pipeline = world > 1 and os.environ.get("SSV_DIST_BACKEND", "nccl") == "nccl" and os.environ.get("SSV_DDP_PIPELINE", "1") != "0"

    def both():
        # one "step" of the benchmark: a Text2Mel iteration and an SSRN iteration (independent models)
        if pipeline:
            t2m.step_begin()
            ssr.step_begin()
            t2m.step_end()
            ssr.step_end()
        else:
            t2m.step()
            ssr.step()

    for _ in range(args.warmup):
        both()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        both()
    barrier()
    dt = time.perf_counter() - t0
    # per-model split (same number of steps, timed separately, not part of `value`)
    split = {}
    for name, tr in (("text2mel", t2m), ("ssrn", ssr)):
        barrier()
        s0 = time.perf_counter()
        for _ in range(args.steps):
            tr.step()
        barrier()
        split[name] = (time.perf_counter() - s0) / args.steps
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax[0])
    frames = args.batch * T_MEL * world * args.steps
    loss_t2m, loss_ssrn = float(t2m.losses.detach()), float(ssr.losses.detach())
    if not (loss_t2m == loss_t2m and loss_ssrn == loss_ssrn):
        raise SystemExit("non-finite loss in the benchmark step")

    if rank == 0:
        res = {"metric": "mel-frames/sec (Text2Mel+SSRN train)", "value": round(frames / dt, 1), "unit": "mel-frames/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32 (conv GEMMs: split-bf16 hi+lo operands, fp32 accumulate)" if use_bf3_mode else "f32", "data": "synthetic",
               "config": {"workload": "train_text2mel + train_ssrn (train/ordinary.py step: fwd, l1+bin-div+guided-att losses, bwd, Adam), "
                                      "batch %d utterances/GPU, N=186, T=325, 80 mel -> 513x1300 linear, hidden 256, random-init" % args.batch,
                          "global_batch": args.batch * world, "parallelism": "dp%d" % world,
                          "launch": "hipGraph replay" if use_graph else "eager",
                          "ddp": ("flat-bucket RCCL all-reduce, pipelined under the other model's step" if pipeline else
                                  ("flat-bucket all-reduce" if world > 1 else "none")),
                          "text2mel_ms": round(split["text2mel"] * 1e3, 3), "ssrn_ms": round(split["ssrn"] * 1e3, 3),
                          "text2mel_fps": round(args.batch * T_MEL * world / split["text2mel"], 1),
                          "ssrn_fps": round(args.batch * T_MEL * world / split["ssrn"], 1),
                          "final_loss": [round(loss_t2m, 5), round(loss_ssrn, 5)]}}
        res["roofline"] = kernel_roofline(dev)
        if world == 1 and not args.no_adversarial:
            # BASELINE config 3 (--adversarial): reported beside the headline, never inside `value`
            a1, a2 = adversarial_cycle_ms("text2mel", args.batch, dev), adversarial_cycle_ms("ssrn", args.batch, dev)
            res["config"]["adversarial"] = {"text2mel_ms_per_iter": round(a1, 3), "ssrn_ms_per_iter": round(a2, 3),
                                            "text2mel_fps": round(args.batch * T_MEL / a1 * 1e3, 1), "ssrn_fps": round(args.batch * T_MEL / a2 * 1e3, 1),
                                            "combined_fps": round(args.batch * T_MEL / (a1 + a2) * 1e3, 1),
                                            "note": "1 G : 5 D cycle average, critic convs / LayerNorms / highway gate on twice-differentiable HIP kernels (dropout, leaky-ReLU, pooling: torch), both iterations replayed from hipGraphs"}
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline()
            res["config"]["speedup_vs_cpu_baseline"] = round(res["value"] / res["cpu_baseline"]["value"], 1)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
