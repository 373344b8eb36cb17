"""GPU tests of the data-parallel paths (BASELINE config 4): gradient arena, backward in segments with one
all-reduce per gradient bucket, captured as several hipGraphs with the collectives between the replays; the adversarial
(WGAN-GP) iterations under the same scheme.  The test box has ONE GPU: two ranks share it and talk over gloo; the driver's
multi-GPU runs use RCCL through the same code."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _melsyn(seed=100, hidden=32, temb=16):
    from spoofsv_amd import train
    from spoofsv_amd.tts import melSyn
    torch.manual_seed(seed)
    m = melSyn(34, True, 200, textemb_dim=temb, freq_bins=80, hidden_dim=hidden)
    m.apply(train.init_weights)
    return m.to(DEV).train()


def _ssrn(seed=101):
    from spoofsv_amd import train
    from spoofsv_amd.tts import SSRN
    torch.manual_seed(seed)
    m = SSRN(80, 65, 32)
    m.apply(train.init_weights)
    return m.to(DEV).train()


def _grads(m):
    torch.cuda.synchronize()
    return {k: p.grad.detach().cpu().numpy().copy() for k, p in m.named_parameters()}


@pytest.mark.parametrize("kind", ["text2mel", "ssrn"])
@pytest.mark.parametrize("graph", [False, True])
def test_segmented_arena_step_is_bit_identical_to_the_plain_step(kind, graph):
    """One rank: the step with the gradient arena and the segmented backward (tts.ddp_plan: 5 / 3 segments, as separate
    hipGraphs when captured) runs exactly the kernels of the plain step, so gradients and post-step weights are bit-equal;
    every gradient lives in the arena, and the forward's cut hook is gone after the step."""
    from spoofsv_amd import train
    if kind == "text2mel":
        a, b = _melsyn(), _melsyn()
        batch = list(train.synthetic_text2mel_batch(4, N=40, T=64, seed=3, device=DEV))       # B*T = 256: split-bf16 GEMMs + resident planes
        gaw = train.guided_attention_mat(40, 64, device=DEV)
    else:
        a, b = _ssrn(), _ssrn()
        batch = list(train.synthetic_ssrn_batch(4, T=40, out_bins=65, seed=3, device=DEV))
        gaw = None
    oa = train.FusedAdam(a.parameters(), 2e-4, (0.5, 0.9), 1e-6, capturable=graph)
    ob = train.FusedAdam(b.parameters(), 2e-4, (0.5, 0.9), 1e-6, capturable=graph)
    oa.refresh_resident_weights(); ob.refresh_resident_weights()
    plain = train.TrainStep(kind, a, oa, batch, gaw, None, graph=False)
    ddp = train.DataParallelRanks(model=b)
    assert ddp.n_buckets == (5 if kind == "text2mel" else 3) and ddp.world == 1
    seg = train.TrainStep(kind, b, ob, batch, gaw, ddp, graph=graph).prepare()
    if graph:       # capture ran warm-up iterations on b: start both from the same state again
        b.load_state_dict(a.state_dict())
        for st in ob.state.values():
            st["exp_avg"].zero_(); st["exp_avg_sq"].zero_()
        ob._step_dev.zero_()
        ob.refresh_resident_weights()
        assert len(seg.stepper._graphs) == ddp.n_buckets + 1               # fwd+seg0 | seg1 | ... | Adam
    for it in range(2):
        la, lb = plain(), seg()
        ga, gb = _grads(a), _grads(b)
        for k in ga:
            assert np.array_equal(ga[k], gb[k]), (it, k)
        for x, y in zip(la, lb):
            assert float(x) == float(y)
    for (k, p), q in zip(a.state_dict().items(), b.state_dict().values()):
        assert torch.equal(p, q), k
    for p in b.parameters():
        assert p.grad.data_ptr() == ddp.arena.slot(p).data_ptr()
    from spoofsv_amd import tts
    assert b._cut is tts._no_cut


def _t2m_rank(rank, world, port, q, graph):
    import torch.distributed as dist
    from spoofsv_amd import train
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = _melsyn(100 + rank)                              # replicas start different; broadcast_parameters must fix that
    ddp = train.DataParallelRanks(model=m)
    ddp.broadcast_parameters(0)
    opt = train.FusedAdam(m.parameters(), 2e-4, (0.5, 0.9), 1e-6, capturable=graph)
    opt.refresh_resident_weights()
    mel, text, spk = train.synthetic_text2mel_batch(8, N=40, T=64, seed=3, device=DEV)
    gaw = train.guided_attention_mat(40, 64, device=DEV)
    sl = slice(4 * rank, 4 * rank + 4)                  # each rank takes its half of the global batch
    w0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    step = train.TrainStep("text2mel", m, opt, [mel[sl], text[sl], spk[sl]], gaw, ddp, graph=graph).prepare()
    if graph:
        m.load_state_dict(w0)
        for st in opt.state.values():
            st["exp_avg"].zero_(); st["exp_avg_sq"].zero_()
        opt._step_dev.zero_()
        opt.refresh_resident_weights()
    losses = [float(v) for v in step()]
    grads = _grads(m)
    step()
    torch.cuda.synchronize()
    q.put((rank, grads, losses, {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}))
    dist.barrier()
    dist.destroy_process_group()


def _spawn(target, args, world=2):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=target, args=(r, world, port, q) + tuple(args)) for r in range(world)]
    import queue
    import time
    for p in procs:
        p.start()
    res, t0 = [], time.time()
    while len(res) < world:
        try:
            res.append(q.get(timeout=2))
        except queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead or time.time() - t0 > 280:                  # a rank died (its traceback is in the captured stderr) or hangs
                for p in procs:
                    if p.is_alive():
                        p.kill()
                raise AssertionError("rank process failed: exit codes %s after %.0f s" % ([p.exitcode for p in procs], time.time() - t0))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return sorted(res, key=lambda r: r[0])


@pytest.mark.parametrize("graph", [False, True])
def test_two_rank_segmented_step_equals_single_process_on_the_global_batch(graph):
    """Two ranks, half the global batch each, backward in five segments with the bucket all-reduces between them (between the
    hipGraph replays when captured): the averaged gradient is the one a single process computes on the whole batch, and the
    replicas stay bit-identical through optimizer steps."""
    from spoofsv_amd import train
    m = _melsyn(100)
    mel, text, spk = train.synthetic_text2mel_batch(8, N=40, T=64, seed=3, device=DEV)
    gaw = train.guided_attention_mat(40, 64, device=DEV)
    opt = train.FusedAdam(m.parameters(), 2e-4, (0.5, 0.9), 1e-6)
    want_loss = [float(v) for v in train.TrainStep("text2mel", m, opt, [mel, text, spk], gaw, None)()]
    want = _grads(m)
    res = _spawn(_t2m_rank, (graph,))
    for k in res[0][3]:
        assert np.array_equal(res[0][3][k], res[1][3][k]), k           # replicas bit-identical after two steps
    scale = max(float(np.abs(w).max()) for w in want.values())
    for k, w in want.items():
        assert np.array_equal(res[0][1][k], res[1][1][k]), k
        # split-bf16 products over a different batch partition: agreement to a few 1e-4 of the gradient's own size
        assert float(np.abs(res[0][1][k] - w).max()) <= 2e-5 * scale + 5e-4 * float(np.abs(w).max()), k
    # per-rank losses average to the global-batch loss
    mean_loss = [(a + b) / 2 for a, b in zip(res[0][2], res[1][2])]
    for a, b in zip(mean_loss, want_loss):
        assert abs(a - b) < 1e-5 * max(1.0, abs(b))


# ---- adversarial (WGAN-GP) iterations, data parallel: BASELINE config 4 ----------------------------------------------------
def _adv_build(kind, seed=21):
    from spoofsv_amd import train
    from spoofsv_amd.critic import linDisc, melDisc
    torch.manual_seed(seed)
    if kind == "text2mel":
        m = _melsyn(seed)
        d = melDisc(80, 16)
    else:
        m = _ssrn(seed)
        d = linDisc(65, 16)
    d.apply(train.init_weights)
    return m, d.to(DEV).eval()                           # eval: no dropout RNG, so the two partitions see the same critic


def _adv_batch(kind, B):
    from spoofsv_amd import train
    if kind == "text2mel":
        return list(train.synthetic_text2mel_batch(B, N=40, T=64, seed=4, device=DEV)), train.guided_attention_mat(40, 64, device=DEV)
    return list(train.synthetic_ssrn_batch(B, T=40, out_bins=65, seed=4, device=DEV)), None


def _adv_run(kind, m, d, batch, gaw, ddp_syn, ddp_disc, graph):
    """One G and one D iteration from the given weights; returns losses and the gradients each iteration produced."""
    from spoofsv_amd import train
    og = train.FusedAdam(m.parameters(), 2e-4, (0.5, 0.9), 1e-6, capturable=True)
    od = train.FusedAdam(d.parameters(), 2e-4, (0.5, 0.9), 1e-6, capturable=True)
    w_m = {k: v.detach().clone() for k, v in m.state_dict().items()}
    w_d = {k: v.detach().clone() for k, v in d.state_dict().items()}
    st = train.AdversarialGraphStep(kind, m, d, og, od, batch, gaw, 10.0, ddp_syn, ddp_disc, graph=graph, coeff_seed=9)
    if graph:
        m.load_state_dict(w_m); d.load_state_dict(w_d)
        for o in (og, od):
            for s in o.state.values():
                s["exp_avg"].zero_(); s["exp_avg_sq"].zero_()
            o._step_dev.zero_()
            o.refresh_resident_weights()
    g_out = [float(v) for v in st.g_step()]
    torch.cuda.synchronize()
    if ddp_syn is not None:       # replayed graphs do not touch the Python-side .grad attributes: the gradients live in the arena
        g_grads = {k: ddp_syn.arena.slot(p).detach().cpu().numpy().copy() for k, p in m.named_parameters()}
    else:
        g_grads = _grads(m)
    d_out = [float(v) for v in st.d_step()]
    torch.cuda.synchronize()
    d_grads = {k: p.grad.detach().cpu().numpy().copy() for k, p in d.named_parameters()}
    st.g_step(); st.d_step()
    torch.cuda.synchronize()
    weights = {k: v.detach().cpu().numpy() for k, v in list(m.state_dict().items()) + [("disc." + k, v) for k, v in d.state_dict().items()]}
    return g_out, g_grads, d_out, d_grads, weights


def _adv_rank(rank, world, port, q, kind, graph):
    import torch.distributed as dist
    from spoofsv_amd import train
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m, d = _adv_build(kind)
    batch, gaw = _adv_batch(kind, 8)
    sl = slice(4 * rank, 4 * rank + 4)
    ddp_syn = train.DataParallelRanks(model=m)
    ddp_disc = train.DataParallelRanks(list(d.parameters()))
    q.put((rank,) + _adv_run(kind, m, d, [b[sl] for b in batch], gaw, ddp_syn, ddp_disc, graph))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind,graph", [("text2mel", True), ("ssrn", True), ("text2mel", False)])
def test_two_rank_adversarial_iterations_equal_single_process_on_the_global_batch(kind, graph):
    """BASELINE config 4 (train/adversarial_wasserstein_gp.py:183-196,261-322 under data parallelism): two ranks with half
    the global batch each.  G iteration: the adaptive critic weight is formed from the all-reduced (global) loss terms and the
    generator gradient buckets are all-reduced between the backward segments; D iteration: penalty coefficients drawn per
    GLOBAL sample index, critic gradients and both loss scalars in one packed all-reduce.  Losses and gradients must equal
    the single-process iteration on the whole batch; replicas stay bit-identical."""
    m, d = _adv_build(kind)
    batch, gaw = _adv_batch(kind, 8)
    g_out, g_grads, d_out, d_grads, _ = _adv_run(kind, m, d, batch, gaw, None, None, graph=False)
    res = _spawn(_adv_rank, (kind, graph))
    for k in res[0][5]:
        assert np.array_equal(res[0][5][k], res[1][5][k]), k           # replicas (generator and critic) bit-identical after 2 G + 2 D
    for r in res:
        for a, b in zip(r[1], g_out):                                  # (l1, bd, att, disc, total): global-batch values on every rank
            assert abs(a - b) < 2e-5 * max(1.0, abs(b)), (r[1], g_out)
        for a, b in zip(r[3], d_out):                                  # (loss_D, loss_gp)
            assert abs(a - b) < 2e-4 * max(1.0, abs(b)), (r[3], d_out)
    scale = max(float(np.abs(w).max()) for w in g_grads.values())
    for k, w in g_grads.items():
        assert np.array_equal(res[0][2][k], res[1][2][k]), k
        assert float(np.abs(res[0][2][k] - w).max()) <= 2e-5 * scale + 5e-4 * float(np.abs(w).max()), k
    for k, w in d_grads.items():
        if k in ("conv1.bias", "conv2.bias", "conv3.bias", "conv4.bias", "conv5.bias", "hc.conv.bias"):
            continue                                                    # exactly-zero true gradients: rounding noise on both sides
        a = res[0][4][k]
        assert np.array_equal(a, res[1][4][k]), k
        assert float(np.linalg.norm(a - w)) <= 3e-3 * max(1e-6, float(np.linalg.norm(w))), k


# ---- the trainers under torchrun-style environments -------------------------------------------------------------------------
def _harness_rank(rank, world, port, q, tmp, adversarial, capture):
    import json
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      SSV_DIST_BACKEND="gloo")
    from spoofsv_amd import harness
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = json.load(open(os.path.join(root, "config.json")))
    cfg.update(SRC_ROOT_DIR=tmp + os.sep, BATCH_SIZE=4, MAX_TEXT_LEN=24, MAX_FRAME_NUM=40, HIDDEN_DIM=32, TEXT_EMB_DIM=16, SSRN_DIM=32,
               DISC_DIM=16, VAL_EVERY_ITER=2, SYNTHETIC_BATCHES_PER_EPOCH=3, MAX_ITERATIONS=5, RATIO=1, MULTI_GPU=True,
               CAPTURE_GRAPHS=capture)
    cfg["STFT"] = {"FFT_LENGTH": 128, "HOP_LENGTH": 32}
    torch.manual_seed(1000 + rank)                      # ranks initialise differently; the trainer must broadcast rank 0's weights
    if adversarial:
        model, disc, logs = harness.adversarial_train("train_text2mel", "conditional", cfg, current_time="ddp")
        extra = [v.detach().cpu().numpy() for v in disc.state_dict().values()]
        hist = logs["loss_train_log_syn"] + logs["loss_train_log_disc"]
    else:
        model, hist = harness.ordinary_train("train_text2mel", "conditional", cfg, current_time="ddp")
        extra = []
    torch.cuda.synchronize()
    q.put((rank, [v.detach().cpu().numpy() for v in model.state_dict().values()] + extra, hist))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("adversarial,capture", [(False, False), (True, False), (True, True)])
def test_trainers_run_data_parallel_under_torchrun_environment(tmp_path, adversarial, capture):
    """main.py's trainers launched one process per rank (RANK / WORLD_SIZE / LOCAL_RANK as torchrun exports them): they build
    the process group themselves, shard the batches, keep the replicas identical (weights bit-equal on both ranks after 5
    iterations, validation and checkpoint included), report the global-batch loss on every rank, and only rank 0 writes
    checkpoints."""
    res = _spawn(_harness_rank, (str(tmp_path), adversarial, capture))
    for a, b in zip(res[0][1], res[1][1]):
        assert np.array_equal(a, b)
    assert res[0][2] == res[1][2] and all(h == h for h in res[0][2]) and len(res[0][2]) == 5
    sub = "adversarial" if adversarial else "not_adversarial"
    cks = os.listdir(os.path.join(str(tmp_path), "checkpoints", "conditional", sub, "ddp"))
    assert "text2mel_iteration_3.tar.pth" in cks and "text2mel_best_model.tar.pth" in cks


@pytest.mark.parametrize("kind", ["text2mel", "ssrn"])
@pytest.mark.parametrize("graph,seg", [(False, False), (True, False), (True, True)])
def test_deferred_batched_weight_gradients_match_the_immediate_step(kind, graph, seg):
    """ops.DeferredWgrad: backward runs only the LayerNorm / gate backward and the data gradients, the weight gradients of all
    equal-shaped layers of a segment are computed by one launch each at the segment's end (fewer, longer slabs).  Same
    products; only the fp32 summation order over the batch differs: losses identical, every gradient within 2e-6 of its
    tensor's largest entry, weights after two steps within 1e-6."""
    from spoofsv_amd import train
    if kind == "text2mel":
        a, b = _melsyn(), _melsyn()
        batch = list(train.synthetic_text2mel_batch(8, N=40, T=64, seed=3, device=DEV))      # B*T = 512: the split-bf16 kernels, 4 slabs
        gaw = train.guided_attention_mat(40, 64, device=DEV)
    else:
        a, b = _ssrn(), _ssrn()
        batch = list(train.synthetic_ssrn_batch(8, T=40, out_bins=65, seed=3, device=DEV))
        gaw = None
    oa = train.FusedAdam(a.parameters(), 2e-4, (0.5, 0.9), 1e-6, capturable=graph)
    ob = train.FusedAdam(b.parameters(), 2e-4, (0.5, 0.9), 1e-6, capturable=graph)
    oa.refresh_resident_weights(); ob.refresh_resident_weights()
    plain = train.TrainStep(kind, a, oa, batch, gaw, None, graph=False)
    ddp = train.DataParallelRanks(model=b) if seg else None
    deferred = train.TrainStep(kind, b, ob, batch, gaw, ddp, graph=graph, defer_wgrad=True).prepare()
    if graph:
        b.load_state_dict(a.state_dict())
        for st in ob.state.values():
            st["exp_avg"].zero_(); st["exp_avg_sq"].zero_()
        ob._step_dev.zero_()
        ob.refresh_resident_weights()
    for it in range(2):
        la, lb = plain(), deferred()
        for x, y in zip(la, lb):
            assert float(x) == float(y)
        if it == 0:
            torch.cuda.synchronize()
            for (k, p), q in zip(a.named_parameters(), b.parameters()):
                g = ddp.arena.slot(q) if seg else q.grad
                assert g is not None and float((p.grad - g).abs().max()) <= 2e-6 * float(p.grad.abs().max()) + 1e-12, k
    for (k, p), q in zip(a.state_dict().items(), b.state_dict().values()):
        assert float((p - q).abs().max()) < 1e-6, k


# ---- RCCL itself, on the one GPU there is: a process group of ONE rank with every collective really issued ------------------
def _rccl_world1(rank, world, port, q, what):
    """Child process: the process group is built BEFORE the first GPU call (device_count() does not initialise HIP), with the
    device it serves, as bench.py and harness._distributed do."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SSV_FORCE_COLLECTIVES="1")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from spoofsv_amd import train
    issued = {"all_reduce": 0, "broadcast": 0}
    real_ar, real_bc = dist.all_reduce, dist.broadcast

    def count_ar(*a, **k):
        issued["all_reduce"] += 1
        return real_ar(*a, **k)

    def count_bc(*a, **k):
        issued["broadcast"] += 1
        return real_bc(*a, **k)
    dist.all_reduce, dist.broadcast = count_ar, count_bc
    out = {}
    if what == "train":
        for kind in ("text2mel", "ssrn"):
            if kind == "text2mel":
                a, b = _melsyn(), _melsyn()
                batch = list(train.synthetic_text2mel_batch(4, N=40, T=64, seed=3, device=DEV))
                gaw = train.guided_attention_mat(40, 64, device=DEV)
            else:
                a, b = _ssrn(), _ssrn()
                batch = list(train.synthetic_ssrn_batch(4, T=40, out_bins=65, seed=3, device=DEV))
                gaw = None
            oa = train.FusedAdam(a.parameters(), 2e-4, (0.5, 0.9), 1e-6, capturable=True)
            ob = train.FusedAdam(b.parameters(), 2e-4, (0.5, 0.9), 1e-6, capturable=True)
            oa.refresh_resident_weights(); ob.refresh_resident_weights()
            plain = train.TrainStep(kind, a, oa, batch, gaw, None, graph=False)
            ddp = train.DataParallelRanks(model=b)
            assert ddp.world == 1 and ddp.collectives
            ddp.broadcast_parameters(0)
            ob.refresh_resident_weights()
            seg = train.TrainStep(kind, b, ob, batch, gaw, ddp, graph=True, defer_wgrad=True).prepare()
            b.load_state_dict(a.state_dict())
            for st in ob.state.values():
                st["exp_avg"].zero_(); st["exp_avg_sq"].zero_()
            ob._step_dev.zero_()
            ob.refresh_resident_weights()
            n0 = issued["all_reduce"]
            for it in range(3):
                la, lb = plain(), seg()
                assert [float(x) for x in la] == [float(y) for y in lb], (kind, it)
            torch.cuda.synchronize()
            out[kind] = dict(per_step=(issued["all_reduce"] - n0) / 3, buckets=ddp.n_buckets,
                             worst=max(float((p - q_).abs().max()) for p, q_ in zip(a.state_dict().values(), b.state_dict().values())))
    else:
        for kind in ("text2mel", "ssrn"):
            m0, d0 = _adv_build(kind)
            m1, d1 = _adv_build(kind)
            batch, gaw = _adv_batch(kind, 4)
            ref = _adv_run(kind, m0, d0, batch, gaw, None, None, graph=False)
            ddp_syn, ddp_disc = train.DataParallelRanks(model=m1), train.DataParallelRanks(list(d1.parameters()))
            n0 = issued["all_reduce"]
            got = _adv_run(kind, m1, d1, batch, gaw, ddp_syn, ddp_disc, graph=True)
            out[kind] = dict(all_reduces=issued["all_reduce"] - n0, g_out=(ref[0], got[0]), d_out=(ref[2], got[2]),
                             worst=max(float(np.abs(ref[4][k] - got[4][k]).max()) for k in ref[4]))
    q.put((0, out, issued))
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_world_size_one_train_step_with_every_collective_issued():
    """The launch structure of the N > 1 step on the hardware there is: process group "nccl" (= RCCL) of one rank,
    SSV_FORCE_COLLECTIVES=1, so the parameter broadcast and EVERY bucket's asynchronous all-reduce really run through RCCL --
    between the replays of the backward segments' hipGraphs, on RCCL's own stream behind an event, Adam's graph waiting for
    the last one.  A sum over one rank changes nothing, so losses must equal the plain eager step's exactly, step after step
    (the batched weight gradients differ from the immediate ones in summation order only: weights within 1e-6).
    train/ordinary.py:165-173 is what this replaces."""
    (_, out, issued), = _spawn(_rccl_world1, ("train",), world=1)
    assert issued["broadcast"] > 0
    for kind, buckets in (("text2mel", 5), ("ssrn", 3)):
        assert out[kind]["buckets"] == buckets and out[kind]["per_step"] == buckets, out[kind]      # one all-reduce per bucket per replayed step
        assert out[kind]["worst"] < 1e-6, out[kind]


def test_rccl_world_size_one_adversarial_iterations_with_every_collective_issued():
    """Same for the WGAN-GP iterations (train/adversarial_wasserstein_gp.py:183-196,290,300): the all-reduce of the four loss
    scalars between the forward graph and the backward graphs of a G iteration, the generator's bucket all-reduces, and the
    critic's packed all-reduce between the two graphs of a D iteration, all through RCCL; results equal to the captured
    single-process iterations."""
    (_, out, issued), = _spawn(_rccl_world1, ("adv",), world=1)
    for kind, buckets in (("text2mel", 5), ("ssrn", 3)):
        r = out[kind]
        assert r["all_reduces"] >= 2 * (buckets + 2), r          # 2 x (scalars + buckets) for G, 2 x packed for D, plus the warm-up iterations
        for a, b in zip(r["g_out"][0], r["g_out"][1]):
            assert abs(a - b) <= 1e-6 * max(1.0, abs(a)), r["g_out"]
        for a, b in zip(r["d_out"][0], r["d_out"][1]):
            assert abs(a - b) <= 1e-5 * max(1.0, abs(a)), r["d_out"]
        assert r["worst"] < 1e-5, r


def test_bench_gpus_2_starts_two_ranks_by_itself_and_reports_them():
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE: the parent (no GPU call) starts two ranks, here sharing the
    one GPU over gloo (SSV_DIST_BACKEND=gloo); rank 0's single JSON line says n_gpus = 2 and counts both ranks' frames.  With RCCL
    the same command needs two GPUs and says so.  train/adversarial_wasserstein_gp.py:183-196 (MULTI_GPU) is what this replaces."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    args = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4",
            "--no-cpu-baseline", "--no-adversarial", "--no-fp32", "--no-ge2e", "--no-roofline"]
    r = subprocess.run(args, env=dict(env, SSV_DIST_BACKEND="gloo"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 8 and rec["config"]["parallelism"] == "dp2"
    assert rec["dist_backend"] == "gloo" and rec["rccl_ranks"] == 0
    assert abs(rec["value"] - 2 * 4 * 325 / (rec["ms_per_step"] * 1e-3)) < 1e-3 * rec["value"]
    if torch.cuda.device_count() < 2:
        r = subprocess.run(args, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode != 0 and "one GPU per rank" in r.stderr and r.stdout.strip() == ""


_REHEARSAL_RANKS = 4      # + this pytest process = 5 processes on the card; the GPU boxes allow 6 at once ("process guard"), so 8 is not possible here


def _rehearsal_args(n):
    return ["--gpus", str(n), "--steps", "2", "--warmup", "1", "--batch", "4", "--check-replicas",
            "--no-cpu-baseline", "--no-ge2e", "--no-fp32", "--no-roofline"]


def _check_rehearsal_line(r, n):
    import json
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    cfg = rec["config"]
    assert rec["n_gpus"] == n and cfg["global_batch"] == 4 * n and cfg["parallelism"] == "dp%d" % n and rec["dist_backend"] == "gloo"
    assert cfg["replica_checksums_equal"] is True                       # Text2Mel + SSRN after the steps; the adversarial models exit non-zero if not
    assert cfg["adversarial_text2mel_ms"] > 0 and cfg["adversarial_ssrn_ms"] > 0
    return rec


def test_bench_many_rank_rehearsal_on_one_gpu_with_the_adversarial_cycle_and_replica_checksums():
    """Config 4's whole launch structure with more than two processes: `python bench.py --gpus 4` starts four ranks that share the one
    GPU over gloo -- four captures in thread_local mode side by side, four process-group inits, `local % ndev`, a free port -- runs the
    segmented data-parallel step of Text2Mel and SSRN AND the WGAN-GP cycle (global adaptive weight, bucketed generator all-reduce,
    packed critic all-reduce), then all-gathers a checksum of every model's parameters: all replicas bitwise equal, one JSON line,
    n_gpus = 4.  (Eight ranks cannot share a card here: the boxes stop a job with more than six GPU processes; the eight-process launch
    itself is rehearsed without a device by tests/test_host_cpu.py.)  train/adversarial_wasserstein_gp.py:183-196."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    n = _REHEARSAL_RANKS
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + _rehearsal_args(n), env=dict(env, SSV_DIST_BACKEND="gloo"),
                       capture_output=True, text=True, timeout=1200)
    rec = _check_rehearsal_line(r, n)
    assert abs(rec["value"] - n * 4 * 325 / (rec["ms_per_step"] * 1e-3)) < 1e-3 * rec["value"]


def test_bench_many_rank_rehearsal_under_the_torchrun_launcher():
    """The same through the driver's launcher form: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N (the launcher never touches the GPU; the ranks read RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* from its environment)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    n = _REHEARSAL_RANKS
    port = 29700 + os.getpid() % 200
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "bench.py")] + _rehearsal_args(n) + ["--no-adversarial"],
                       env=dict(env, SSV_DIST_BACKEND="gloo"), capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    import json
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == n and rec["config"]["replica_checksums_equal"] is True
