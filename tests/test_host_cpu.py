"""CPU-only tests of the host logic: the module mirror (names, state-dict keys, initialisation order),
the guided-attention matrix, synthetic batches and the data-parallel wrapper over gloo (world_size 2)."""
import hashlib
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from _golden import load, sub
from spoofsv_amd import train
from spoofsv_amd.ge2e import GE2ELoss, SpeechEmbedder
from spoofsv_amd.tts import SSRN, highwayConv, melSyn


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def test_state_dict_keys_match_reference():
    g = load("melsyn_train.npz")
    hidden, temb = int(g["dims"][0]), int(g["dims"][1])
    m = melSyn(34, True, 200, textemb_dim=temb, freq_bins=80, hidden_dim=hidden)
    ref = sub(g, "sd/")
    assert list(m.state_dict().keys()) == list(ref.keys())
    for k, v in m.state_dict().items():
        assert tuple(v.shape) == tuple(ref[k].shape), k
    s = SSRN(80, 65, 16)
    ref = sub(load("ssrn_small.npz"), "sd/")
    assert list(s.state_dict().keys()) == list(ref.keys())
    e = SpeechEmbedder(40, 32, 3, 16)
    ref = sub(load("ge2e_embedder.npz"), "sd/")
    assert list(e.state_dict().keys()) == list(ref.keys())
    hc = highwayConv(16, 3, 9, causal=True)
    assert list(hc.state_dict().keys()) == ["conv.weight", "conv.bias", "ln1.weight", "ln1.bias", "ln2.weight", "ln2.bias"]
    L = GE2ELoss(torch.device("cpu"))
    assert float(L.w) == 10.0 and float(L.b) == -5.0


def test_full_size_init_reproduces_reference_checksums():
    """G9: manual_seed -> construct -> apply(init_weights) gives the reference's weights bit for bit, so
    full-size parity runs need no shipped checkpoints."""
    g = load("init_pin.npz")
    for tag, ctor in (("t2m", lambda: melSyn(34, True, 200, 128, 80, 256)), ("ssrn", lambda: SSRN(80, 513, 256))):
        torch.manual_seed(1234)
        m = ctor()
        m.apply(train.init_weights)
        sd = m.state_dict()
        assert list(sd.keys()) == [str(n) for n in g[tag + "/names"]]
        assert sum(p.numel() for p in m.parameters()) == int(g[tag + "/numel"])
        for k, sha in zip(sd.keys(), g[tag + "/sha"]):
            assert _sha(sd[k].numpy()) == str(sha), k


def test_guided_attention_matrix_matches_reference_loop():
    g = load("gaw.npz")
    W = train.guided_attention_mat(186, 325)
    assert np.array_equal(W.numpy()[g["n"], g["t"]], g["w"])
    assert np.array_equal(W.numpy()[93], g["row93"])


def test_synthetic_batches_and_shift():
    mel, text, spk = train.synthetic_text2mel_batch(3, N=20, T=17, seed=5)
    assert mel.shape == (3, 80, 17) and text.shape == (3, 1, 20) and spk.shape == (3, 200, 1)
    assert text.dtype == torch.int64 and int(text.min()) >= 0 and int(text.max()) < 34
    assert torch.all(text[:, :, -1] == 0) and torch.all(text[:, :, -2] == 1)
    s = train.shift_right(mel)
    assert torch.all(s[:, :, 0] == 0) and torch.equal(s[:, :, 1:], mel[:, :, :-1])
    mel2, lin = train.synthetic_ssrn_batch(2, T=9, seed=1)
    assert lin.shape == (2, 513, 36)
    a, _, _ = train.synthetic_text2mel_batch(3, N=20, T=17, seed=5)
    assert torch.equal(a, mel)          # deterministic per seed (rank)


def _ddp_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                       # replicas start DIFFERENT on purpose
    params = [torch.nn.Parameter(torch.randn(7, 5)), torch.nn.Parameter(torch.randn(300)), torch.nn.Parameter(torch.randn(3))]
    ddp = train.DataParallelRanks(params, bucket_mb=0.001)   # tiny buckets -> several all-reduces
    ddp.broadcast_parameters(0)
    after_bcast = [p.detach().clone() for p in params]
    for i, p in enumerate(params):
        p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
    params[2].grad = None if False else params[2].grad
    ddp.all_reduce_grads()
    l = ddp.all_reduce_mean(torch.tensor(float(rank)), torch.tensor(10.0 * rank))
    q.put((rank, [p.numpy() for p in after_bcast], [p.grad.numpy() for p in params], [float(v) for v in l]))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_ranks_gloo_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_ddp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, w0, g0, l0), (r1, w1, g1, l1) = res
    for a, b in zip(w0, w1):                             # broadcast made the replicas identical
        assert np.array_equal(a, b)
    for i, (a, b) in enumerate(zip(g0, g1)):             # gradients averaged: (1 + 2)/2 * (i+1)
        assert np.allclose(a, 1.5 * (i + 1)) and np.array_equal(a, b)
    assert l0 == l1 == [0.5, 5.0]


def test_ge2e_eval_host_logic_matches_reference_loops():
    """cossim_eval against the reference's own get_cossim output (fixture) and the one-shot EER sweep against the
    oracle's literal restatement of the reference's threshold loops, both test variants."""
    import numpy as np
    import torch
    from _golden import load, rel_err, t
    from oracle import ge2e_oracle as GO
    from spoofsv_amd import ge2e_harness as GH
    g = load("ge2e_train.npz")
    assert rel_err(GH.cossim_eval(t(g["ev_ver"]), t(g["ev_cent"])), t(g["ev_sim"])) < 1e-5
    torch.manual_seed(6)
    for trial in range(4):
        N, size_1, enroll = 5, 16, 2
        es1 = 2 * enroll
        sim = 0.45 + 0.6 * torch.rand(N, size_1 - es1, N)
        idx = torch.arange(N)
        sim[idx, :, idx] += 0.15                                   # genuine trials score higher
        a, b = GH.eer_sweep(sim, size_1, es1, spoof=True), GO.eer_sweep(sim, size_1, es1, spoof=True)
        for k in b:
            assert abs(a[k] - b[k]) < 1e-6, (trial, k, a[k], b[k])
        sim2 = sim[:, :6]
        a, b = GH.eer_sweep(sim2, size_1, es1, spoof=False), GO.eer_sweep(sim2, size_1, es1, spoof=False)
        for k in b:
            assert abs(a[k] - b[k]) < 1e-6, (trial, k, a[k], b[k])


def test_prefetcher_yields_the_same_batches_in_order():
    import json
    import torch
    from spoofsv_amd import harness
    cfg = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "config.json")))
    cfg["SYNTHETIC_BATCHES_PER_EPOCH"] = 5
    cfg["MAX_TEXT_LEN"], cfg["MAX_FRAME_NUM"] = 20, 30
    plain = list(harness.BatchSource(cfg, "train_text2mel", 3))
    pre = list(harness.Prefetcher(harness.BatchSource(cfg, "train_text2mel", 3), torch.device("cpu")))
    assert len(pre) == len(plain) == 5
    for a, b in zip(pre, plain):
        assert a.keys() == b.keys()
        for k in a:
            assert torch.equal(a[k], b[k])

    class Broken:
        def __len__(self): return 1
        def __iter__(self):
            yield {"data_0": torch.zeros(1)}
            raise ValueError("loader failed")
    import pytest
    with pytest.raises(ValueError):
        list(harness.Prefetcher(Broken(), torch.device("cpu")))


class _CpuEmbedder(torch.nn.Module):
    """Stand-in for SpeechEmbedder on the CPU (same structure: LSTM -> last frame -> Linear -> L2 normalise)."""

    def __init__(self):
        super().__init__()
        self.LSTM_stack = torch.nn.LSTM(6, 8, num_layers=2, batch_first=True)
        self.projection = torch.nn.Linear(8, 5)

    def forward(self, x):
        h, _ = self.LSTM_stack(x.float())
        e = self.projection(h[:, -1])
        return e / torch.norm(e, dim=1, keepdim=True)


class _CpuGE2ELoss(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.tensor(10.0))
        self.b = torch.nn.Parameter(torch.tensor(-5.0))

    def forward(self, emb):
        from oracle import ge2e_oracle as GO
        out = GO.ge2e_loss(emb, self.w, self.b)
        return out[0] if isinstance(out, tuple) else out


def _ge2e_setup():
    torch.manual_seed(77)
    net, crit = _CpuEmbedder(), _CpuGE2ELoss()
    x = torch.randn(4, 3, 7, 6)            # N = 4 speakers, M = 3 utterances, 7 frames, 6 mels
    return net, crit, x


def _ge2e_worker(rank, world, port, q):
    from spoofsv_amd import ge2e
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    net, crit, x = _ge2e_setup()
    opt = torch.optim.SGD([{"params": net.parameters()}, {"params": crit.parameters()}], lr=0.01)
    n_local = x.shape[0] // world
    loss = ge2e.sharded_train_iteration(net, crit, opt, x[rank * n_local:(rank + 1) * n_local], n_local, 3)
    q.put((rank, float(loss), [p.detach().numpy().copy() for p in list(net.parameters()) + list(crit.parameters())]))
    dist.barrier()
    dist.destroy_process_group()


def test_ge2e_sharded_iteration_equals_single_process_gloo_world2():
    """speakers split over 2 ranks, one all-gather of the embeddings, SUM all-reduce of the embedder
    gradients -> the same loss and the same post-step weights as train_iteration on the whole batch."""
    from spoofsv_amd import ge2e
    net, crit, x = _ge2e_setup()
    opt = torch.optim.SGD([{"params": net.parameters()}, {"params": crit.parameters()}], lr=0.01)
    want_loss = float(ge2e.train_iteration(net, crit, opt, x, 4, 3))
    want = [p.detach().numpy().copy() for p in list(net.parameters()) + list(crit.parameters())]
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_ge2e_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, loss, weights in res:
        assert abs(loss - want_loss) <= 1e-5 * abs(want_loss)
        for a, b in zip(weights, want):
            assert np.allclose(a, b, rtol=1e-5, atol=1e-6)


def _make_corpus(tmp, n_items=5, with_cache=True, wav=False):
    """A corpus in the reference's layout (data/dataset.py:37-51,84-91): path lists, one-line transcripts, speaker codes."""
    import json as _json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = _json.load(open(os.path.join(root, "config.json")))
    data = os.path.join(tmp, "corpus") + os.sep
    cfg.update(DATA_ROOT_DIR=data, SPK_EMB_DIR=os.path.join(tmp, "spk_emb") + os.sep, SRC_ROOT_DIR=os.path.join(tmp, "runs") + os.sep)
    os.makedirs(os.path.join(data, "data_path", "ordinary"))
    os.makedirs(cfg["SPK_EMB_DIR"])
    spec = os.path.join(tmp, "spec") + os.sep
    rng = np.random.RandomState(0)
    wavs, txts = [], []
    for i in range(n_items):
        spk = "p%d" % (225 + i % 2)
        wdir, tdir = os.path.join(data, "wav22", spk), os.path.join(data, "txt", spk)
        os.makedirs(wdir, exist_ok=True); os.makedirs(tdir, exist_ok=True)
        name = "%s_%03d" % (spk, i + 1)
        wavs.append(os.path.join(wdir, name + ".wav"))
        txts.append(os.path.join(tdir, name + ".txt"))
        open(txts[-1], "w").write(["Please call Stella.", "Ask her to bring these things!", "Six spoons", "of fresh \"snow\" peas", "Bob."][i % 5] + "\n")
        np.save(os.path.join(cfg["SPK_EMB_DIR"], spk + ".npy"), rng.rand(200).astype(np.float32))
        if wav:
            from scipy.io import wavfile
            n = 22050 // 2 + 3000 * i
            y = 0.3 * np.sin(2 * np.pi * (200 + 40 * i) * np.arange(n) / 22050) * np.hanning(n)
            wavfile.write(wavs[-1], 22050, (y * 32767).astype(np.int16))
        if with_cache:
            T = 7 + 3 * i
            os.makedirs(os.path.join(spec, spk), exist_ok=True)
            np.save(spec + wavs[-1][-17:-4] + "_mel.npy", rng.rand(80, T).astype(np.float32))
            np.save(spec + wavs[-1][-17:-4] + "_lin.npy", rng.rand(513, 4 * T).astype(np.float32))
    for mode in ("train", "validate", "synthesize"):
        open(os.path.join(data, "data_path", "ordinary", "wav.path." + mode), "w").write("\n".join(wavs) + "\n")
        open(os.path.join(data, "data_path", "ordinary", "txt.path." + mode), "w").write("\n".join(txts) + "\n")
    return cfg, spec


def test_corpus_source_collates_like_the_reference(tmp_path):
    """data/dataset.py:175-258: text ids (lower-case, 'E' appended, unknown characters dropped, '"' folded), zero padding to the
    batch's longest item, speaker codes (B, 200, 1); one permutation per epoch; ranks take disjoint slices."""
    from spoofsv_amd import harness
    cfg, spec = _make_corpus(str(tmp_path))
    src = harness.BatchSource(cfg, "train_text2mel", 2, spec, pattern="conditional", mode="validate")      # validate: file order
    batches = list(src)
    assert len(src) == 3 and len(batches) == 3
    b0 = batches[0]
    assert tuple(b0["data_0"].shape) == (2, 80, 10) and b0["data_1"].dtype == torch.int64 and tuple(b0["data_2"].shape) == (2, 200, 1)
    vocab = cfg["VOCABULARY"]
    want0 = [vocab.index(ch) for ch in "please call stella.e".replace("e", "e")[:-1]] + [1]
    assert b0["data_1"][0, 0, :len(want0)].tolist() == want0 and b0["data_1"].shape[-1] == len("ask her to bring these things") + 1
    assert torch.all(b0["data_1"][0, 0, len(want0):] == 0)                    # 'P' padding
    assert torch.all(b0["data_0"][0, :, 7:] == 0) and not torch.all(b0["data_0"][1, :, 7:] == 0)
    assert tuple(batches[2]["data_0"].shape) == (1, 80, 19)                   # last, partial batch (drop_last is False)
    quote = harness.text2id('of fresh "snow" peas', vocab)
    assert quote.count(len(vocab) - 2) == 2                                    # '"' shares the id of "'"
    ssrn = list(harness.BatchSource(cfg, "train_ssrn", 2, spec, mode="validate"))
    assert tuple(ssrn[1]["data_0"].shape) == (2, 80, 16) and tuple(ssrn[1]["data_1"].shape) == (2, 513, 64)
    tr = harness.BatchSource(cfg, "train_ssrn", 5, spec, mode="train")
    e1, e2 = next(iter(tr))["data_0"], next(iter(tr))["data_0"]
    assert e1.shape[0] == 5 and not torch.equal(e1, e2)                       # reshuffled every epoch
    r0 = list(harness.BatchSource(cfg, "train_ssrn", 2, spec, mode="validate", rank=0, world=2))
    r1 = list(harness.BatchSource(cfg, "train_ssrn", 2, spec, mode="validate", rank=1, world=2))
    assert len(r0) == 2 and r0[0]["data_0"].shape[-1] == 10 and r1[0]["data_0"].shape[-1] == 16
    # 5 utterances, global batch 2 x 2: the ragged last global batch wraps around, so both ranks run the same number of
    # iterations on full shards (a rank that stopped early would leave the others waiting in an all-reduce)
    assert len(r1) == 2 and all(b["data_0"].shape[0] == 2 for b in r0 + r1)
    assert torch.equal(r0[1]["data_0"][1, :, :7], b0["data_0"][0, :, :7])          # rank 0's second item of the last batch = utterance 0 again


# ---- data parallel: gradient arena, segmented backward with per-bucket all-reduce -------------------------------
class _CutNet(torch.nn.Module):
    """CPU stand-in with the structure train.DataParallelRanks(model=...) expects of tts.melSyn / tts.SSRN: a ``_cut`` hook in
    the forward and a ``ddp_plan``: three stages, two cuts, three gradient buckets in backward order."""

    def __init__(self):
        super().__init__()
        self._cut = lambda name, x: x
        self.a = torch.nn.Linear(6, 8)
        self.b = torch.nn.Linear(8, 8)
        self.c = torch.nn.Linear(8, 3)

    def forward(self, x):
        h = self._cut("ab", torch.tanh(self.a(x)))
        h = self._cut("bc", torch.tanh(self.b(h)))
        return self.c(h)

    def ddp_plan(self):
        return [("c", [[self.c.weight], [self.c.bias]]), ("b", [[self.b.weight, self.b.bias]]), ("a", [[self.a.weight], [self.a.bias]])], ["bc", "ab"]


def _cutnet_setup():
    torch.manual_seed(5)
    net = _CutNet()
    x, y = torch.randn(8, 6), torch.randn(8, 3)
    return net, x, y


def test_grad_arena_layout_views_and_adopt():
    from spoofsv_amd import gradarena, train
    net, x, y = _cutnet_setup()
    ddp = train.DataParallelRanks(model=net)
    ar = ddp.arena
    assert ddp.cut_names == ["bc", "ab"] and ddp.n_buckets == 3 and [n for n in ar.names] == ["c", "b", "a"]
    for (a0, a1), (b0, b1) in zip(ar.ranges, ar.ranges[1:]):
        assert a1 == b0 and a0 % 64 == 0                               # buckets are contiguous, 256-byte aligned ranges
    for p in net.parameters():
        v = gradarena.view(p, claim=False)
        assert v is not None and v.shape == p.shape and v.data_ptr() == ar.slot(p).data_ptr()
        assert (v.data_ptr() - ar.flat.data_ptr()) % 16 == 0           # every group starts 16-byte aligned
    blk = gradarena.block((net.b.weight, net.b.bias), 9, 8, claim=False)            # adjacent slots of one group: one (rows, C) block
    assert blk is not None and blk.data_ptr() == gradarena.view(net.b.weight, claim=False).data_ptr()
    assert gradarena.block((net.b.bias, net.b.weight), 9, 8) is None and gradarena.block((net.a.weight, net.a.bias), 8, 8) is None
    assert gradarena.view(net.a.weight.unsqueeze(-1), claim=False).shape == (8, 6, 1)          # nn.Linear weight used as a 1x1 conv
    assert gradarena.view(torch.zeros(8, 6)) is None
    # gradients produced outside the arena are copied in; missing ones become zero slots
    torch.nn.functional.mse_loss(net(x), y).backward()
    want = {n: p.grad.clone() for n, p in net.named_parameters()}
    net.a.bias.grad = None
    assert ar.adopt() == 5
    for n, p in net.named_parameters():
        assert p.grad.data_ptr() == ar.slot(p).data_ptr()
        assert torch.equal(p.grad, torch.zeros_like(p) if n == "a.bias" else want[n])
    # a slot that already holds p.grad is not handed out again: a second gradient of the parameter in one backward (or a
    # backward without zero_grad(set_to_none=True)) must be ACCUMULATED by autograd, not written over what p.grad aliases
    assert gradarena.view(net.a.weight) is None and gradarena.block((net.b.weight, net.b.bias), 9, 8) is None
    assert gradarena.grad_like(net.a.weight).data_ptr() != ar.slot(net.a.weight).data_ptr()
    before = net.c.weight.grad.clone()
    torch.nn.functional.mse_loss(net(x), y).backward()                # no zero_grad: accumulates into the arena in place
    assert net.c.weight.grad.data_ptr() == ar.slot(net.c.weight).data_ptr() and torch.allclose(net.c.weight.grad, 2 * before)
    net.a.weight.grad = None
    assert gradarena.view(net.a.weight, claim=False) is not None      # free again after zero_grad(set_to_none=True)
    ar.release()
    assert gradarena.view(net.a.weight) is None


class _TwiceFn(torch.autograd.Function):
    """y = x * w with the parameter gradient written where the gradient arena says (as the fused operators of ops.py do)."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return x * w

    @staticmethod
    def backward(ctx, dy):
        from spoofsv_amd import gradarena
        x, w = ctx.saved_tensors
        dw = gradarena.grad_like(w)
        dw.copy_(dy * x)
        return dy * w, dw


def test_grad_arena_parameter_used_twice_in_one_backward_is_summed_not_overwritten():
    """A leaf's AccumulateGrad runs only after ALL its users have delivered, so p.grad is still None when the second user asks
    for a slot: the arena must remember that the slot is taken (claimed) until autograd has accumulated, or both users write
    the same memory and autograd adds the slot to itself (2*g2 instead of g1 + g2)."""
    from spoofsv_amd import gradarena
    torch.manual_seed(0)
    w = torch.nn.Parameter(torch.randn(6))
    x1, x2 = torch.randn(6), torch.randn(6)
    ar = gradarena.GradArena([("only", [[w]])])
    for it in range(2):                       # the claim is released by the post-accumulate hook: the second iteration gets the slot again
        w.grad = None
        (_TwiceFn.apply(x1, w).sum() + 3.0 * _TwiceFn.apply(x2, w).sum()).backward()
        assert torch.allclose(w.grad, x1 + 3.0 * x2), (it, w.grad, x1 + 3.0 * x2)
        assert not ar.claimed
        ar.adopt()
        assert w.grad.data_ptr() == ar.slot(w).data_ptr() and torch.allclose(w.grad, x1 + 3.0 * x2)
    # single use: the kernel's destination IS the slot and autograd adopts it without a copy
    w.grad = None
    _TwiceFn.apply(x1, w).sum().backward()
    assert w.grad.data_ptr() == ar.slot(w).data_ptr() and torch.allclose(w.grad, x1)
    ar.release()


def test_grad_arena_dropped_without_release_leaves_no_hooks_and_a_later_backward_runs():
    """An arena that is garbage-collected (a DataParallelRanks dropped between two train() calls on the same model) must not leave
    post-accumulate hooks behind: a hook that returned anything but None made the next backward raise 'hook returned bool', and
    stale hooks piled up with every new arena over the same parameters."""
    import gc
    from spoofsv_amd import gradarena
    torch.manual_seed(0)
    w = torch.nn.Parameter(torch.randn(6))
    x1 = torch.randn(6)
    for _ in range(3):
        ar = gradarena.GradArena([("only", [[w]])])
        w.grad = None
        _TwiceFn.apply(x1, w).sum().backward()
        assert w.grad.data_ptr() == ar.slot(w).data_ptr()
        del ar
        gc.collect()
        assert len(w._post_accumulate_grad_hooks or {}) == 0
        assert gradarena.view(w) is None                  # the slot of a dead arena is stale
        w.grad = None
        _TwiceFn.apply(x1, w).sum().backward()            # used to raise TypeError: expected Variable, but hook returned 'bool'
        assert torch.allclose(w.grad, x1)
    # the hook itself returns None whether or not the arena is alive
    ar = gradarena.GradArena([("only", [[w]])])
    hook = list(w._post_accumulate_grad_hooks.values())[0]
    assert hook(w) is None
    ar.release(); ar.release()                            # idempotent
    assert len(w._post_accumulate_grad_hooks or {}) == 0


def _segmented_iteration(net, x, y, ddp):
    """forward with cuts, backward segment by segment, each bucket's all-reduce started right after its segment."""
    from spoofsv_amd import train
    cuts = train.Cuts(ddp.cut_names)
    for p in net.parameters():
        p.grad = None
    with train._cuts_installed(net, cuts):
        loss = torch.nn.functional.mse_loss(net(x), y)
    order = []
    segs = train.backward_segments(cuts, [(loss, torch.full_like(loss, ddp.grad_scale))], None, ddp)
    phases = []
    for i, seg in enumerate(segs):
        phases.append(("graph", seg))
        def hand_over(i=i):
            # bucket i goes to its all-reduce while the rest of backward has not run: its gradients exist, later buckets' do not
            assert all(p.grad is not None for p in ddp.bucket_params[i])
            assert all(p.grad is None for later in ddp.bucket_params[i + 1:] for p in later)
            order.append(i)
            ddp.start_bucket(i)
        phases.append(("eager", hand_over))
    phases.append(("eager", ddp.finish))
    train.PhasedStep(phases, graph=False).run()
    assert order == [0, 1, 2] and net._cut("x", loss) is loss          # hook restored after the forward
    return loss


def _cutnet_worker(rank, world, port, q):
    from spoofsv_amd import train
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    net, x, y = _cutnet_setup()
    ddp = train.DataParallelRanks(model=net)
    n = x.shape[0] // world
    sl = slice(rank * n, (rank + 1) * n)
    _segmented_iteration(net, x[sl], y[sl], ddp)
    q.put((rank, {k: p.grad.numpy().copy() for k, p in net.named_parameters()},
           all(p.grad.data_ptr() == ddp.arena.slot(p).data_ptr() for p in net.parameters())))
    dist.barrier()
    dist.destroy_process_group()


def test_segmented_backward_with_bucketed_all_reduce_gloo_world2():
    """Two ranks, half the batch each, backward in three segments with one all-reduce per gradient bucket: every rank ends
    with the gradient a single process computes on the whole batch, living in the arena."""
    from spoofsv_amd import train
    net, x, y = _cutnet_setup()
    torch.nn.functional.mse_loss(net(x), y).backward()
    want = {k: p.grad.numpy().copy() for k, p in net.named_parameters()}
    # single process through the same segmented path: identical to the plain backward
    net2, _, _ = _cutnet_setup()
    _segmented_iteration(net2, x, y, train.DataParallelRanks(model=net2))
    for k, p in net2.named_parameters():
        assert np.allclose(p.grad.numpy(), want[k], rtol=1e-6, atol=1e-7), k
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_cutnet_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, grads, in_arena in res:
        assert in_arena
        for k in want:
            assert np.allclose(grads[k], want[k], rtol=1e-5, atol=1e-7), (rank, k)
    for k in want:
        assert np.array_equal(res[0][1][k], res[1][1][k])


def _packed_worker(rank, world, port, q):
    from spoofsv_amd import train
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    params = [torch.nn.Parameter(torch.zeros(4, 3)), torch.nn.Parameter(torch.zeros(5))]
    ddp = train.DataParallelRanks(params)
    for i, p in enumerate(params):
        p.grad = torch.full_like(p, float(rank + 1) * (i + 1)) * ddp.grad_scale         # pre-scaled, as the steps seed backward
    h = ddp.pack(extra=(torch.tensor(float(rank)) * ddp.grad_scale, torch.tensor(4.0 + rank) * ddp.grad_scale))
    ddp.exchange()
    ddp.finish()
    extras = ddp.unpack(h)
    vec = torch.tensor([1.0 * rank, 2.0, 3.0 * rank, 4.0])
    ddp.all_reduce_mean_(vec)
    q.put((rank, [p.grad.numpy().copy() for p in params], [float(e) for e in extras], vec.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_packed_exchange_with_loss_scalars_gloo_world2():
    """The critics' path: gradients and the two loss scalars packed into one flat bucket, ONE all-reduce, gradients re-pointed
    at the reduced slices; plus the in-place scalar mean used for the adaptive generator weight."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 34500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_packed_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, grads, extras, vec in res:
        assert np.allclose(grads[0], 1.5) and np.allclose(grads[1], 3.0)
        assert np.allclose(extras, [0.5, 4.5]) and np.allclose(vec, [0.5, 2.0, 1.5, 4.0])


def test_fused_adam_state_dict_loads_into_torch_adam_and_steps():
    """The reference's -R resume (train/ordinary.py:188-197) loads ``optimizer_state_dict`` into torch.optim.Adam and steps:
    the saved param group must carry every key torch's Adam reads (weight_decay, amsgrad, maximize, ...)."""
    import pytest
    from spoofsv_amd import train
    p = torch.nn.Parameter(torch.randn(7))
    opt = train.FusedAdam([p], 2e-4, (0.5, 0.9), 1e-6)
    opt.state[p] = {"step": torch.tensor(0.0), "exp_avg": torch.full_like(p, 0.1), "exp_avg_sq": torch.full_like(p, 0.2)}
    opt._steps = 3
    sd = opt.state_dict()
    ref = torch.optim.Adam([torch.nn.Parameter(p.detach().clone())], 1e-3)
    assert set(ref.param_groups[0]) == set(sd["param_groups"][0])
    ref.load_state_dict(sd)
    q = ref.param_groups[0]["params"][0]
    q.grad = torch.ones_like(q)
    ref.step()
    assert float(ref.state[q]["step"]) == 4.0 and ref.param_groups[0]["lr"] == 2e-4 and ref.param_groups[0]["betas"] == (0.5, 0.9)
    for bad in (dict(weight_decay=0.1), dict(amsgrad=True), dict(maximize=True)):
        with pytest.raises(ValueError):
            train.FusedAdam([p], **bad)


def test_trainers_refuse_what_they_do_not_implement():
    import json
    import pytest
    from spoofsv_amd import harness
    cfg = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "config.json")))
    cfg["APPLY_DROPOUT"] = True
    with pytest.raises(RuntimeError, match="APPLY_DROPOUT"):
        harness.ordinary_train("train_text2mel", "conditional", cfg)
    with pytest.raises(RuntimeError, match="APPLY_DROPOUT"):
        harness.adversarial_train("train_ssrn", "conditional", cfg)
    from spoofsv_amd.critic import melDisc
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        melDisc(80, 16)(torch.zeros(1, 80, 8))                        # the critics have no stock-op branch either


def _pad_worker(rank, world, port, q):
    from spoofsv_amd import harness
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    T, N = (20, 9) if rank == 0 else (14, 12)                    # each rank's own longest item
    sp = {"data_0": torch.ones(3, 80, T), "data_1": torch.ones(3, 1, N, dtype=torch.int64), "data_2": torch.ones(3, 200, 1),
          "data_3": torch.ones(3, 513, 4 * T)}
    out = harness._pad_to_global(sp, world)
    q.put((rank, {k: tuple(v.shape) for k, v in out.items()}, float(out["data_0"].sum()), int(out["data_1"].sum()), float(out["data_3"].sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_ragged_rank_batches_are_padded_to_the_global_longest_item_gloo_world2():
    """every rank zero-pads its shard to the longest mel / text of the GLOBAL batch (one all-reduce(max)), so the
    mean-type losses see the shapes a single process would give the whole batch (data/dataset.py:215-224 pads per batch)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_pad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, shapes, s0, s1, s3 in res:
        assert shapes == {"data_0": (3, 80, 20), "data_1": (3, 1, 12), "data_2": (3, 200, 1), "data_3": (3, 513, 80)}, (rank, shapes)
        T, N = (20, 9) if rank == 0 else (14, 12)
        assert s0 == 3 * 80 * T and s1 == 3 * N and s3 == 3 * 513 * 4 * T        # padding is zeros ('P' = id 0 for the text)


def _run_bench(args, env_extra, timeout=300):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_refuses_a_world_size_that_is_not_the_gpus_asked_for():
    """`--gpus N` is what the line reports as n_gpus: a launch with another WORLD_SIZE exits non-zero before any GPU call
    (the reference turns multi-GPU on with one flag, train/adversarial_wasserstein_gp.py:183-196)."""
    r = _run_bench(["--gpus", "8"], {"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr and r.stdout.strip() == ""


def test_bench_gpus_n_without_a_launcher_starts_n_ranks_and_propagates_their_failure():
    """`python bench.py --gpus 2` with no WORLD_SIZE becomes two ranks (children of a parent that never touches the GPU).  On this
    GPU-less host every rank stops at its own "needs a ROCm GPU" check: one such message per rank that reached it, and the parent's exit
    code is non-zero with nothing on stdout."""
    r = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], {})
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU present: the launched ranks would run the benchmark (covered by tests/test_gpu_ddp.py)")
    assert r.returncode != 0 and r.stdout.strip() == ""
    # one message per rank that got as far as the check before the launcher ended it (the launcher waits 10 s for the others)
    assert 1 <= r.stderr.count("needs a ROCm GPU") <= 2, r.stderr


def test_bench_gpus_8_launch_plumbing_with_eight_processes_no_gpu():
    """BASELINE config 4 is eight ranks; nothing on this host can run eight GPU processes, but everything `bench.py --gpus 8` does
    around the GPU work can: eight children of a parent that touches no device, a free rendezvous port on 127.0.0.1, a gloo
    group of eight, local ranks folded onto the visible devices, one all-gather, ONE JSON line from rank 0, clean teardown
    (`--launch-check`).  The first real 8-GPU run is then not the first 8-process run of this launcher
    (train/adversarial_wasserstein_gp.py:183-196 turns multi-GPU on with one flag inside one process)."""
    import json
    r = _run_bench(["--gpus", "8", "--launch-check"], {}, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["launch_check"] is True and rec["n_gpus"] == 8 and rec["ranks"] == list(range(8)) and rec["distinct_pids"] == 8
    assert rec["local_ranks_folded_onto"] == list(range(min(8, rec["devices_visible"])))


def test_bench_gpus_8_one_rank_dying_ends_the_other_seven_and_fails():
    """A rank that exits after the rendezvous (here: rank 5, exit code 3) leaves seven ranks waiting in a collective for ever; the
    launcher must end exactly those and return non-zero with nothing on stdout -- and within its grace period, not at a timeout."""
    import time
    t0 = time.time()
    r = _run_bench(["--gpus", "8", "--launch-check"], {"SSV_LAUNCH_CHECK_DIE": "5"}, timeout=300)
    assert r.returncode == 3 and r.stdout.strip() == "", (r.returncode, r.stdout)
    assert time.time() - t0 < 120


def test_bench_launch_check_under_the_torchrun_launcher():
    """The driver's form: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
    bench.py --gpus N`; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the launcher's environment."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    port = 29600 + os.getpid() % 300
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "8", "--launch-check"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    assert json.loads(lines[0])["ranks"] == list(range(8))
