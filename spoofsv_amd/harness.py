"""Drop-in trainers and synthesiser with the reference's call signatures, running the generator on the HIP
hot path:

  ordinary_train(train_step, train_pattern, cfg, spec_dir, resume_checkpoints, current_time)      train/ordinary.py:130-135
  adversarial_train(...same...)                                          train/adversarial_wasserstein_gp.py:148-153
  synthesize(pattern, cfg, spec_dir, current_time)                                                      synthesize.py:41

``cfg`` is the reference's config.json dict (same keys).  Checkpoints are written with the reference's key
names (``model_state_dict``, ``optimizer_state_dict`` / ``opt_state_dict_syn`` ... ) and
file names (``text2mel_iteration_N.tar.pth``, ``*_best_model.tar.pth``) so that either code base resumes the
other's runs.

Data: the reference's VCTK feature extraction (librosa STFT etc., data/dataset.py) is CPU preprocessing outside
the hot path and is not rebuilt.  Batches come from ``BatchSource``: synthetic VCTK-shaped tensors, or -- when
``spec_dir`` holds the reference's own ``pXXX/pXXX_NNN_{mel,lin}.npy`` spectrogram cache (data/dataset.py:85-91)
plus ``cfg['SPK_EMB_DIR']/pXXX.npy`` -- those files, zero-padded per batch like collate_pad_3 (:215-224).
Two extra, optional config keys bound a run: ``MAX_ITERATIONS`` and ``SYNTHETIC_BATCHES_PER_EPOCH``.
Griffin-Lim / wav writing (synthesize.py:138-147) is CPU post-processing: synthesize() stops
at the linear spectrogram and stores it as .npy.
"""
import glob
import os
import time

import numpy as np
import torch

from . import ops, resident, train
from .critic import linDisc, melDisc
from .tts import SSRN, melSyn


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("spoofsv_amd: no ROCm device visible; the HIP hot path has no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _distributed(cfg):
    """-> (rank, world).  One process per GPU replaces the reference's in-process nn.DataParallel
    (``MULTI_GPU``, train/ordinary.py:165-173, train/adversarial_wasserstein_gp.py:183-196): launched under ``torchrun`` /
    ``python -m torch.distributed.run`` (RANK / WORLD_SIZE / LOCAL_RANK in the environment) the trainers build the process
    group HERE, before the first GPU call, bind the rank's GPU, shard the batches by rank and average gradients over
    RCCL.  ``MULTI_GPU: true`` in a single process with several GPUs visible is refused: silently training on one GPU (or
    N unsynchronised replicas) is not what the flag asks for.  ``APPLY_DROPOUT`` selects the reference's dropout generator
    (models/TTSModel_dropout.py), which is outside this path: refused loudly rather than ignored."""
    import torch.distributed as dist
    if cfg.get("APPLY_DROPOUT"):
        raise RuntimeError("spoofsv_amd: APPLY_DROPOUT=true selects models/TTSModel_dropout.py, which this hot path does not "
                           "implement (SURVEY.md section 2, row 5); set it to false")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    if world > 1 or os.environ.get("SSV_FORCE_COLLECTIVES") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        # bind the rank's GPU FIRST and tell the process group which device it serves: RCCL then creates its communicator
        # on that device eagerly instead of guessing from the first collective (device_count() does not initialise HIP)
        ndev = torch.cuda.device_count()
        local = int(os.environ.get("LOCAL_RANK", rank)) % max(ndev, 1)
        torch.cuda.set_device(local)
        backend = os.environ.get("SSV_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        return rank, world
    if cfg.get("MULTI_GPU") and torch.cuda.device_count() > 1:
        raise RuntimeError("spoofsv_amd: MULTI_GPU=true needs one process per GPU: launch with `python -m torch.distributed.run "
                           "--nproc-per-node <gpus> --master-addr 127.0.0.1 main.py ...` (set MULTI_GPU=false to train on one GPU)")
    return 0, 1


def _rank0_print(rank):
    import builtins
    return builtins.print if rank == 0 else (lambda *a, **k: None)


def _pad_to_global(sp, world):
    """the collate functions pad a batch to ITS longest item (data/dataset.py:215-224); with the global batch
    sharded over ranks every rank pads to the longest item of the GLOBAL batch, so the mean-type losses keep the
    denominators a single process would use.  One all-reduce(max) of two integers; a no-op for fixed-shape batches."""
    import torch.distributed as dist
    if world == 1:
        return sp
    # The two integers travel as a HOST tensor when the back end can reduce one (gloo), else through the device: either way the
    # result is needed on the host to size the padding -- batches come from the prefetch thread as host tensors, so this is
    # the loader's synchronisation, not the training stream's (the captured step never passes through here).
    on_host = dist.get_backend() != "nccl"
    dims = torch.tensor([sp["data_0"].shape[-1], sp["data_1"].shape[-1]], dtype=torch.int64,
                        device="cpu" if on_host else torch.device("cuda", torch.cuda.current_device()))
    dist.all_reduce(dims, op=dist.ReduceOp.MAX)
    T, N = (int(v) for v in dims.tolist())
    out = dict(sp)
    pad = lambda t, n: torch.nn.functional.pad(t, (0, n - t.shape[-1])) if t.shape[-1] < n else t
    out["data_0"] = pad(sp["data_0"], T)
    out["data_1"] = pad(sp["data_1"], N)
    if "data_3" in sp:
        out["data_3"] = pad(sp["data_3"], 4 * T)
    return out


def text2id(text, vocabulary):
    """data/dataset.py:175-185: lower-case, map through VOCABULARY (the double quote folds into the single
    quote), append 'E'."""
    text = text.lower().replace('"', "'") + "E"
    table = {ch: i for i, ch in enumerate(vocabulary)}
    return [table[ch] for ch in text if ch in table]


class CorpusSource:
    """The reference's ``VCTKDataset`` + ``DataLoader(shuffle=True, collate_fn=collate_pad_*)`` (data/dataset.py:13-258,
    train/ordinary.py:199-200) over a corpus laid out as the reference expects: ``DATA_ROOT_DIR/data_path/ordinary/
    {wav,txt}.path.{train,validate}`` (one path per line; a wav path ends in ``pXXX/pXXX_NNN.wav``), one-line transcripts,
    ``SPK_EMB_DIR/pXXX.npy`` speaker codes.  Spectrograms come from the ``.npy`` cache of data/dataset.py:85-91; files that
    are not cached yet are extracted once, up front, on the GPU (``extract_features``).  Batches are zero-padded to their own
    longest item, as the collate functions do ('P' = id 0 pads the text)."""

    def __init__(self, cfg, step, pattern, mode, batch_size, spec_dir=None, seed=0, rank=0, world=1, stage=None):
        self.cfg, self.step, self.mode, self.B = cfg, step, mode, batch_size
        self.seed, self.rank, self.world, self.epoch = seed, rank, world, 0
        root = cfg["DATA_ROOT_DIR"]
        if pattern == "ubm-finetune":
            base, tag = os.path.join(root, "data_path", "ubm-finetune"), ".%s.%s" % (stage or "ubm", mode)
        else:
            base, tag = os.path.join(root, "data_path", "ordinary"), "." + mode
        with open(os.path.join(base, "wav.path" + tag)) as f:
            self.wavlist = [ln.strip() for ln in f if ln.strip()]
        with open(os.path.join(base, "txt.path" + tag)) as f:
            self.txtlist = [ln.strip() for ln in f if ln.strip()]
        if len(self.wavlist) != len(self.txtlist):
            raise RuntimeError("corpus lists differ in length: %d wav paths, %d transcripts" % (len(self.wavlist), len(self.txtlist)))
        self.cache = spec_dir or (os.path.join(cfg["SRC_ROOT_DIR"], "spec_cache") + os.sep)
        missing = [w for w in self.wavlist if not os.path.exists(self.cache + w[-17:-4] + "_mel.npy")]
        if missing:
            extract_features(missing, cfg, self.cache)

    def __len__(self):
        per = self.B * self.world
        return (len(self.wavlist) + per - 1) // per

    def _item(self, idx):
        key = self.wavlist[idx][-17:-4]
        out = {"data_0": torch.from_numpy(np.load(self.cache + key + "_mel.npy")).float()}
        if self.step == "train_ssrn":
            out["data_1"] = torch.from_numpy(np.load(self.cache + key + "_lin.npy")).float()
            return out
        with open(self.txtlist[idx]) as f:
            text = f.readline().strip()
        out["data_1"] = torch.tensor(text2id(text, self.cfg["VOCABULARY"]), dtype=torch.long).view(1, -1)
        spk = np.load(os.path.join(self.cfg["SPK_EMB_DIR"], self.wavlist[idx][-12:-8] + ".npy"))
        out["data_2"] = torch.from_numpy(np.asarray(spk, dtype=np.float32)).view(-1, 1)
        if self.step == "synthesize" and self.mode != "validate":      # data/dataset.py:131-132
            out["data_3"] = torch.from_numpy(np.load(self.cache + key + "_lin.npy")).float()
        return out

    @staticmethod
    def _pad_stack(items, key):
        width = max(it[key].shape[-1] for it in items)
        return torch.stack([torch.nn.functional.pad(it[key], (0, width - it[key].shape[-1])) for it in items], 0)

    def __iter__(self):
        n = len(self.wavlist)
        order = np.arange(n)
        if self.mode == "train":                                    # DataLoader(shuffle=True): a fresh permutation per epoch
            order = np.random.RandomState(self.seed + self.epoch).permutation(n)
        self.epoch += 1
        per = self.B * self.world
        for i in range(len(self)):
            if self.world > 1:
                # Data parallel: every rank must run the SAME number of iterations (each one ends in collectives) on FULL, equal
                # shards (gradients are averaged with 1/world; the penalty coefficients are drawn for B * world samples), so the
                # ragged last global batch wraps around to the head of this epoch's order instead of leaving some ranks short.
                idx = order[np.arange(i * per + self.rank * self.B, i * per + (self.rank + 1) * self.B) % n]
            else:
                idx = order[i * per:(i + 1) * per]                     # one process: the reference's partial last batch (drop_last=False)
            items = [self._item(int(k)) for k in idx]
            yield {key: (torch.stack([it[key] for it in items], 0) if key == "data_2" else self._pad_stack(items, key)) for key in items[0]}


class BatchSource:
    """Yields the dicts the reference's collate functions produce: data_0 mel (B,80,T), data_1 text (B,1,N)
    int64 or lin (B,513,4T), data_2 spk (B,200,1), data_3 lin (synthesis)."""

    def __init__(self, cfg, step, batch_size, spec_dir=None, seed=0, rank=0, world=1, pattern="conditional", mode="train"):
        self.cfg, self.step, self.B = cfg, step, batch_size
        self.rank, self.world, self.seed = rank, world, seed
        self.files = []
        self.corpus = None
        lists = os.path.join(cfg.get("DATA_ROOT_DIR", ""), "data_path", "ubm-finetune" if pattern == "ubm-finetune" else "ordinary")
        if os.path.isdir(lists):          # a corpus in the reference's layout: real batches (CorpusSource)
            self.corpus = CorpusSource(cfg, step, pattern, mode, batch_size, spec_dir, seed, rank, world)
        if spec_dir and os.path.isdir(spec_dir):
            self.files = sorted(glob.glob(os.path.join(spec_dir, "p*", "*_mel.npy")))
        self.n_synth = int(cfg.get("SYNTHETIC_BATCHES_PER_EPOCH", 8))
        self.vocab = len(cfg["VOCABULARY"]) - 1
        self.F = cfg["COARSE_MELSPEC"]["FREQ_BINS"]
        self.bins = 1 + cfg["STFT"]["FFT_LENGTH"] // 2

    def __len__(self):
        if self.corpus is not None:
            return len(self.corpus)
        if self.files:
            return max(1, len(self.files) // (self.B * self.world))
        return self.n_synth

    def _synthetic(self, i):
        seed = self.seed + 1000 * i + self.rank
        if self.step == "train_ssrn":
            mel, lin = train.synthetic_ssrn_batch(self.B, self.cfg["MAX_FRAME_NUM"], self.F, self.bins, seed)
            return {"data_0": mel, "data_1": lin}
        mel, text, spk = train.synthetic_text2mel_batch(self.B, self.cfg["MAX_TEXT_LEN"], self.cfg["MAX_FRAME_NUM"], self.F,
                                                        self.cfg["SPK_EMB_DIM"], self.vocab, seed)
        out = {"data_0": mel, "data_1": text, "data_2": spk}
        if self.step == "synthesize":
            out["data_3"] = train.synthetic_ssrn_batch(self.B, self.cfg["MAX_FRAME_NUM"], self.F, self.bins, seed)[1]
        return out

    def _cached(self, i):
        idx = [(i * self.world + self.rank) * self.B + j for j in range(self.B)]
        mels = [np.load(self.files[k % len(self.files)]) for k in idx]
        T = max(m.shape[1] for m in mels)
        mel = torch.zeros(self.B, self.F, T)
        for j, m in enumerate(mels):
            mel[j, :, :m.shape[1]] = torch.from_numpy(m)
        if self.step == "train_ssrn":
            lin = torch.zeros(self.B, self.bins, 4 * T)
            for j, k in enumerate(idx):
                l = np.load(self.files[k % len(self.files)].replace("_mel.npy", "_lin.npy"))
                lin[j, :, :l.shape[1]] = torch.from_numpy(l)
            return {"data_0": mel, "data_1": lin}
        raise RuntimeError("cached text2mel batches need the corpus transcripts; use synthetic batches")

    def __iter__(self):
        if self.corpus is not None:
            yield from self.corpus
            return
        for i in range(len(self)):
            yield self._cached(i) if (self.files and self.step == "train_ssrn") else self._synthetic(i)


class Prefetcher:
    """Iterate over a batch source with the host work (file reads / synthetic generation, padding) done one or two batches
    ahead on a background thread into pinned memory, and the host-to-device copies issued without blocking: the role of the
    reference's multi-worker DataLoader (train/ordinary.py:199-200), sized for one process per GPU.  Yields the same dicts,
    already on ``device``, in the same order."""

    def __init__(self, source, device, depth=2):
        self.source, self.device, self.depth = source, device, depth

    def __len__(self):
        return len(self.source)

    def __iter__(self):
        import queue
        import threading
        q = queue.Queue(maxsize=self.depth)
        pin = torch.cuda.is_available() and self.device.type == "cuda"
        stop = object()

        def work():
            try:
                for sp in self.source:
                    q.put({k: (v.pin_memory() if pin else v) for k, v in sp.items()})
                q.put(stop)
            except BaseException as e:          # surface loader errors in the consumer
                q.put(e)

        th = threading.Thread(target=work, daemon=True)
        th.start()
        while True:
            item = q.get()
            if item is stop:
                break
            if isinstance(item, BaseException):
                raise item
            yield {k: v.to(self.device, non_blocking=pin) for k, v in item.items()}
        th.join()


def _build(train_step, train_pattern, cfg, adversarial):
    if train_step == "train_text2mel":
        model = melSyn(vocab_len=len(cfg["VOCABULARY"]) - 1, condition=(train_pattern == "conditional"),
                       spkemb_dim=cfg["SPK_EMB_DIM"], textemb_dim=cfg["TEXT_EMB_DIM"],
                       freq_bins=cfg["COARSE_MELSPEC"]["FREQ_BINS"], hidden_dim=cfg["HIDDEN_DIM"])
        disc = melDisc(cfg["COARSE_MELSPEC"]["FREQ_BINS"], cfg["DISC_DIM"]) if adversarial else None
    else:
        model = SSRN(freq_bins=cfg["COARSE_MELSPEC"]["FREQ_BINS"], output_bins=1 + cfg["STFT"]["FFT_LENGTH"] // 2,
                     ssrn_dim=cfg["SSRN_DIM"])
        disc = linDisc(1 + cfg["STFT"]["FFT_LENGTH"] // 2, cfg["DISC_DIM"]) if adversarial else None
    return model, disc


def _adam(params, cfg, fused=True):
    a = cfg["ADAM"]
    if fused:
        return train.FusedAdam(params, a["ALPHA"], (a["BETA_1"], a["BETA_2"]), a["EPSILON"])
    return torch.optim.Adam(params, a["ALPHA"], (a["BETA_1"], a["BETA_2"]), a["EPSILON"])


def _free_run(model, text_id, spk_emb, frames, freq_bins, graph=False, incremental=False):
    """The reference's synthesis loop (synthesize.py:103-109, ordinary.py:59-65).  ``graph=True`` runs the same loop as
    a replayed hipGraph of one fixed-shape step (spoofsv_amd/synth.py; config key SYNTH_GRAPH): 1.5x faster at batch 1,
    same values up to the arithmetic mode of the first few frames (short prefixes run the exact-fp32 kernels step by step)."""
    resident.ensure(model, ops._stream())       # frozen weights: split once, not once per conv call (~30 launches per step)
    if incremental:       # config key SYNTH_INCREMENTAL (default on in synthesize / generate_test_utterances): one new column
        from . import synth       # per step instead of the whole prefix (spoofsv_amd/synth.py, IncrementalSynthesizer)
        return synth.free_run_incremental(model, text_id, spk_emb, frames)
    if graph:
        from . import synth
        return synth.free_run(model, text_id, spk_emb, frames)
    B = text_id.shape[0]
    dev = text_id.device
    init = torch.zeros((B, freq_bins, 1), device=dev)
    Y, A, pma, K, V = model(melspec=init, textid=text_id, spkemb=spk_emb, pma=torch.zeros((B,), device=dev).long())
    inputs = torch.cat((init, Y), dim=-1)
    for _ in range(frames - 1):
        Y, A, pma = model(melspec=inputs, textid=None, spkemb=spk_emb, K=K, V=V, A_last=A, pma=pma)
        inputs = torch.cat((inputs, Y[:, :, -1:]), dim=-1)
    return Y, A


def _save(path, payload):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    torch.save(payload, path)


def validate(loader, trainloader, gaw, cfg, model, train_step="train_text2mel"):
    """train/ordinary.py:46-128: mean loss over the validation loader and the loss of one training batch, the model in eval mode.
    Text2Mel is validated by FREE-RUNNING synthesis of as many frames as the ground truth has (:61-65), then the three training
    losses against it; SSRN by one forward pass (:80-85).  The free run goes through ``_free_run`` (column-incremental)."""
    dev = _device()
    F = cfg["COARSE_MELSPEC"]["FREQ_BINS"]

    def one(sp, tag):
        mel_gt = sp["data_0"].to(dev)
        if train_step == "train_text2mel":
            Y, A = _free_run(model, sp["data_1"].to(dev), sp["data_2"].to(dev), mel_gt.shape[-1], F,
                             incremental=cfg.get("SYNTH_INCREMENTAL", True))
            terms = train.text2mel_losses(Y, A, mel_gt, gaw)
        else:
            terms = ops.spec_losses(model(mel_gt), sp["data_1"].to(dev))
        terms = [float(t) for t in terms]
        print("{} set loss: {} {}".format(tag, " ".join(str(t) for t in terms), sum(terms)))
        return sum(terms)
    with torch.no_grad():
        total, n = 0.0, 0
        for sp in loader:
            total, n = total + one(sp, "val"), n + 1
        last = float("nan")
        for sp in trainloader:
            last = one(sp, "train")
            break
    return total / max(n, 1), last


def ordinary_train(train_step, train_pattern, cfg, spec_dir=None, resume_checkpoints=None, current_time=None):
    """Non-adversarial training, train/ordinary.py:130-293.  Under torchrun: data parallel, one rank per GPU (``_distributed``)."""
    rank, world = _distributed(cfg)
    print = _rank0_print(rank)                               # the reference prints from its one process: rank 0 here
    dev = _device()
    save_dir = os.path.join(cfg["SRC_ROOT_DIR"], "checkpoints", train_pattern, "not_adversarial", str(current_time))
    model, _ = _build(train_step, train_pattern, cfg, False)
    epoch = iteration = 0
    loss_val_log = []
    if resume_checkpoints is None:
        model.apply(train.init_weights)
        model.to(dev)
        opt = _adam(model.parameters(), cfg)
    else:
        ck = torch.load(resume_checkpoints, map_location="cpu")
        model.load_state_dict(ck["model_state_dict"])
        model.to(dev)
        opt = _adam(model.parameters(), cfg)
        opt.load_state_dict(ck["optimizer_state_dict"])
        epoch, iteration, loss_val_log = ck["epoch"], ck["iteration"], ck["loss_val_log"]
    model.train()
    ddp = None
    if world > 1:
        ddp = train.DataParallelRanks(model=model)           # gradient arena + bucketed all-reduce overlapped with backward
        if resume_checkpoints is None:
            ddp.broadcast_parameters(0)                      # what DataParallel's replicate does, once instead of per iteration
            opt.refresh_resident_weights()
    src = Prefetcher(BatchSource(cfg, train_step, cfg["BATCH_SIZE"], spec_dir, rank=rank, world=world, pattern=train_pattern), dev)
    val_src = BatchSource(cfg, train_step, 8, spec_dir, seed=7919, pattern=train_pattern, mode="validate")     # batch 8, :200
    if val_src.corpus is None:
        val_src.n_synth = int(cfg.get("SYNTHETIC_VALIDATION_BATCHES", 1))
    gaw = train.guided_attention_mat(cfg["MAX_TEXT_LEN"], cfg["MAX_FRAME_NUM"], device=dev)
    max_iter = cfg.get("MAX_ITERATIONS")
    history = []
    while epoch < cfg["MAX_EPOCHS"]:
        for i, sp in enumerate(src):
            t0 = time.time()
            sp = _pad_to_global(sp, world)
            mel_gt = sp["data_0"].to(dev)
            if train_step == "train_text2mel":
                l1, bd, la, att = train.text2mel_step(model, opt, mel_gt, sp["data_1"].to(dev), sp["data_2"].to(dev), gaw, ddp=ddp)
                terms = (l1, bd, la)
            else:
                terms = train.ssrn_step(model, opt, mel_gt, sp["data_1"].to(dev), ddp=ddp)
            if ddp is not None:
                terms = ddp.all_reduce_mean(*terms)          # the log shows the global-batch loss, as the reference's gathered outputs do
            terms = tuple(float(t.detach()) for t in terms)
            history.append(sum(terms))
            print("Iteration {}/{} for epoch {}, loss: {} {} global iteration {}".format(
                i + 1, len(src), epoch + 1, " ".join(str(t) for t in terms), sum(terms), iteration + 1))
            if iteration % cfg["VAL_EVERY_ITER"] == 0 and iteration > 0 and rank == 0:      # replicas are identical: rank 0 validates and saves
                model.eval()                                       # train/ordinary.py:264-267
                loss_val, loss_val_train = validate(val_src, src.source, gaw, cfg, model, train_step)
                model.train()
                loss_val_log.append(loss_val)
                print("Validation loss of No.{} validation: {} on validation set. {} on train set.".format(
                    iteration // cfg["VAL_EVERY_ITER"], loss_val, loss_val_train))
                payload = {"epoch": epoch + 1, "iteration": iteration + 1, "model_state_dict": model.state_dict(),
                           "optimizer_state_dict": opt.state_dict(), "loss_val_log": loss_val_log}
                _save(os.path.join(save_dir, "{}_iteration_{}.tar.pth".format(train_step[6:], iteration + 1)), payload)
                if loss_val_log.index(min(loss_val_log)) == len(loss_val_log) - 1:
                    _save(os.path.join(save_dir, "{}_best_model.tar.pth".format(train_step[6:])), payload)
            iteration += 1
            print("Time elapsed {}s".format(time.time() - t0))
            if max_iter is not None and iteration >= max_iter:
                return model, history
        epoch += 1
    return model, history


def adversarial_train(train_step, train_pattern, cfg, spec_dir=None, resume_checkpoints=None, current_time=None):
    """WGAN-GP training, train/adversarial_wasserstein_gp.py:148-450: one generator iteration, then RATIO critic
    iterations (:267); the generator runs on the HIP path in all of them (:278, :329), the critic on its twice-differentiable HIP ops
    (spoofsv_amd/critic.py).  Under torchrun: data parallel, one rank per GPU (``_distributed``; the reference's MULTI_GPU
    branch :183-196) -- both iteration kinds then run through ``train.AdversarialGraphStep`` with the gradient all-reduces
    overlapped with backward; so does a single rank with the optional config key CAPTURE_GRAPHS (hipGraph replay, needs
    fixed batch shapes)."""
    rank, world = _distributed(cfg)
    print = _rank0_print(rank)
    dev = _device()
    save_dir = os.path.join(cfg["SRC_ROOT_DIR"], "checkpoints", train_pattern, "adversarial", str(current_time))
    model, disc = _build(train_step, train_pattern, cfg, True)
    epoch = iteration = 0
    logs = {"wd_log": [], "loss_train_log_syn": [], "loss_train_log_syn_onlyfromD": [], "loss_train_log_disc": [], "loss_val_log": []}
    stepped = world > 1 or bool(cfg.get("CAPTURE_GRAPHS"))
    ck = None
    if resume_checkpoints is None:
        model.apply(train.init_weights)
        disc.apply(train.init_weights)
    else:
        ck = torch.load(resume_checkpoints, map_location="cpu")
        epoch, iteration = ck["epoch"], ck["iteration"]
        model.load_state_dict(ck["model_state_dict"])
        disc.load_state_dict(ck["disc_state_dict"])
        for k in logs:
            logs[k] = ck[k]
    model.to(dev)
    disc.to(dev)
    model.train()
    disc.train()
    src = Prefetcher(BatchSource(cfg, train_step, cfg["BATCH_SIZE"], spec_dir, rank=rank, world=world, pattern=train_pattern), dev)
    gaw = train.guided_attention_mat(cfg["MAX_TEXT_LEN"], cfg["MAX_FRAME_NUM"], device=dev)
    max_iter = cfg.get("MAX_ITERATIONS")
    val_src = BatchSource(cfg, train_step, 8, spec_dir, seed=7919, pattern=train_pattern, mode="validate")
    if val_src.corpus is None:
        val_src.n_synth = int(cfg.get("SYNTHETIC_VALIDATION_BATCHES", 1))
    if stepped:
        return _adversarial_train_stepped(train_step, cfg, dev, model, disc, src, gaw, save_dir, logs, max_iter, val_src, ck,
                                          epoch, iteration, rank, world)
    opt_syn = _adam(model.parameters(), cfg)
    opt_disc = _adam(disc.parameters(), cfg, fused=False)
    if ck is not None:
        opt_syn.load_state_dict(ck["opt_state_dict_syn"])
        opt_disc.load_state_dict(ck["opt_state_dict_disc"])
    while epoch < cfg["MAX_EPOCHS"]:
        for i, sp in enumerate(src):
            t0 = time.time()
            opt_syn.zero_grad(set_to_none=True)
            opt_disc.zero_grad(set_to_none=True)
            target = "D" if iteration % (cfg["RATIO"] + 1) else "G"
            mel_gt = sp["data_0"].to(dev)
            if train_step == "train_text2mel":
                gt = mel_gt
                pred, att = model(train.shift_right(mel_gt), sp["data_1"].to(dev), sp["data_2"].to(dev))
            else:
                gt = sp["data_1"].to(dev)
                pred, att = model(mel_gt), None
            B, C, T = gt.shape
            if target == "G":
                disc_syn = disc(pred)
                l1, bd = ops.spec_losses(pred, gt)
                base = l1 + bd
                if att is not None:
                    base = base + ops.guided_att_loss(att, gaw)
                loss_disc = torch.mean(-disc_syn)
                # adaptive weight of the critic term (:290): plain Python floats, as the reference's .item() calls
                loss = base + (float(base) / abs(float(loss_disc))) * loss_disc
                with ops.input_grads_only(disc):      # the critic's own parameter gradients of a G iteration are zeroed unread (:264-265)
                    loss.backward()
                opt_syn.step()
                logs["loss_train_log_syn"].append(float(loss))
                logs["loss_train_log_syn_onlyfromD"].append(float(loss_disc))
                print("training G  L1:{}, BD:{}, DISC:{}, ALL:{}".format(float(l1), float(bd), float(loss_disc), float(loss)))
            else:
                coeff = torch.rand(B).view(B, 1, 1).expand(B, C, T).to(dev)          # CPU RNG, as :300
                mid = (coeff * gt.detach() + (1 - coeff) * pred.detach()).requires_grad_(True)
                out_mid = disc(mid)
                with ops.input_grads_only(disc):      # this pass asks for d out / d mid only (ops.input_grads_only)
                    grads = torch.autograd.grad(outputs=out_mid, inputs=mid, grad_outputs=torch.ones_like(out_mid),
                                                retain_graph=True, create_graph=True)[0]
                loss_gp = torch.mean(cfg["LAMBDA"] * (torch.norm(grads, p=2, dim=(1, 2)) - 1) ** 2)
                loss_gp.backward()
                disc_real = disc(gt.detach())                                         # the reference's call order: ground truth, then prediction (:313-314)
                loss_D = torch.mean(disc(pred.detach()) - disc_real)
                loss_D.backward()
                opt_disc.step()
                logs["loss_train_log_disc"].append(float(loss_D) + float(loss_gp))
                logs["wd_log"].append(-float(loss_D))
                print("training D  DISC:{}, WD:{}".format(float(loss_D) + float(loss_gp), -float(loss_D)))
            if iteration % cfg["VAL_EVERY_ITER"] == 0 and iteration > 0:
                _adversarial_validate_and_save(train_step, cfg, model, disc, opt_syn, opt_disc, val_src, src.source, gaw, logs, save_dir,
                                               epoch, iteration, print)
            iteration += 1
            print("Time elapsed {}s.".format(time.time() - t0))
            if max_iter is not None and iteration >= max_iter:
                return model, disc, logs
        epoch += 1
    return model, disc, logs


def _adversarial_validate_and_save(train_step, cfg, model, disc, opt_syn, opt_disc, val_src, train_src, gaw, logs, save_dir, epoch, iteration, print):
    """train/adversarial_wasserstein_gp.py:392-437: validation pass, then the iteration checkpoint and, when the validation loss
    is the best so far, the best-model checkpoint (reference key names)."""
    model.eval()
    loss_val, loss_val_train = validate(val_src, train_src, gaw, cfg, model, train_step)
    model.train()
    logs["loss_val_log"].append(loss_val)
    print("Validation loss of No.{} validation: {} on validation set. {} on train set.".format(
        iteration // cfg["VAL_EVERY_ITER"], loss_val, loss_val_train))
    payload = {"epoch": epoch + 1, "iteration": iteration + 1, "model_state_dict": model.state_dict(),
               "disc_state_dict": disc.state_dict(), "opt_state_dict_syn": opt_syn.state_dict(),
               "opt_state_dict_disc": opt_disc.state_dict()}
    payload.update(logs)
    if logs["loss_val_log"].index(min(logs["loss_val_log"])) == len(logs["loss_val_log"]) - 1:      # :398-416
        _save(os.path.join(save_dir, "{}_best_model.tar.pth".format(train_step[6:])), payload)
    _save(os.path.join(save_dir, "{}_iteration_{}.tar.pth".format(train_step[6:], iteration + 1)), payload)


def _restore_adam(opt, saved):
    """Put a FusedAdam whose state tensors are baked into captured hipGraphs back to ``saved`` (a torch-style optimizer state
    dict, or None = a fresh optimizer) by writing INTO the existing tensors."""
    params = [p for g in opt.param_groups for p in g["params"]]
    step = 0
    for i, p in enumerate(params):
        st = opt.state.get(p)
        if not st:
            continue
        if saved is None:
            st["exp_avg"].zero_(); st["exp_avg_sq"].zero_()
        else:
            sv = saved["state"][i]
            st["exp_avg"].copy_(sv["exp_avg"]); st["exp_avg_sq"].copy_(sv["exp_avg_sq"])
            step = int(sv["step"])
    opt._steps = step
    if opt._step_dev is not None:
        opt._step_dev.fill_(step)


def _adversarial_train_stepped(train_step, cfg, dev, model, disc, src, gaw, save_dir, logs, max_iter, val_src, ck, epoch, iteration,
                               rank, world):
    """adversarial_train on ``train.AdversarialGraphStep``: both iteration kinds replayed from hipGraphs when the batch shapes
    are fixed (config key CAPTURE_GRAPHS with the synthetic source), run phase by phase otherwise; data parallel when
    ``world`` > 1.  Same schedule, losses, logs and checkpoints as the eager loop."""
    print = _rank0_print(rank)
    a = cfg["ADAM"]
    opt_syn = train.FusedAdam(model.parameters(), a["ALPHA"], (a["BETA_1"], a["BETA_2"]), a["EPSILON"], capturable=True)
    opt_disc = train.FusedAdam(disc.parameters(), a["ALPHA"], (a["BETA_1"], a["BETA_2"]), a["EPSILON"], capturable=True)
    ddp_syn = ddp_disc = None
    if world > 1:
        ddp_syn = train.DataParallelRanks(model=model)
        ddp_disc = train.DataParallelRanks(list(disc.parameters()))
        if ck is None:
            ddp_syn.broadcast_parameters(0)
            ddp_disc.broadcast_parameters(0)
        # The critics' dropout masks are Philox(seed = torch.cuda.initial_seed(), device-side call counter, element): without a
        # per-rank seed every rank would draw the SAME masks for its different shard, where the reference draws an independent
        # mask per sample of the global batch (models/discriminator.py:27-36).  Must precede the capture: the seed is frozen in it.
        torch.cuda.manual_seed(int(cfg.get("SEED", 0)) + 7919 * (rank + 1))
    graph = bool(cfg.get("CAPTURE_GRAPHS")) and not src.source.files and src.source.corpus is None
    w_model = {k: v.detach().clone() for k, v in model.state_dict().items()}
    w_disc = {k: v.detach().clone() for k, v in disc.state_dict().items()}
    kind = "text2mel" if train_step == "train_text2mel" else "ssrn"

    def pick(sp):
        sp = _pad_to_global(sp, world)
        keys = ("data_0", "data_1", "data_2") if kind == "text2mel" else ("data_0", "data_1")
        return [sp[k].to(dev) for k in keys]
    first = pick(next(iter(src.source)))          # shapes only; the prefetching iterator starts with the training loop
    stepper = train.AdversarialGraphStep(kind, model, disc, opt_syn, opt_disc, first, gaw, cfg["LAMBDA"], ddp_syn, ddp_disc, graph=graph,
                                         coeff_seed=cfg.get("SEED", 0))
    # capturing ran warm-up iterations: put weights and optimizer state back to the start of training (or to the checkpoint),
    # writing into the tensors the graphs hold, and re-split the resident weight planes the captured convolutions read
    model.load_state_dict(w_model)
    disc.load_state_dict(w_disc)
    _restore_adam(opt_syn, ck["opt_state_dict_syn"] if ck is not None else None)
    _restore_adam(opt_disc, ck["opt_state_dict_disc"] if ck is not None else None)
    if ck is not None and not opt_syn.state:      # nothing ran yet (eager phases): plain load
        opt_syn.load_state_dict(ck["opt_state_dict_syn"])
        opt_disc.load_state_dict(ck["opt_state_dict_disc"])
    opt_syn.refresh_resident_weights()
    opt_disc.refresh_resident_weights()
    while epoch < cfg["MAX_EPOCHS"]:
        for sp in src:
            t0 = time.time()
            stepper.load(pick(sp))
            if iteration % (cfg["RATIO"] + 1) == 0:
                l1, bd, la, ld, tot = [float(v) for v in stepper.g_step()]
                logs["loss_train_log_syn"].append(tot)
                logs["loss_train_log_syn_onlyfromD"].append(ld)
                print("training G  L1:{}, BD:{}, ATT:{}, DISC:{}, ALL:{}".format(l1, bd, la, ld, tot))
            else:
                ld, gp = [float(v) for v in stepper.d_step()]
                logs["loss_train_log_disc"].append(ld + gp)
                logs["wd_log"].append(-ld)
                print("training D  DISC:{}, WD:{}".format(ld + gp, -ld))
            if iteration % cfg["VAL_EVERY_ITER"] == 0 and iteration > 0 and rank == 0:      # replicas are identical: rank 0 validates and saves
                _adversarial_validate_and_save(train_step, cfg, model, disc, opt_syn, opt_disc, val_src, src.source, gaw, logs, save_dir,
                                               epoch, iteration, print)
            iteration += 1
            print("Time elapsed {}s.".format(time.time() - t0))
            if max_iter is not None and iteration >= max_iter:
                return model, disc, logs
        epoch += 1
    return model, disc, logs


def synthesize(pattern, cfg, spec_dir, current_time=None, texts=None, spk_emb=None, max_frames=None):
    """synthesize.py:41-147: Text2Mel free-running loop, SSRN, then the vocoder tail (max-normalise, power, Griffin-Lim
    with 64 iterations, de-emphasis, peak 0.75; spoofsv_amd.vocoder) written as S<k>_B1.wav like synthesize.py:147.
    Returns the list of (mel, lin, attention) numpy triples and stores them under SRC_ROOT_DIR/samples/<time>/.
    cfg["VOCODE"] = False skips the waveform; cfg["GRIFFIN_LIM_ITERS"] overrides the 64 of synthesize.py:144."""
    dev = _device()
    sample_dir = os.path.join(cfg["SRC_ROOT_DIR"], "samples", str(current_time))
    os.makedirs(sample_dir, exist_ok=True)
    m1 = melSyn(vocab_len=len(cfg["VOCABULARY"]) - 1, condition=(pattern == "conditional"), spkemb_dim=cfg["SPK_EMB_DIM"],
                textemb_dim=cfg["TEXT_EMB_DIM"], freq_bins=cfg["COARSE_MELSPEC"]["FREQ_BINS"], hidden_dim=cfg["HIDDEN_DIM"])
    m2 = SSRN(freq_bins=cfg["COARSE_MELSPEC"]["FREQ_BINS"], output_bins=1 + cfg["STFT"]["FFT_LENGTH"] // 2, ssrn_dim=cfg["SSRN_DIM"])
    for m, key in ((m1, "INFERENCE_TEXT2MEL_MODEL"), (m2, "INFERENCE_SSRN_MODEL")):
        path = cfg.get(key)
        if path and os.path.exists(path):
            m.load_state_dict(torch.load(path, map_location="cpu")["model_state_dict"])
        else:   # no trained checkpoint available offline: seeded random weights
            torch.manual_seed(1234)
            m.apply(train.init_weights)
        m.to(dev).eval()
    frames = max_frames or cfg["MAX_FRAME_NUM"]
    outs = []
    voc = None
    if cfg.get("VOCODE", True):
        from scipy.io import wavfile               # what librosa 0.7.0's output.write_wav calls (synthesize.py:147)
        from .vocoder import Vocoder
        voc = Vocoder(cfg["STFT"]["FFT_LENGTH"], cfg["STFT"]["HOP_LENGTH"], dev)
    syn_list = os.path.join(cfg.get("DATA_ROOT_DIR", ""), "data_path", "ordinary", "wav.path.synthesize")
    if texts is None and os.path.exists(syn_list):
        # synthesize.py:62,90-147 proper: the 'synthesize' split of the corpus in batches of 8 -- free run for as many frames as
        # the ground truth has, both models' losses against it, then the whole batch through the vocoder (S<k>_B<i>.wav)
        gaw = train.guided_attention_mat(cfg["MAX_TEXT_LEN"], cfg["MAX_FRAME_NUM"], device=dev)
        loader = CorpusSource(cfg, "synthesize", pattern, "synthesize", 8, spec_dir)
        with torch.no_grad():
            for i, sp in enumerate(loader):
                mel_gt, lin_gt = sp["data_0"].to(dev), sp["data_3"].to(dev)
                Y, A = _free_run(m1, sp["data_1"].to(dev), sp["data_2"].to(dev), mel_gt.shape[-1], cfg["COARSE_MELSPEC"]["FREQ_BINS"],
                                 graph=cfg.get("SYNTH_GRAPH", False), incremental=cfg.get("SYNTH_INCREMENTAL", True))
                t2m = [float(t) for t in train.text2mel_losses(Y, A, mel_gt, gaw)]
                print("syn set text2mel loss: {} {}".format(" ".join(str(t) for t in t2m), sum(t2m)))
                resident.ensure(m2, ops._stream())
                lin = m2(Y).contiguous()
                ss = [float(t) for t in ops.spec_losses(lin, lin_gt)]
                print("syn set ssrn loss: {} {}".format(" ".join(str(t) for t in ss), sum(ss)))
                wav = voc.spectrogram2wav(lin, cfg, n_iter=cfg.get("GRIFFIN_LIM_ITERS", 64)).cpu().numpy() if voc is not None else None
                for k in range(lin.shape[0]):
                    if wav is not None:
                        wavfile.write(os.path.join(sample_dir, "S{}_B{}.wav".format(k + 1, i + 1)), cfg["SAMPLING_RATE"], wav[k])
                    outs.append((Y[k].cpu().numpy(), lin[k].cpu().numpy(), A[k].cpu().numpy()))
        return outs
    if texts is None:
        with open(cfg["TTS_TEXTS"]) as f:
            texts = [ln.strip() for ln in f if ln.strip()][:1]
    with torch.no_grad():
        for k, text in enumerate(texts):
            ids = torch.tensor(text2id(text, cfg["VOCABULARY"]), dtype=torch.long, device=dev).view(1, 1, -1)
            spk = (spk_emb if spk_emb is not None else torch.full((1, cfg["SPK_EMB_DIM"], 1), 0.06)).to(dev).float()
            Y, A = _free_run(m1, ids, spk, frames, cfg["COARSE_MELSPEC"]["FREQ_BINS"], graph=cfg.get("SYNTH_GRAPH", False),
                             incremental=cfg.get("SYNTH_INCREMENTAL", True))
            resident.ensure(m2, ops._stream())
            lin = m2(Y)
            mel_np, lin_np, a_np = Y[0].cpu().numpy(), lin[0].cpu().numpy(), A[0].cpu().numpy()
            np.save(os.path.join(sample_dir, "S{}_mel.npy".format(k + 1)), mel_np)
            np.save(os.path.join(sample_dir, "S{}_lin.npy".format(k + 1)), lin_np)
            if voc is not None:
                wav = voc.spectrogram2wav(lin.contiguous(), cfg, n_iter=cfg.get("GRIFFIN_LIM_ITERS", 64),
                                          graph=cfg.get("SYNTH_GRAPH", False))
                wavfile.write(os.path.join(sample_dir, "S{}_B1.wav".format(k + 1)), cfg["SAMPLING_RATE"], wav[0].cpu().numpy())
            outs.append((mel_np, lin_np, a_np))
    return outs


def generate_test_utterances(cfg, current_time, eval_utt_num=20, speakers=None, texts=None, max_frames=None):
    """generate_test_utterances.py:56-139: for every speaker, synthesize the first ``eval_utt_num`` sentences of TTS_TEXTS as
    ONE batch (texts zero-padded to a common length, :67-72; the speaker code repeated, :105) -- Text2Mel free run for
    MAX_FRAME_NUM further steps (:108-116), SSRN (:120), then per utterance Griffin-Lim(64), de-emphasis, trim(30 dB), clip to
    9 s, peak 0.75 and ``s<id>/s<id>_<nnn>.wav`` (:128-139).  Here the vocoder runs once per speaker on the whole batch
    (spoofsv_amd.vocoder); only trim / clip / write stay per utterance on the host.  The Kaldi / GE2E / ASVspoof trial-list
    bookkeeping of :141-260 is not reproduced.  ``speakers``: {name: (SPK_EMB_DIM,) array}; default: the .npy files of
    SPK_EMB_DIR.  Returns {speaker: [wav paths]}."""
    from scipy.io import wavfile
    from .vocoder import Vocoder, trim_silence
    dev = _device()
    if texts is None:
        with open(cfg["TTS_TEXTS"]) as f:
            texts = [ln.strip() for ln in f if ln.strip()]
    texts = texts[:eval_utt_num]
    ids = [text2id(t, cfg["VOCABULARY"]) for t in texts]
    width = max(len(i) for i in ids)
    text_id = torch.tensor([list(i) + [0] * (width - len(i)) for i in ids], dtype=torch.long, device=dev).view(len(ids), 1, width)
    m1 = melSyn(vocab_len=len(cfg["VOCABULARY"]) - 1, condition=True, spkemb_dim=cfg["SPK_EMB_DIM"], textemb_dim=cfg["TEXT_EMB_DIM"],
                freq_bins=cfg["COARSE_MELSPEC"]["FREQ_BINS"], hidden_dim=cfg["HIDDEN_DIM"])
    m2 = SSRN(freq_bins=cfg["COARSE_MELSPEC"]["FREQ_BINS"], output_bins=1 + cfg["STFT"]["FFT_LENGTH"] // 2, ssrn_dim=cfg["SSRN_DIM"])
    for m, key in ((m1, "INFERENCE_TEXT2MEL_MODEL"), (m2, "INFERENCE_SSRN_MODEL")):
        path = cfg.get(key)
        if path and os.path.exists(path):
            m.load_state_dict(torch.load(path, map_location="cpu")["model_state_dict"])
        else:   # no trained checkpoint available offline: seeded random weights
            torch.manual_seed(1234)
            m.apply(train.init_weights)
        m.to(dev).eval()
    if speakers is None:
        d = cfg["SPK_EMB_DIR"]
        speakers = {f[:-4]: np.load(os.path.join(d, f)) for f in sorted(os.listdir(d)) if f.endswith(".npy")}
    voc = Vocoder(cfg["STFT"]["FFT_LENGTH"], cfg["STFT"]["HOP_LENGTH"], dev)
    save_dir = os.path.join(cfg["SRC_ROOT_DIR"], "test", str(current_time), "spoof_data")
    frames = (max_frames or cfg["MAX_FRAME_NUM"]) + 1                   # first frame + MAX_FRAME_NUM further steps (:110-116)
    sr, out = cfg["SAMPLING_RATE"], {}
    with torch.no_grad():
        for spk, emb in speakers.items():
            e = torch.as_tensor(np.asarray(emb, dtype=np.float32), device=dev).view(1, -1, 1).expand(len(ids), -1, -1).contiguous()
            Y, _ = _free_run(m1, text_id, e, frames, cfg["COARSE_MELSPEC"]["FREQ_BINS"], graph=cfg.get("SYNTH_GRAPH", False),
                             incremental=cfg.get("SYNTH_INCREMENTAL", True))
            resident.ensure(m2, ops._stream())
            lin = m2(Y).contiguous()
            wav = voc.spectrogram2wav(lin, cfg, n_iter=cfg.get("GRIFFIN_LIM_ITERS", 64), graph=cfg.get("SYNTH_GRAPH", False),
                                      peak=None).cpu().numpy()
            sdir = os.path.join(save_dir, "s" + spk[1:])
            os.makedirs(sdir, exist_ok=True)
            paths = []
            for k in range(len(ids)):
                y, _ = trim_silence(wav[k], 30)
                y = y[:9 * sr]
                if len(y):
                    y = (y / np.max(y) * 0.75).astype(np.float32)
                path = os.path.join(sdir, "s{}_{}.wav".format(spk[1:], str(k + 1).zfill(3)))
                wavfile.write(path, sr, y)
                paths.append(path)
            out[spk] = paths
    return out


def extract_features(wav_paths, cfg, spec_dir):
    """data/dataset.py:94-123 for a list of wav files: read, trim leading/trailing silence (22 dB), then pre-emphasis, |STFT|,
    mel projection, normalisation and time reduction on the GPU (spoofsv_amd.vocoder.Vocoder.wav2spectrogram), and the
    reference's cache files ``<spec_dir>/<pXXX>/<pXXX_NNN>_{mel,lin}.npy`` (:85-91,120-123) that ``BatchSource`` reads.
    File decoding is ``scipy.io.wavfile`` (PCM or float wav at its native rate, as ``librosa.load(sr=None)`` returns it;
    resampling, metagen.py:29-62, is not reproduced).  Returns the list of (mel, lin) shapes written."""
    from scipy.io import wavfile
    from .vocoder import Vocoder, trim_silence
    dev = _device()
    voc = Vocoder(cfg["STFT"]["FFT_LENGTH"], cfg["STFT"]["HOP_LENGTH"], dev)
    shapes = []
    for path in wav_paths:
        sr, y = wavfile.read(path)
        if y.ndim > 1:
            y = y.mean(axis=1)                                   # librosa.load(mono=True)
        if y.dtype.kind in "iu":
            y = y.astype(np.float32) / float(1 << (8 * y.dtype.itemsize - 1))
        y, _ = trim_silence(y.astype(np.float32), 22)
        mel, lin = voc.wav2spectrogram(torch.from_numpy(np.ascontiguousarray(y)).to(dev), sr, cfg)
        key = path[-17:-4]                                       # 'pXXX/pXXX_NNN', data/dataset.py:85
        os.makedirs(os.path.join(spec_dir, os.path.dirname(key)), exist_ok=True)
        np.save(os.path.join(spec_dir, key + "_mel.npy"), mel.cpu().numpy())
        np.save(os.path.join(spec_dir, key + "_lin.npy"), lin.cpu().numpy())
        shapes.append((tuple(mel.shape), tuple(lin.shape)))
    return shapes
