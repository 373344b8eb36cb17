"""Training harness pieces on the HIP hot path: initialisation, guided-attention weights, fused Adam,
synthetic VCTK-shaped batches, the per-iteration train steps of the reference's trainers and the
one-process-per-GPU data-parallel wrapper.

What each piece replaces in the reference:
  init_weights           train/ordinary.py:16-19
  guided_attention_mat   train/ordinary.py:21-28 (Python double loop -> one vectorised float64 pass)
  FusedAdam              optim.Adam(params, ALPHA, (BETA_1, BETA_2), EPSILON), train/ordinary.py:182
  text2mel_step          train/ordinary.py:221-238 (teacher-forced forward, 3 losses, backward, Adam)
  ssrn_step              train/ordinary.py:240-254
  DataParallelRanks      nn.DataParallel (train/ordinary.py:165-173) -> one rank per GPU, RCCL all-reduce
"""
import ctypes
import os
import contextlib
import gc

import numpy as np
import torch
import torch.distributed as dist

from . import _lib, ops
from . import resident as _resident
from .ops import _p, _stream


def init_weights(layer):
    """He-normal on every weight with more than one dimension (train/ordinary.py:16-19)."""
    if hasattr(layer, "weight"):
        if len(layer.weight.shape) > 1:
            torch.nn.init.kaiming_normal_(layer.weight, nonlinearity="relu")


def guided_attention_mat(max_text_len, max_frame_num, device=None, g=0.2):
    """W[n, t] = 1 - exp(-(t/T - n/N)^2 / (2 g^2)) (train/ordinary.py:21-28), evaluated in float64 and
    rounded once to float32 -- the same values the reference's per-entry Python arithmetic stores."""
    n = torch.arange(max_text_len, dtype=torch.float64).unsqueeze(1) / max_text_len
    t = torch.arange(max_frame_num, dtype=torch.float64).unsqueeze(0) / max_frame_num
    W = (1 - torch.exp(-(t - n) ** 2 / (2 * g * g))).float()
    return W.to(device) if device is not None else W


# --------------------------------------------------------------------------------------------- Adam
_CHUNK = 32768



@contextlib.contextmanager
def _no_gc():
    """No cyclic garbage collection while a stream capture is in progress.  A collection can start at any allocation, in any thread
    (autograd's backward thread included), and finalise objects of EARLIER captures -- a CUDAGraph and its private pool kept alive by a
    reference cycle -- whose destructors free device memory: not allowed during a capture, and an error thrown in a destructor
    aborts the process (seen once in the GPU suite: 'Fatal Python error: Aborted ... Garbage-collecting' inside a backward
    under capture).  Collect first, then hold the collector off until the capture has ended."""
    gc.collect()
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()

class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (no weight decay / amsgrad, as the reference uses it), all parameters
    updated by ONE multi-tensor kernel launch (ssv_adam_multi).  State keys match torch's Adam
    (``step``, ``exp_avg``, ``exp_avg_sq``) so optimizer state dicts interchange.

    With ``capturable=True`` the step count lives on the device and advances inside the kernel, so a
    captured hipGraph of the whole training step replays correctly.
    """

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, capturable=False, resident=True,
                 weight_decay=0, amsgrad=False, maximize=False):
        if weight_decay != 0 or amsgrad or maximize:
            raise ValueError("FusedAdam implements plain Adam as the reference uses it (train/ordinary.py:182): "
                             "weight_decay, amsgrad and maximize are not supported")
        # the full set of torch.optim.Adam defaults, so that a saved state dict loads into torch.optim.Adam (the reference's
        # -R resume, train/ordinary.py:188-197) and steps there
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False, maximize=False,
                                      foreach=None, capturable=False, differentiable=False, fused=None,
                                      decoupled_weight_decay=False))
        self.capturable = capturable
        self._table = None
        self._host_tables = None
        self._host_flip = 0
        self._copied = None
        self._key = None
        self._step_dev = None
        self._steps = 0
        self._resident = None
        self.resident = resident           # False for models that never reach the HIP conv path

    def refresh_resident_weights(self):
        """(Re)write the resident pre-split planes of every conv weight this optimizer owns (one launch; see
        ``spoofsv_amd.resident``).  ``step`` does it after every update; call it once before the first forward
        -- and after anything that writes weights behind autograd's back -- so that step too finds current planes."""
        if not self.resident:
            return
        if self._resident is None:
            self._resident = _resident.ResidentWeights([p for g in self.param_groups for p in g["params"]])
        self._resident.refresh(_stream())

    def _build(self, plist):
        key = tuple((p.data_ptr(), p.grad.data_ptr()) for p in plist)
        if key == self._key:
            return
        rows = []
        for p in plist:
            st = self.state[p]
            if "exp_avg" not in st:
                st["step"] = torch.tensor(0.0)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            if not (p.is_contiguous() and p.grad.is_contiguous() and p.dtype == torch.float32 and p.grad.dtype == torch.float32):
                raise RuntimeError("FusedAdam needs dense float32 parameters and gradients")
            n = p.numel()
            for off in range(0, n, _CHUNK):
                m = min(_CHUNK, n - off)
                rows.append((p.data_ptr() + 4 * off, p.grad.data_ptr() + 4 * off, st["exp_avg"].data_ptr() + 4 * off,
                             st["exp_avg_sq"].data_ptr() + 4 * off, m))
        arr = np.array(rows, dtype=np.int64)
        assert ctypes.sizeof(_lib.AdamChunk) == 40
        # Pinned staging buffer + async copy, both allocated once (sized for every parameter): refilling
        # them is legal while a hipGraph is being captured (the copy becomes a memcpy node).
        # Two pinned tables used alternately, and an event after each copy: a rebuild (gradient addresses can move between
        # eager iterations) must not overwrite a pinned table whose asynchronous copy has not executed yet.
        if self._host_tables is None:
            cap = sum((p.numel() + _CHUNK - 1) // _CHUNK for g in self.param_groups for p in g["params"])
            self._host_tables = [torch.empty((cap, 5), dtype=torch.int64).pin_memory() for _ in range(2)]
            self._copied = [None, None]
            self._table = torch.empty((cap, 5), dtype=torch.int64, device=plist[0].device)
        self._host_flip ^= 1
        host, ev = self._host_tables[self._host_flip], self._copied[self._host_flip]
        capturing = torch.cuda.is_current_stream_capturing()
        if ev is not None and not capturing:
            ev.synchronize()
        host[:len(rows)].copy_(torch.from_numpy(arr))
        self._table.copy_(host, non_blocking=True)
        if not capturing:
            ev = torch.cuda.Event()
            ev.record()
            self._copied[self._host_flip] = ev
        self._nchunks = len(rows)
        self._key = key

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        if len(self.param_groups) != 1:
            raise RuntimeError("FusedAdam keeps one device-side step counter: use one parameter group")
        for group in self.param_groups:
            plist = [p for p in group["params"] if p.grad is not None]
            if not plist:
                continue
            ops._dev(plist[0], "parameter")
            self._build(plist)
            self._steps += 1
            b1, b2 = group["betas"]
            step_dev = None
            if self.capturable:
                if self._step_dev is None:
                    self._step_dev = torch.full((1,), self._steps - 1, dtype=torch.int32, device=plist[0].device)
                step_dev = _p(self._step_dev)
            _lib.call("ssv_adam_multi", _p(self._table), self._nchunks, float(group["lr"]), float(b1), float(b2),
                      float(group["eps"]), self._steps, step_dev, _stream())
        self.refresh_resident_weights()
        return loss

    # The per-parameter ``step`` entries of torch's Adam state are only materialised when the state is exported (one
    # host-side tensor op per parameter per iteration would cost more host time than the whole launch), and read back
    # on import so that a resumed run continues the bias correction where the checkpoint left it.
    def state_dict(self):
        if self._step_dev is not None:
            self._steps = int(self._step_dev.item())       # graph replays advance only the device counter
        for st in self.state.values():
            if "exp_avg" in st:
                st["step"] = torch.tensor(float(self._steps))
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        steps = [int(st["step"]) for st in self.state.values() if "step" in st]
        self._steps = max(steps) if steps else 0
        if self._step_dev is not None:
            self._step_dev.fill_(self._steps)
        self._key = None                      # moment tensors were replaced: rebuild the chunk table


# --------------------------------------------------------------------------------------------- data
def synthetic_text2mel_batch(B, N=186, T=325, freq_bins=80, spk_dim=200, vocab=34, seed=0, device=None):
    """VCTK-shaped synthetic batch: mel in (0,1), ids in [2, vocab) ending in
    'E'(1) and 'P'(0) as data/dataset.py:175-185 pads them, speaker codes in the 0.04-0.09 range of
    spk_emb/*.npy."""
    g = torch.Generator().manual_seed(seed)
    mel = torch.rand(B, freq_bins, T, generator=g)
    text = torch.randint(2, vocab, (B, 1, N), generator=g)
    text[:, :, -2] = 1
    text[:, :, -1] = 0
    spk = 0.04 + 0.05 * torch.rand(B, spk_dim, 1, generator=g)
    if device is not None:
        mel, text, spk = mel.to(device), text.to(device), spk.to(device)
    return mel, text, spk


def synthetic_ssrn_batch(B, T=325, freq_bins=80, out_bins=513, seed=0, device=None):
    g = torch.Generator().manual_seed(seed)
    mel = torch.rand(B, freq_bins, T, generator=g)
    lin = torch.rand(B, out_bins, 4 * T, generator=g)
    if device is not None:
        mel, lin = mel.to(device), lin.to(device)
    return mel, lin


def shift_right(mel_gt):
    """Teacher forcing input: [0 | mel[:, :, :-1]] (train/ordinary.py:226)."""
    if mel_gt.is_cuda:          # one HIP launch that also leaves the result's operand scale list (ops.shift_right)
        return ops.shift_right(mel_gt)
    return torch.cat((torch.zeros_like(mel_gt[:, :, :1]), mel_gt[:, :, :-1]), dim=-1)          # host tensors (the tests' oracle inputs)


# --------------------------------------------------------------------------------------------- DDP
class DataParallelRanks:
    """One process per GPU; utterances are sharded by rank and gradients are averaged with flat all-reduces over RCCL/xGMI
    (backend "nccl" on ROCm; "gloo" in the CPU tests).

    The reference's nn.DataParallel (train/ordinary.py:165-173) re-broadcasts all parameters every iteration and reduces
    gradients to GPU 0; here replicas stay in sync because every rank applies the same averaged gradient, so the only
    traffic is the gradient all-reduce (96.3 MB for Text2Mel).  xGMI is point-to-point: a few large messages keep the links
    busy, per-tensor all-reduces (214 tensors) would be latency bound.

    Two modes:
      * ``DataParallelRanks(model=m)`` -- gradient ARENA (gradarena.py): the backward kernels write parameter gradients
        straight into one flat buffer laid out bucket by bucket in the order backward finishes them (``tts.ddp_plan``).
        ``start_bucket(i)`` hands bucket i to an asynchronous all-reduce as soon as the backward segment producing it has
        been enqueued, so the exchange overlaps the rest of backward (``SegmentedBackward``); no packing copy.
      * ``DataParallelRanks(params)`` -- any parameter list (the critics, 0.5-0.7 MB): gradients are packed into persistent
        flat buckets after backward, all-reduced, and ``p.grad`` re-pointed at the averaged slices.
    Gradients are SUMMED over ranks: callers seed backward with ``grad_scale`` (= 1/world) so the sum is the global-batch
    mean (exact for power-of-two worlds); the legacy ``all_reduce_grads`` scales after the sum instead.
    """

    def __init__(self, params=None, bucket_mb=64, group=None, model=None, segmented=True):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # Collectives are issued when there is somebody to talk to -- or, with SSV_FORCE_COLLECTIVES=1 and an initialised
        # process group of ONE rank, regardless: every all-reduce / broadcast of the multi-GPU step then really goes through the
        # back end (RCCL on one MI355X) with results that cannot change, which is how the launch structure of the N > 1 step
        # (asynchronous collectives between hipGraph replays) is rehearsed on a single GPU (tests/test_gpu_ddp.py).
        self.collectives = self.world > 1 or (dist.is_initialized() and os.environ.get("SSV_FORCE_COLLECTIVES") == "1")
        self.grad_scale = 1.0 / self.world
        self.bucket_elems = max(1, int(bucket_mb * 1024 * 1024 // 4))
        self._flat = None
        self._works = []
        self.arena = None
        self.cut_names = []
        if model is not None:
            from . import gradarena, tts
            plan, cuts = tts.ddp_plan(model)
            if not segmented:
                plan, cuts = [("all", [g for _, gs in plan for g in gs])], []
            self.arena = gradarena.GradArena(plan)
            self.cut_names = cuts
            self.params = list(self.arena.params)
            self.bucket_params = [[p for g in gs for p in g] for _, gs in plan]
        else:
            self.params = [p for p in params if p.requires_grad]

    def close(self):
        """Teardown: outstanding collectives awaited, the gradient arena detached from the parameters (hooks and slots), so the
        same model can be wrapped again or trained without the wrapper.  Also runs when the wrapper is garbage-collected."""
        self.finish()
        if self.arena is not None:
            self.arena.release()

    def __del__(self):
        try:
            if self.arena is not None:
                self.arena.release()
        except Exception:
            pass

    @property
    def n_buckets(self):
        return len(self.arena.ranges) if self.arena is not None else 1

    def broadcast_parameters(self, src=0):
        """Make every replica start from rank `src`'s weights (what DataParallel's replicate does)."""
        if not self.collectives:
            return
        with torch.no_grad():
            for p in self.params:
                dist.broadcast(p.data, src, group=self.group)
        _resident.invalidate(self.params)      # p.data writes do not bump the version the resident planes are checked against

    # ---- arena mode -------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def adopt_bucket(self, i):
        """End of backward segment ``i`` (device work: belongs inside the captured phase): every gradient of bucket ``i``
        must live in the arena.  The fused operators wrote them there; anything else is copied in."""
        self.arena.adopt(self.bucket_params[i])

    @torch.no_grad()
    def start_bucket(self, i):
        """Backward has produced every gradient of bucket ``i`` (enqueued on the current stream): start the bucket's
        all-reduce (asynchronous: RCCL runs it on its own stream behind an event on the current one)."""
        if self.collectives:
            self._works.append(dist.all_reduce(self.arena.bucket(i), op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Make the current stream wait for every collective started since the last ``finish``."""
        for w in self._works:
            w.wait()
        self._works = []

    # ---- packed mode ------------------------------------------------------------------------------------------------
    def _buckets(self, grads):
        buckets, cur, n = [], [], 0
        for p, g in zip(self.params, grads):
            if g is None:
                continue
            if n + g.numel() > self.bucket_elems and cur:
                buckets.append(cur)
                cur, n = [], 0
            cur.append((p, g))
            n += g.numel()
        if cur:
            buckets.append(cur)
        return buckets

    @torch.no_grad()
    def pack(self, extra=()):
        """Pack every ``p.grad`` (and the scalars ``extra``, e.g. losses to be averaged for the log) into the persistent flat
        buckets; returns the handle for ``exchange`` / ``unpack``."""
        grads = [p.grad for p in self.params]
        buckets = self._buckets(grads)
        tail = [e.detach().reshape(1).float() for e in extra]
        sizes = [sum(g.numel() for _, g in b) for b in buckets]
        if tail:
            if not buckets:
                buckets, sizes = [[]], [0]
            sizes[-1] += len(tail)
        if self._flat is None or [f.numel() for f in self._flat] != sizes:
            dev = (grads[0] if grads and grads[0] is not None else tail[0]).device
            self._flat = [torch.empty(n, dtype=torch.float32, device=dev) for n in sizes]
        for j, (flat, bucket) in enumerate(zip(self._flat, buckets)):
            parts = [g.reshape(-1) for _, g in bucket] + (tail if j == len(buckets) - 1 else [])
            torch.cat(parts, out=flat)
        return buckets, len(tail)

    def exchange(self):
        """All-reduce (sum) the packed buckets; asynchronous, ``finish`` waits."""
        if self.collectives:
            for flat in self._flat:
                self._works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    @torch.no_grad()
    def unpack(self, handle, scale=None):
        """Re-point every ``p.grad`` at its slice of the (now reduced) bucket -- no copy back, and the optimizer's pointer
        table stays stable from step to step.  Returns the reduced ``extra`` scalars."""
        buckets, ntail = handle
        for flat, bucket in zip(self._flat, buckets):
            if scale is not None:
                flat.mul_(scale)
            off = 0
            for p, g in bucket:
                n = g.numel()
                p.grad = flat[off:off + n].view_as(g)
                off += n
        last = self._flat[-1]
        return [last[last.numel() - ntail + j] for j in range(ntail)]

    # ---- legacy one-shot API (gradients NOT pre-scaled) ----------------------------------------------------------------
    @torch.no_grad()
    def all_reduce_grads(self, grads=None):
        """Average gradients over ranks after a complete backward."""
        if not self.collectives:
            return
        if self.arena is not None:
            for i in range(self.n_buckets):
                self.adopt_bucket(i)
                self.start_bucket(i)
            self.finish()
            self.arena.flat.mul_(1.0 / self.world)
            return
        if grads is not None:
            for p, g in zip(self.params, grads):
                p.grad = g
        h = self.pack()
        self.exchange()
        self.finish()
        self.unpack(h, 1.0 / self.world)

    @torch.no_grad()
    def all_reduce_mean(self, *scalars):
        """Average loss scalars over ranks (global-batch semantics for the adaptive critic weight,
        train/adversarial_wasserstein_gp.py:290)."""
        if not self.collectives:
            return scalars
        v = torch.stack([s.detach().reshape(()) for s in scalars])
        dist.all_reduce(v, op=dist.ReduceOp.SUM, group=self.group)
        v /= self.world
        return tuple(v[i] for i in range(len(scalars)))

    @torch.no_grad()
    def all_reduce_mean_(self, vec):
        """In-place variant on a persistent device vector (the form used between captured hipGraphs)."""
        if self.collectives:
            dist.all_reduce(vec, op=dist.ReduceOp.SUM, group=self.group)
            vec.mul_(1.0 / self.world)
        return vec


class _cuts_installed:
    """Install a ``Cuts`` as the model's ``_cut`` hook for the duration of ONE forward call only: a forward run outside a
    segmented step must record an uncut tape."""

    def __init__(self, model, cuts):
        self.model, self.cuts = model, cuts

    def __enter__(self):
        self.prev = getattr(self.model, "_cut", None)
        if self.prev is not None and self.cuts.names:
            self.model._cut = self.cuts

    def __exit__(self, *exc):
        if self.prev is not None:
            self.model._cut = self.prev
        return False


class Cuts:
    """The ``_cut`` hook of ``tts.melSyn`` / ``tts.SSRN`` for a segmented backward: at every named cut the activation is
    detached (so a ``backward`` call stops there) and the pair (tensor on the tape, detached leaf) is remembered."""

    def __init__(self, names):
        self.groups = [[n] if isinstance(n, str) else list(n) for n in names]      # cuts that end the same backward segment
        self.names = [n for g in self.groups for n in g]
        self.rec = {}

    def reset(self):
        self.rec = {}

    def __call__(self, name, x):
        if name not in self.names or not x.requires_grad:
            return x
        leaf = x.detach().requires_grad_(True)
        if hasattr(x, "_ssv_amax"):              # the producer's operand-scale list (ops.amax_of) stays valid: same storage, same version
            leaf._ssv_amax = x._ssv_amax
        self.rec[name] = (x, leaf)
        return leaf


def backward_segments(cuts, roots, side_roots=None, ddp=None, defer=None):
    """The backward pass as a list of closures, one per segment.  Segment 0 differentiates ``roots`` (``[(tensor, grad)]``,
    grad None = 1) down to the nearest cuts; segment i continues from cut ``cuts.names[i-1]`` with the gradient that arrived
    at its leaf, plus the ``side_roots[name]`` that join the tape there (the attention loss of Text2Mel).  Every parameter
    of gradient bucket i (tts.ddp_plan) is final once segment i has run."""
    side_roots = side_roots or {}

    def run(pairs):
        ts = [t for t, _ in pairs]
        gs = [g for _, g in pairs]
        if defer is None:
            torch.autograd.backward(ts, gs)
            return
        with defer:                       # weight gradients of this segment: queued, then one launch per layer shape
            torch.autograd.backward(ts, gs)
        defer.flush()

    arena = ddp is not None and ddp.arena is not None

    def seg0():
        run([(t, g if g is not None else torch.ones_like(t)) for t, g in roots])
        if arena:
            ddp.adopt_bucket(0)
    segs = [seg0]
    for i, group in enumerate(cuts.groups):
        def seg(group=group, i=i):
            pairs = []
            for name in group:
                if name in cuts.rec:
                    x, leaf = cuts.rec[name]
                    pairs.append((x, leaf.grad))
                    pairs += [(t, g if g is not None else torch.ones_like(t)) for t, g in side_roots.get(name, [])]
            run(pairs)
            for name in group:
                if name in cuts.rec:
                    cuts.rec[name][1].grad = None
            if arena:
                ddp.adopt_bucket(i + 1)
        segs.append(seg)
    return segs


class PhasedStep:
    """A training iteration as a list of phases ``(kind, fn)``: ``"graph"`` phases are device work only (kernel launches
    through libssv_hip / torch) and are captured into hipGraphs -- consecutive ones into one graph, all graphs sharing one
    memory pool because activations recorded in an earlier phase are consumed by later ones -- while ``"eager"`` phases
    (RCCL collectives, which stay outside captures) are called between the replays.  ``graph=False`` runs everything
    eagerly, same order."""

    def __init__(self, phases, graph=True, warmup=2, drain=None):
        self.phases = phases
        self.use_graph = graph
        self.plan = None
        self.warmup = warmup
        self.drain = drain        # called after every eager phase WHILE CAPTURING: waits for the collectives it started

    def _eager(self):
        for _, fn in self.phases:
            fn()

    def prepare(self):
        if not self.use_graph or self.plan is not None:
            return self
        # The warm-up runs on a side stream and the capture on torch's capture stream: the parameters' AccumulateGrad nodes
        # are created on the first and used on the second, which autograd reports on every backward call.  Intended here.
        if hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
            torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(self.warmup):
                self._eager()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        pool = torch.cuda.graph_pool_handle()
        plan, i = [], 0
        while i < len(self.phases):
            kind, fn = self.phases[i]
            if kind == "eager":
                fn()
                # nothing of this collective may still be running when the next capture begins: communication back ends
                # work from threads of their own (gloo copies and synchronises there; RCCL's watchdog polls events)
                if self.drain is not None:
                    self.drain()
                torch.cuda.synchronize()
                plan.append(fn)
                i += 1
                continue
            j = i
            while j < len(self.phases) and self.phases[j][0] == "graph":
                j += 1
            g = torch.cuda.CUDAGraph()
            # thread_local: only THIS thread's calls are checked against the capture; the communication back end's helper
            # threads (see above) may touch the device while a later iteration's capture is in progress
            with _no_gc(), torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local"):
                for _, f in self.phases[i:j]:
                    f()
            plan.append(g.replay)
            self._graphs = getattr(self, "_graphs", []) + [g]
            i = j
        self.plan = plan
        return self

    def run(self):
        if self.plan is None:
            self._eager()
        else:
            for f in self.plan:
                f()


# --------------------------------------------------------------------------------------------- steps
def text2mel_losses(pred, att, mel_gt, gaw):
    """(l1, bin_div, att) of train/ordinary.py:230-236 on the HIP loss kernels."""
    l1, bd = ops.spec_losses(pred, mel_gt)
    return l1, bd, ops.guided_att_loss(att, gaw)


class TrainStep:
    """One non-adversarial optimizer iteration (train/ordinary.py:221-254) of ``kind`` "text2mel" (batch = mel, text, spk) or
    "ssrn" (batch = mel, lin): forward, the reference's losses, backward, Adam.

    Single rank: one phase (one hipGraph when ``graph``).  With a ``DataParallelRanks(model=...)``: backward runs in the
    segments of ``tts.ddp_plan`` and every gradient bucket's all-reduce is started right after its segment, overlapping the
    remaining segments; Adam follows the last collective.  ``out`` holds the iteration's loss terms (device scalars)."""

    def __init__(self, kind, model, opt, batch=None, gaw=None, ddp=None, graph=False, defer_wgrad=False):
        self.kind, self.model, self.opt, self.gaw, self.ddp = kind, model, opt, gaw, ddp
        # weight gradients of equal-shaped layers batched into one launch per backward segment (ops.DeferredWgrad)
        self.defer = ops.DeferredWgrad() if defer_wgrad else None
        self.static = [b.clone() for b in batch] if (graph and batch is not None) else None
        self.batch = self.static if self.static is not None else batch
        self.out = self.att = None
        self._seeds = {}              # (entries, value, device) -> constant gradient vector that seeds the backward
        self.loss_log = None          # a list: receives the loss terms of every EAGERLY executed iteration (warm-up; not replays)
        seg = ddp is not None and ddp.arena is not None
        if graph and ddp is not None and not seg:
            # A replayed backward writes the gradient tensors of the CAPTURE; the packed exchange reads the Python-level p.grad,
            # which after a capture points at its own flat buckets (or at nothing): the replay would exchange stale data.
            raise ValueError("TrainStep(graph=True) needs an arena DataParallelRanks(model=...); a packed DataParallelRanks(params) "
                             "only works with graph=False")
        self.cuts = Cuts(ddp.cut_names if seg else [])
        self._segs = None
        nseg = len(self.cuts.groups) + 1
        phases = [("graph", self._forward_seg0)]
        if seg:
            for i in range(nseg):
                if i > 0:
                    phases.append(("graph", lambda i=i: self._segs[i]()))
                last = i == nseg - 1
                phases.append(("eager", (lambda i=i: (ddp.start_bucket(i), ddp.finish())) if last else (lambda i=i: ddp.start_bucket(i))))
        elif ddp is not None:
            phases.append(("eager", ddp.all_reduce_grads))
        phases.append(("graph", self._adam))
        self.stepper = PhasedStep(phases, graph=graph, drain=ddp.finish if ddp is not None else None)

    def _forward_seg0(self):
        self.opt.zero_grad(set_to_none=True)
        self.cuts.reset()
        if self.defer is not None:
            self.defer.begin_step()
        scale = self.ddp.grad_scale if (self.ddp is not None and self.ddp.arena is not None) else 1.0
        # The loss terms stay the vectors the loss kernels wrote -- (l1, bd) and (att) -- and the backward is seeded with CONSTANT gradient
        # vectors for them (made once, outside any capture): summing selected scalars put ~7 launches of a few microseconds each
        # (select backward: zeros + scatter, adds, the seed's fill) in a row between the forward's last kernel and the backward's first.
        def seedvec(n, like):
            key = (n, float(scale), like.device)
            g = self._seeds.get(key)
            if g is None:
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("TrainStep: run one eager iteration before capturing (the backward's seed vectors are made then)")
                g = self._seeds[key] = torch.full((n,), float(scale), dtype=torch.float32, device=like.device)
            return g

        def seed(v):
            return (v, seedvec(v.numel(), v))
        if self.kind == "text2mel":
            mel, text, spk = self.batch
            with _cuts_installed(self.model, self.cuts):
                pred, att = self.model(shift_right(mel), text, spk)
            lv, av = ops.spec_losses_vec(pred, mel, seedvec(2, pred)), ops.guided_att_loss_vec(att, self.gaw)
            lvd, avd = lv.detach(), av.detach()
            self.out, self.att = (lvd[0], lvd[1], avd[0]), att.detach()
            if "dec_in" in self.cuts.rec:
                segs = backward_segments(self.cuts, [seed(lv)], {"dec_in": [seed(av)]}, self.ddp, self.defer)
            else:
                segs = backward_segments(self.cuts, [seed(lv), seed(av)], None, self.ddp, self.defer)
        else:
            mel, lin = self.batch
            with _cuts_installed(self.model, self.cuts):
                pred = self.model(mel)
            lv = ops.spec_losses_vec(pred, lin, seedvec(2, pred))          # (the seed is known: the loss's forward and backward share one pass)
            lvd = lv.detach()
            self.out = (lvd[0], lvd[1])
            segs = backward_segments(self.cuts, [seed(lv)], None, self.ddp, self.defer)
        self._segs = segs
        if self.loss_log is not None and not torch.cuda.is_current_stream_capturing():
            self.loss_log.append(torch.stack([o.reshape(()) for o in self.out]).clone())
        segs[0]()

    def _adam(self):
        self.opt.step()

    def prepare(self):
        """Warm up and capture (``graph=True``).  One-shot: a prepared step ignores further calls; to capture again -- other static
        buffers, a graph invalidated by a freed pool -- call ``release()`` first."""
        self.stepper.prepare()
        if self.defer is not None:
            self.defer.finish_uploads()
        return self

    def release(self):
        """Drop the captured hipGraphs (nothing may be replaying them) and free the batched weight gradients' frozen job tables for the
        next ``prepare()``.  The step runs eagerly until then."""
        torch.cuda.synchronize()
        self.stepper.plan = None
        self.stepper._graphs = []
        if self.defer is not None:
            self.defer.release_capture()
        return self

    def __call__(self, *batch):
        """Run one iteration; with a batch argument, on that batch (copied into the static buffers of a captured step)."""
        if batch:
            if self.static is not None:
                for dst, src in zip(self.static, batch):
                    dst.copy_(src, non_blocking=True)
            else:
                self.batch = list(batch)
        self.stepper.run()
        return self.out


def text2mel_step(model, optimizer, mel_gt, text_id, spk_emb, gaw, ddp=None):
    """One non-adversarial Text2Mel iteration (train/ordinary.py:221-238), eagerly.  Returns the loss terms."""
    st = TrainStep("text2mel", model, optimizer, [mel_gt, text_id, spk_emb], gaw, ddp, graph=False)
    l1, bd, la = st()
    return l1, bd, la, st.att


def ssrn_step(model, optimizer, mel_gt, lin_gt, ddp=None):
    """One non-adversarial SSRN iteration (train/ordinary.py:240-254), eagerly."""
    return TrainStep("ssrn", model, optimizer, [mel_gt, lin_gt], None, ddp, graph=False)()


# --------------------------------------------------------------------------------------------- WGAN-GP
class AdversarialGraphStep:
    """The generator iteration and the critic iteration of the reference's WGAN-GP trainer
    (train/adversarial_wasserstein_gp.py:261-322), each captured ONCE as hipGraph(s) over static input buffers and
    replayed per iteration -- the eager form is host-bound (hundreds of small critic kernels, three critic forwards and a
    double backward per D iteration, Python autograd glue), not GPU-bound.

    Differences to the eager reference that make capture possible, none changing the mathematics:
      * the adaptive weight of the critic term, (l1+bd+att).item()/|disc|.item() (:290, :338), is formed on the device
        from detached tensors instead of through two host round trips;
      * the interpolation coefficients of the gradient penalty (:300) are drawn on the host like the reference's
        ``torch.rand(B)``, from a generator of their own, and copied into a static buffer before the replay;
      * D iterations run the generator forward without recording a tape (the reference records one and discards it:
        only ``pred.detach()`` is used, :311-313).

    Data parallel (``ddp_syn`` = DataParallelRanks(model=generator), ``ddp_disc`` = DataParallelRanks(critic parameters);
    BASELINE config 4, the reference's MULTI_GPU branch :183-196): every rank holds its shard of the global batch.
      * G iteration: forward graph -> all-reduce(mean) of (l1, bd, att, disc) so the adaptive weight is the GLOBAL-batch value
        -> backward in the segments of ``tts.ddp_plan``, each gradient bucket's all-reduce overlapping the next segment
        -> generator Adam.
      * D iteration: one graph up to the critic gradients (packed with the two loss scalars into one flat bucket) -> one
        all-reduce -> critic Adam.  The penalty coefficients are drawn for the GLOBAL batch from the shared generator and
        sliced by rank, so the iteration does not depend on how the batch is sharded.
    ``kind``: "text2mel" (batch = mel, text, spk) or "ssrn" (batch = mel, lin).  Optimizers must be FusedAdam(capturable=True).
    """

    def __init__(self, kind, model, disc, opt_syn, opt_disc, batch, gaw=None, lam=10.0, ddp_syn=None, ddp_disc=None, graph=True,
                 coeff_seed=0, defer_wgrad=False):
        self.kind, self.model, self.disc, self.opt_syn, self.opt_disc = kind, model, disc, opt_syn, opt_disc
        self.defer = ops.DeferredWgrad() if defer_wgrad else None          # generator iterations only (see TrainStep)
        self.static = [b.clone() for b in batch]
        self.gaw, self.lam = gaw, float(lam)
        self.ddp_syn, self.ddp_disc = ddp_syn, ddp_disc
        self.world = ddp_syn.world if ddp_syn is not None else 1
        self.rank = dist.get_rank(ddp_syn.group) if (ddp_syn is not None and dist.is_initialized()) else 0
        self.g_out = self.d_out = None
        dev = self.static[0].device
        B = self.static[0].shape[0]
        self.coeff = torch.zeros((B, 1, 1), device=dev)
        self._coeff_gen = torch.Generator().manual_seed(int(coeff_seed))
        self.scalars = torch.zeros(4, device=dev)                      # (l1, bd, att, disc), averaged over ranks on G iterations
        seg = ddp_syn is not None and ddp_syn.arena is not None
        if graph and ddp_syn is not None and not seg:
            raise ValueError("AdversarialGraphStep(graph=True) needs an arena ddp_syn = DataParallelRanks(model=...) for the generator "
                             "(see TrainStep); the critic's ddp_disc stays a packed DataParallelRanks(params)")
        self.cuts = Cuts(ddp_syn.cut_names if seg else [])
        self._segs = None
        # ---- generator iteration
        if ddp_syn is None:
            g_phases = [("graph", self._g_forward), ("graph", self._g_backward0), ("graph", self.opt_syn.step)]
        else:
            g_phases = [("graph", self._g_forward), ("eager", lambda: ddp_syn.all_reduce_mean_(self.scalars)), ("graph", self._g_backward0)]
            if seg:
                nseg = len(self.cuts.groups) + 1
                for i in range(nseg):
                    if i > 0:
                        g_phases.append(("graph", lambda i=i: self._segs[i]()))
                    last = i == nseg - 1
                    g_phases.append(("eager", (lambda i=i: (ddp_syn.start_bucket(i), ddp_syn.finish())) if last else (lambda i=i: ddp_syn.start_bucket(i))))
            else:
                g_phases.append(("eager", self._g_exchange_packed))
            g_phases.append(("graph", self.opt_syn.step))
        # ---- critic iteration
        if ddp_disc is None:
            d_phases = [("graph", self._d_compute), ("graph", self.opt_disc.step)]
        else:
            d_phases = [("graph", self._d_compute), ("graph", self._d_pack), ("eager", lambda: (ddp_disc.exchange(), ddp_disc.finish())),
                        ("graph", self._d_unpack_step)]
        def drain():
            for d in (ddp_syn, ddp_disc):
                if d is not None:
                    d.finish()
        self.g_stepper = PhasedStep(g_phases, graph=graph, drain=drain)
        self.d_stepper = PhasedStep(d_phases, graph=graph, drain=drain)
        if graph:
            # warm up both kinds alternately (as the training loop runs them), then capture
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(2):
                    self.g_stepper._eager()
                    self.d_stepper._eager()
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            self.g_stepper.warmup = self.d_stepper.warmup = 0
            self.g_stepper.prepare()
            self.d_stepper.prepare()
            if self.defer is not None:
                self.defer.finish_uploads()

    def _forward(self):
        if self.kind == "text2mel":
            mel, text, spk = self.static
            pred, att = self.model(shift_right(mel), text, spk)
            return pred, att, mel
        mel, lin = self.static
        return self.model(mel), None, lin

    # ---- G
    def _g_forward(self):
        self.opt_syn.zero_grad(set_to_none=True)
        self.opt_disc.zero_grad(set_to_none=True)
        self.cuts.reset()
        with _cuts_installed(self.model, self.cuts):
            pred, att, gt = self._forward()
        l1, bd = ops.spec_losses(pred, gt)
        la = ops.guided_att_loss(att, self.gaw) if att is not None else None
        ld = torch.mean(-self.disc(pred))
        self._g_terms = (l1, bd, la, ld)
        torch.stack([l1.detach(), bd.detach(), la.detach() if la is not None else l1.detach() * 0, ld.detach()], out=self.scalars)

    def _g_backward0(self):
        if self.defer is not None:
            self.defer.begin_step()
        l1, bd, la, ld = self._g_terms
        g = self.scalars                                             # global-batch means when data parallel
        base_g = g[0] + g[1] + g[2]
        weight = (base_g / g[3].abs()).detach()
        scale = self.ddp_syn.grad_scale if self.ddp_syn is not None else 1.0
        seed = lambda t: (t, torch.full_like(t, scale))
        top = l1 + bd + weight * ld
        if "dec_in" in self.cuts.rec:
            segs = backward_segments(self.cuts, [seed(top)], {"dec_in": [seed(la)]}, self.ddp_syn, self.defer)
        else:
            segs = backward_segments(self.cuts, [seed(top + la if la is not None else top)], None, self.ddp_syn, self.defer)
        self._segs = segs
        # the first segment holds the critic: a generator iteration needs its INPUT gradient only (the critic's own parameter gradients
        # would be discarded: the reference zeroes them before the next critic iteration, train/adversarial_wasserstein_gp.py:264-265)
        with ops.input_grads_only(self.disc):
            segs[0]()
        total = base_g + weight * g[3]
        self.g_out = (g[0], g[1], g[2], g[3], total)
        self._g_terms = None

    def _g_exchange_packed(self):
        h = self.ddp_syn.pack()
        self.ddp_syn.exchange()
        self.ddp_syn.finish()
        self.ddp_syn.unpack(h)

    # ---- D
    def _d_compute(self):
        self.opt_syn.zero_grad(set_to_none=True)
        self.opt_disc.zero_grad(set_to_none=True)
        with torch.no_grad():
            pred, _, gt = self._forward()
        B = gt.shape[0]
        scale = self.ddp_disc.grad_scale if self.ddp_disc is not None else 1.0
        mid = (self.coeff * gt + (1 - self.coeff) * pred).requires_grad_(True)
        out = self.disc(mid)
        with ops.input_grads_only(self.disc):        # only d out / d mid is asked for: no weight / bias / LayerNorm parameter gradients in this pass
            grads = torch.autograd.grad(outputs=out, inputs=mid, grad_outputs=torch.ones_like(out), retain_graph=True, create_graph=True)[0]
        loss_gp = ops.grad_penalty(grads, self.lam)          # mean(lam * (||grad||_2 - 1)^2), :305-308
        # disc(pred) and disc(gt) as ONE critic call on the concatenated batch: the critic has no cross-sample operation
        # (LayerNorm is per column, dropout per element), so mean(disc(pred) - disc(gt)) is unchanged and the iteration runs
        # a third fewer (small, launch-bound) critic kernels
        both = self.disc(torch.cat((pred, gt), dim=0))
        loss_d = torch.mean(both[:B] - both[B:])
        # The reference calls loss_gp.backward() and loss_D.backward() (:309, :315): the second call ADDS into every critic parameter's .grad,
        # one small element-wise launch per parameter and iteration (~30: most of the at::native launches of the captured cycle).  Same sums,
        # two launches: both gradient lists from torch.autograd.grad, added by one multi-tensor torch._foreach_add_.
        params = [p for p in self.disc.parameters() if p.requires_grad]
        g_gp = torch.autograd.grad(loss_gp, params, grad_outputs=torch.full_like(loss_gp, scale), allow_unused=True)
        g_d = torch.autograd.grad(loss_d, params, grad_outputs=torch.full_like(loss_d, scale), allow_unused=True)
        pairs = [(a, b) for a, b in zip(g_gp, g_d) if a is not None and b is not None]
        if pairs:
            torch._foreach_add_([a for a, _ in pairs], [b for _, b in pairs])
        for p, a, b in zip(params, g_gp, g_d):
            p.grad = a if a is not None else b
        self.d_out = (loss_d.detach(), loss_gp.detach())

    def _d_pack(self):
        ld, gp = self.d_out
        s = self.ddp_disc.grad_scale
        self._d_handle = self.ddp_disc.pack(extra=(ld * s, gp * s))

    def _d_unpack_step(self):
        self.d_out = tuple(self.ddp_disc.unpack(self._d_handle))
        self.opt_disc.step()

    def load(self, batch):
        """Copy a new batch (same shapes) into the static buffers the graphs read; without graphs the batch is simply used
        (shapes may change from iteration to iteration)."""
        if self.g_stepper.plan is None and any(d.shape != s.shape for d, s in zip(self.static, batch)):
            self.static = [b for b in batch]
            B = self.static[0].shape[0]
            if self.coeff.shape[0] != B:
                self.coeff = torch.zeros((B, 1, 1), device=self.coeff.device)
            return
        for dst, src in zip(self.static, batch):
            dst.copy_(src, non_blocking=True)

    def g_step(self):
        """-> (l1, bin_div, att, disc, total) device scalars of this iteration (global-batch means when data parallel)."""
        self.g_stepper.run()
        return self.g_out

    def d_step(self):
        """-> (loss_D, loss_gp) device scalars (Wasserstein estimate = -loss_D)."""
        B = self.coeff.shape[0]
        c = torch.rand(B * self.world, generator=self._coeff_gen)      # one draw per GLOBAL sample, every rank the same stream
        self.coeff.copy_(c[self.rank * B:(self.rank + 1) * B].view(B, 1, 1))
        self.d_stepper.run()
        return self.d_out
