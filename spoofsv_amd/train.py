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


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (no weight decay / amsgrad, as the reference uses it), all parameters
    updated by ONE multi-tensor kernel launch (ssv_adam_multi).  State keys match torch's Adam
    (``step``, ``exp_avg``, ``exp_avg_sq``) so optimizer state dicts interchange.

    With ``capturable=True`` the step count lives on the device and advances inside the kernel, so a
    captured hipGraph of the whole training step replays correctly.
    """

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, capturable=False, resident=True):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.capturable = capturable
        self._table = None
        self._host_table = None
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
        if self._host_table is None:
            cap = sum((p.numel() + _CHUNK - 1) // _CHUNK for g in self.param_groups for p in g["params"])
            self._host_table = torch.empty((cap, 5), dtype=torch.int64).pin_memory()
            self._table = torch.empty((cap, 5), dtype=torch.int64, device=plist[0].device)
        self._host_table[:len(rows)].copy_(torch.from_numpy(arr))
        self._table.copy_(self._host_table, non_blocking=True)
        self._nchunks = len(rows)
        self._key = key

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
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
    return torch.cat((torch.zeros_like(mel_gt[:, :, :1]), mel_gt[:, :, :-1]), dim=-1)


# --------------------------------------------------------------------------------------------- steps
def text2mel_losses(pred, att, mel_gt, gaw):
    """(l1, bin_div, att) of train/ordinary.py:230-236 on the HIP loss kernels."""
    l1, bd = ops.spec_losses(pred, mel_gt)
    return l1, bd, ops.guided_att_loss(att, gaw)


def text2mel_step(model, optimizer, mel_gt, text_id, spk_emb, gaw, ddp=None):
    """One non-adversarial Text2Mel iteration (train/ordinary.py:221-238).  Returns the loss terms."""
    optimizer.zero_grad(set_to_none=True)
    pred, att = model(shift_right(mel_gt), text_id, spk_emb)
    l1, bd, la = text2mel_losses(pred, att, mel_gt, gaw)
    loss = l1 + bd + la
    loss.backward()
    if ddp is not None:
        ddp.all_reduce_grads()
    optimizer.step()
    return l1, bd, la, att


def ssrn_step(model, optimizer, mel_gt, lin_gt, ddp=None):
    """One non-adversarial SSRN iteration (train/ordinary.py:240-254)."""
    optimizer.zero_grad(set_to_none=True)
    pred = model(mel_gt)
    l1, bd = ops.spec_losses(pred, lin_gt)
    (l1 + bd).backward()
    if ddp is not None:
        ddp.all_reduce_grads()
    optimizer.step()
    return l1, bd


# --------------------------------------------------------------------------------------------- DDP
class DataParallelRanks:
    """One process per GPU; utterances are sharded by rank and gradients are averaged with ONE flat
    all-reduce per bucket over RCCL/xGMI (backend "nccl" on ROCm; "gloo" in the CPU tests).

    The reference's nn.DataParallel (train/ordinary.py:165-173) re-broadcasts all parameters every
    iteration and reduces gradients to GPU 0; here replicas stay in sync because every rank applies the
    same averaged gradient, so the only traffic is the gradient all-reduce (96.3 MB for Text2Mel).
    Gradients are packed into a few large flat buckets because xGMI is point-to-point: large messages
    keep each link busy, per-tensor all-reduces (214 tensors) would be launch/latency bound.
    """

    def __init__(self, params, bucket_mb=64, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.bucket_elems = max(1, int(bucket_mb * 1024 * 1024 // 4))
        self._flat = None

    def broadcast_parameters(self, src=0):
        """Make every replica start from rank `src`'s weights (what DataParallel's replicate does)."""
        if self.world == 1:
            return
        with torch.no_grad():
            for p in self.params:
                dist.broadcast(p.data, src, group=self.group)
        _resident.invalidate(self.params)      # p.data writes do not bump the version the resident planes are checked against

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
    def all_reduce_grads_begin(self, grads=None):
        """Start averaging gradients over ranks and return a handle for ``all_reduce_grads_end``.  ``grads`` defaults
        to each parameter's ``.grad`` (pass the tensors a captured hipGraph writes when replaying one).  Gradients are
        packed into persistent flat buckets (one ``cat`` kernel each) and every bucket is all-reduced asynchronously:
        packing bucket i+1 overlaps the collective of bucket i, and whatever the caller enqueues before calling
        ``..._end`` (e.g. the other model's step) overlaps the collectives on RCCL's own stream."""
        if self.world == 1:
            return None
        if grads is None:
            grads = [p.grad for p in self.params]
        buckets = self._buckets(grads)
        if self._flat is None or len(self._flat) != len(buckets):
            self._flat = [torch.empty(sum(g.numel() for _, g in b), dtype=b[0][1].dtype, device=b[0][1].device) for b in buckets]
        works = []
        for flat, bucket in zip(self._flat, buckets):
            torch.cat([g.reshape(-1) for _, g in bucket], out=flat)
            works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        return buckets, works

    @torch.no_grad()
    def all_reduce_grads_end(self, handle):
        """Wait for the collectives, scale by 1/world and re-point every ``p.grad`` at its averaged slice of the
        bucket -- no copy back, and the optimizer's pointer table stays stable from step to step."""
        if handle is None:
            return
        buckets, works = handle
        for flat, bucket, w in zip(self._flat, buckets, works):
            w.wait()
            flat.mul_(1.0 / self.world)
            off = 0
            for p, g in bucket:
                n = g.numel()
                p.grad = flat[off:off + n].view_as(g)
                off += n

    def all_reduce_grads(self, grads=None):
        self.all_reduce_grads_end(self.all_reduce_grads_begin(grads))

    @torch.no_grad()
    def all_reduce_mean(self, *scalars):
        """Average loss scalars over ranks (global-batch semantics for the adaptive critic weight,
        train/adversarial_wasserstein_gp.py:290)."""
        if self.world == 1:
            return scalars
        v = torch.stack([s.detach().reshape(()) for s in scalars])
        dist.all_reduce(v, op=dist.ReduceOp.SUM, group=self.group)
        v /= self.world
        return tuple(v[i] for i in range(len(scalars)))


# --------------------------------------------------------------------------------------------- WGAN-GP
class AdversarialGraphStep:
    """The generator iteration and the critic iteration of the reference's WGAN-GP trainer
    (train/adversarial_wasserstein_gp.py:261-322), each captured ONCE as a hipGraph over static input buffers and
    replayed per iteration -- the eager form is host-bound (hundreds of small critic kernels, three critic forwards and a
    double backward per D iteration, Python autograd glue), not GPU-bound.

    Differences to the eager reference that make capture possible, none changing the mathematics:
      * the adaptive weight of the critic term, (l1+bd+att).item()/|disc|.item() (:290, :338), is formed on the device
        from detached tensors instead of through two host round trips;
      * the interpolation coefficients of the gradient penalty (:300) come from the device RNG instead of the CPU RNG;
      * D iterations run the generator forward without recording a tape (the reference records one and discards it:
        only ``pred.detach()`` is used, :311-313).
    ``kind``: "text2mel" (batch = mel, text, spk) or "ssrn" (batch = mel, lin).  Optimizers must be FusedAdam(capturable=True).
    """

    def __init__(self, kind, model, disc, opt_syn, opt_disc, batch, gaw=None, lam=10.0):
        self.kind, self.model, self.disc, self.opt_syn, self.opt_disc = kind, model, disc, opt_syn, opt_disc
        self.static = [b.clone() for b in batch]
        self.gaw, self.lam = gaw, float(lam)
        self.g_out = self.d_out = None
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                self._g_iter()
                self._d_iter()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.g_graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_graph):
            self.g_out = self._g_iter()
        self.d_graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.d_graph):
            self.d_out = self._d_iter()

    def _forward(self):
        if self.kind == "text2mel":
            mel, text, spk = self.static
            pred, att = self.model(shift_right(mel), text, spk)
            return pred, att, mel
        mel, lin = self.static
        return self.model(mel), None, lin

    def _g_iter(self):
        self.opt_syn.zero_grad(set_to_none=True)
        self.opt_disc.zero_grad(set_to_none=True)
        pred, att, gt = self._forward()
        l1, bd = ops.spec_losses(pred, gt)
        base = l1 + bd
        la = None
        if att is not None:
            la = ops.guided_att_loss(att, self.gaw)
            base = base + la
        ld = torch.mean(-self.disc(pred))
        loss = base + (base.detach() / ld.detach().abs()) * ld
        loss.backward()
        self.opt_syn.step()
        return tuple(t.detach() for t in (l1, bd, la if la is not None else l1 * 0, ld, loss))

    def _d_iter(self):
        self.opt_syn.zero_grad(set_to_none=True)
        self.opt_disc.zero_grad(set_to_none=True)
        with torch.no_grad():
            pred, _, gt = self._forward()
        B, C, T = gt.shape
        coeff = torch.rand(B, 1, 1, device=gt.device)
        mid = (coeff * gt + (1 - coeff) * pred).requires_grad_(True)
        out = self.disc(mid)
        grads = torch.autograd.grad(outputs=out, inputs=mid, grad_outputs=torch.ones_like(out), retain_graph=True, create_graph=True)[0]
        loss_gp = torch.mean(self.lam * (torch.norm(grads, p=2, dim=(1, 2)) - 1) ** 2)
        loss_gp.backward()
        # disc(pred) and disc(gt) as ONE critic call on the concatenated batch: the critic has no cross-sample operation
        # (LayerNorm is per column, dropout per element), so mean(disc(pred) - disc(gt)) is unchanged and the iteration runs
        # a third fewer (small, launch-bound) critic kernels
        both = self.disc(torch.cat((pred, gt), dim=0))
        loss_d = torch.mean(both[:B] - both[B:])
        loss_d.backward()
        self.opt_disc.step()
        return loss_d.detach(), loss_gp.detach()

    def load(self, batch):
        """Copy a new batch (same shapes) into the static buffers the graphs read."""
        for dst, src in zip(self.static, batch):
            dst.copy_(src, non_blocking=True)

    def g_step(self):
        """-> (l1, bin_div, att, disc, total) device scalars of this iteration."""
        self.g_graph.replay()
        return self.g_out

    def d_step(self):
        """-> (loss_D, loss_gp) device scalars (Wasserstein estimate = -loss_D)."""
        self.d_graph.replay()
        return self.d_out
