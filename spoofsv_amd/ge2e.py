"""Host-side mirror of the reference's GE2E speaker embedder (``GE2E/speech_embedder_net.py``).

``SpeechEmbedder().forward(x)``: (B, frames, n_mels) -> (B, proj) unit-norm d-vectors
(speech_embedder_net.py:27-33); ``GE2ELoss(device).forward(emb)``: (N, M, D) -> scalar loss
(:43-49 with GE2E/utils.py:16-55).  The LSTM stack, the projection and the loss run in libssv_hip.so, forward (what
BASELINE.json's north_star asks) and backward (one iteration of GE2E/train_speech_embedder.py:70-86
with torch's own clip_grad_norm_ and SGD on top).  State-dict keys equal the reference's
(``LSTM_stack.weight_ih_l0`` ... ``projection.bias``), so its checkpoints load unchanged.

The reference reads its sizes from a module-global ``hparam`` loaded from config/config.yaml; here they
are constructor arguments whose defaults are that file's values (nmels 40, hidden 768, 3 layers,
proj 256; GE2E/config/config.yaml:16,21-23).
"""
import ctypes

import torch
import torch.nn as nn

from . import _lib
from .ops import _c, _dev, _p, _stream, _ws


def _ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


_FWD_CACHE = {}       # the last inference call's workspace and what it was prepared for (see _EmbedderFn.forward)


class _EmbedderFn(torch.autograd.Function):
    """SpeechEmbedder.forward (speech_embedder_net.py:27-33) with its backward: LSTM stack -> last frame -> Linear -> x/|x|.
    Inputs: train flag (keep every frame for the backward), x, projection weight, projection bias, then per layer
    w_ih, w_hh, b_ih, b_hh."""

    @staticmethod
    def forward(ctx, train, x, pw, pb, *lstm):
        layers = len(lstm) // 4
        Bn, T, F = x.shape
        H, P = lstm[1].shape[1], pw.shape[0]
        w_ih, w_hh = [_c(lstm[4 * l]) for l in range(layers)], [_c(lstm[4 * l + 1]) for l in range(layers)]
        b_ih, b_hh = [_c(lstm[4 * l + 2]) for l in range(layers)], [_c(lstm[4 * l + 3]) for l in range(layers)]
        pw, pb = _c(pw), _c(pb)
        h_last = torch.empty((Bn, H), dtype=torch.float32, device=x.device)
        args = (_ptr_array(w_ih), _ptr_array(w_hh), _ptr_array(b_ih), _ptr_array(b_hh))
        if train:
            saved = _ws(_lib.query("ssv_lstm_saved_bytes", Bn, T, F, H, layers), x.device)
            nb = _lib.query("ssv_lstm_train_fwd_workspace", Bn, T, F, H, layers)
            ws = _ws(nb, x.device)
            _lib.call("ssv_lstm_train_fwd", _p(x), *args, _p(h_last), _p(saved), Bn, T, F, H, layers, _p(ws), nb, _stream())
        else:
            # d-vector extraction runs batch after batch on fixed weights: the workspace (with the split weight planes and their scale) is kept
            # between calls and re-used as long as shape, arithmetic mode, stream and every weight's (address, version) are the same
            nb = _lib.query("ssv_lstm_fwd_workspace", Bn, T, F, H, layers)
            key = (Bn, T, F, H, layers, _lib.precision(), x.device, _stream().value, tuple((t.data_ptr(), t._version) for t in lstm))
            hit = _FWD_CACHE.get("key") == key
            ws = _FWD_CACHE["ws"] if hit else _ws(nb, x.device)
            _lib.call("ssv_lstm_fwd_cached", _p(x), *args, _p(h_last), Bn, T, F, H, layers, _p(ws), nb, 1 if hit else 0, _stream())
            _FWD_CACHE["key"], _FWD_CACHE["ws"] = key, ws
        e = torch.empty((Bn, P), dtype=torch.float32, device=x.device)
        norms = torch.empty((Bn,), dtype=torch.float32, device=x.device) if train else None
        nb2 = _lib.query("ssv_proj_l2norm_fwd_workspace", Bn, P)
        ws2 = _ws(nb2, x.device)
        _lib.call("ssv_proj_l2norm_fwd", _p(h_last), _p(pw), _p(pb), _p(e), _p(norms), Bn, H, P, _p(ws2), nb2, _stream())
        if train:
            ctx.save_for_backward(saved, h_last, e, norms, pw, *w_ih, *w_hh)
            ctx.dims = (Bn, T, F, H, P, layers)
        return e

    @staticmethod
    def backward(ctx, de):
        Bn, T, F, H, P, layers = ctx.dims
        saved, h_last, e, norms, pw = ctx.saved_tensors[:5]
        w_ih = list(ctx.saved_tensors[5:5 + layers])
        w_hh = list(ctx.saved_tensors[5 + layers:5 + 2 * layers])
        dev = e.device
        de = _c(de)
        dh = torch.empty((Bn, H), dtype=torch.float32, device=dev)
        dpw = torch.empty_like(pw)
        dpb = torch.empty((P,), dtype=torch.float32, device=dev)
        nb = _lib.query("ssv_proj_l2norm_bwd_workspace", Bn, P)
        ws = _ws(nb, dev)
        _lib.call("ssv_proj_l2norm_bwd", _p(de), _p(e), _p(norms), _p(h_last), _p(pw), _p(dh), _p(dpw), _p(dpb), Bn, H, P,
                  _p(ws), nb, _stream())
        dw_ih = [torch.empty_like(w) for w in w_ih]
        dw_hh = [torch.empty_like(w) for w in w_hh]
        db_ih = [torch.empty((4 * H,), dtype=torch.float32, device=dev) for _ in range(layers)]
        db_hh = [torch.empty((4 * H,), dtype=torch.float32, device=dev) for _ in range(layers)]
        nb2 = _lib.query("ssv_lstm_bwd_workspace", Bn, T, F, H, layers)
        ws2 = _ws(nb2, dev)
        _lib.call("ssv_lstm_bwd", _p(dh), _p(saved), _ptr_array(w_ih), _ptr_array(w_hh), _ptr_array(dw_ih), _ptr_array(dw_hh),
                  _ptr_array(db_ih), _ptr_array(db_hh), Bn, T, F, H, layers, _p(ws2), nb2, _stream())
        grads = []
        for l in range(layers):
            grads += [dw_ih[l], dw_hh[l], db_ih[l], db_hh[l]]
        return (None, None, dpw, dpb) + tuple(grads)


class SpeechEmbedder(nn.Module):
    def __init__(self, nmels=40, hidden=768, num_layer=3, proj=256):
        super().__init__()
        self.LSTM_stack = nn.LSTM(nmels, hidden, num_layers=num_layer, batch_first=True)
        for name, param in self.LSTM_stack.named_parameters():      # speech_embedder_net.py:20-24
            if "bias" in name:
                nn.init.constant_(param, 0.0)
            elif "weight" in name:
                nn.init.xavier_normal_(param)
        self.projection = nn.Linear(hidden, proj)
        self.dims = (nmels, hidden, num_layer, proj)

    def forward(self, x):
        _dev(x, "utterance batch")
        x = _c(x)
        nmels, H, layers, P = self.dims
        if x.shape[2] != nmels:
            raise RuntimeError("SpeechEmbedder: input has %d mel bins, model expects %d" % (x.shape[2], nmels))
        lstm = self.LSTM_stack
        params = []
        for l in range(layers):
            params += [getattr(lstm, "%s_l%d" % (kind, l)) for kind in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
        if self.training and torch.is_grad_enabled():
            # keeps every frame for backpropagation through time (ssv_lstm_saved_bytes: 5.8 GB at 880 x 120 frames)
            return _EmbedderFn.apply(True, x, self.projection.weight, self.projection.bias, *params)
        with torch.no_grad():       # d-vector extraction (GE2E/dvector_create.py:100): nothing is kept
            return _EmbedderFn.apply(False, x, self.projection.weight, self.projection.bias, *params)


class _GE2ELossFn(torch.autograd.Function):
    """GE2ELoss.forward (speech_embedder_net.py:43-49, utils.py:16-55) and its gradient w.r.t. embeddings, w, b."""

    @staticmethod
    def forward(ctx, emb, w, b):
        e = _c(emb)
        N, M, D = e.shape
        w1, b1 = _c(w.reshape(1)), _c(b.reshape(1))
        loss = torch.empty((1,), dtype=torch.float32, device=e.device)
        per = torch.empty((N, M), dtype=torch.float32, device=e.device)
        nb = _lib.query("ssv_ge2e_loss_fwd_workspace", N, M, D)
        ws = _ws(nb, e.device)
        _lib.call("ssv_ge2e_loss_fwd", _p(e), _p(w1), _p(b1), _p(loss), _p(per), N, M, D, _p(ws), nb, _stream())
        ctx.save_for_backward(e, w1, b1)
        ctx.mark_non_differentiable(per)
        return loss[0], per

    @staticmethod
    def backward(ctx, dloss, _dper):
        e, w1, b1 = ctx.saved_tensors
        N, M, D = e.shape
        demb = torch.empty_like(e)
        dw = torch.empty((1,), dtype=torch.float32, device=e.device)
        db = torch.empty((1,), dtype=torch.float32, device=e.device)
        g = _c(dloss.reshape(1).float())
        nb = _lib.query("ssv_ge2e_loss_bwd_workspace", N, M, D)
        ws = _ws(nb, e.device)
        _lib.call("ssv_ge2e_loss_bwd", _p(e), _p(w1), _p(b1), _p(g), _p(demb), _p(dw), _p(db), N, M, D, _p(ws), nb, _stream())
        return demb, dw.reshape(()), db.reshape(())


class GE2ELoss(nn.Module):
    def __init__(self, device):
        super().__init__()
        self.w = nn.Parameter(torch.tensor(10.0).to(device), requires_grad=True)
        self.b = nn.Parameter(torch.tensor(-5.0).to(device), requires_grad=True)
        self.device = device

    def forward(self, embeddings, return_per_embedding=False):
        _dev(embeddings, "embeddings")
        loss, per = _GE2ELossFn.apply(embeddings.float(), self.w, self.b)
        return (loss, per) if return_per_embedding else loss


def train_iteration(embedder_net, ge2e_loss, optimizer, mel_db_batch, N, M):
    """One iteration of GE2E/train_speech_embedder.py:70-86 (the batch permutation of :67-73 is undone at :78 before the
    loss and does not enter the arithmetic, so it is omitted): forward, loss, backward on the HIP path, then torch's own
    ``clip_grad_norm_`` (3.0 on the embedder, 1.0 on the loss parameters) and the optimizer step, as the reference does."""
    optimizer.zero_grad()
    x = mel_db_batch.reshape(N * M, mel_db_batch.size(-2), mel_db_batch.size(-1))
    embeddings = embedder_net(x).reshape(N, M, -1)
    loss = ge2e_loss(embeddings)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(embedder_net.parameters(), 3.0)
    torch.nn.utils.clip_grad_norm_(ge2e_loss.parameters(), 1.0)
    optimizer.step()
    return loss.detach()


# --------------------------------------------------------------------------------------------- multi-GPU
# The embedder shards by utterance; the loss needs every embedding of a speaker and every centroid, so there is exactly one
# exchange: an all-gather of the (N_local*M, P) embeddings -- 113 KB per rank at 880 utterances over 8 ranks.  Every rank then
# evaluates the same full loss, so the gradient of a rank's own embeddings is its slice of d(loss)/d(embeddings) (no second
# collective for the activations); the embedder's weight gradients are partial sums over the local utterances and are summed
# over ranks, the loss parameters' gradients are already complete on every rank.
class _GatherEmbeddings(torch.autograd.Function):
    @staticmethod
    def forward(ctx, e, group):
        import torch.distributed as dist
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        e = e.contiguous()
        parts = [torch.empty_like(e) for _ in range(world)]
        dist.all_gather(parts, e, group=group)
        ctx.rank, ctx.n = rank, e.shape[0]
        return torch.cat(parts, 0)

    @staticmethod
    def backward(ctx, g):
        return g[ctx.rank * ctx.n:(ctx.rank + 1) * ctx.n].contiguous(), None


def gather_embeddings(e, group=None):
    """All ranks' embeddings in rank order (differentiable); the identity without an initialised process group."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return e
    return _GatherEmbeddings.apply(e, group)


def sharded_train_iteration(embedder_net, ge2e_loss, optimizer, mel_db_local, N_local, M, group=None):
    """``train_iteration`` with the N speakers of the batch split over the ranks of ``group`` (rank r holds speakers
    [r*N_local, (r+1)*N_local), each with its M utterances): local embedder forward, one all-gather of the embeddings, the
    full loss on every rank, backward, one SUM all-reduce of the embedder's gradients, then the reference's clipping and
    optimizer step -- every rank ends with the weights the single-process iteration on the whole batch produces."""
    import torch.distributed as dist
    optimizer.zero_grad()
    x = mel_db_local.reshape(N_local * M, mel_db_local.size(-2), mel_db_local.size(-1))
    e_all = gather_embeddings(embedder_net(x), group)
    loss = ge2e_loss(e_all.reshape(-1, M, e_all.shape[-1]))
    loss.backward()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        grads = [p.grad for p in embedder_net.parameters() if p.grad is not None]
        flat = torch.cat([g.reshape(-1) for g in grads])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        off = 0
        for g in grads:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()
    torch.nn.utils.clip_grad_norm_(embedder_net.parameters(), 3.0)
    torch.nn.utils.clip_grad_norm_(ge2e_loss.parameters(), 1.0)
    optimizer.step()
    return loss.detach()
