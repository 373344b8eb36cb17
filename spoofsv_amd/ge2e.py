"""Host-side mirror of the reference's GE2E speaker embedder (``GE2E/speech_embedder_net.py``).

``SpeechEmbedder().forward(x)``: (B, frames, n_mels) -> (B, proj) unit-norm d-vectors
(speech_embedder_net.py:27-33); ``GE2ELoss(device).forward(emb)``: (N, M, D) -> scalar loss
(:43-49 with GE2E/utils.py:16-55).  Forward only, as BASELINE.json's north_star asks; the LSTM
stack, the projection and the loss run in libssv_hip.so.  State-dict keys equal the reference's
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

    @torch.no_grad()
    def forward(self, x):
        _dev(x, "utterance batch")
        x = _c(x)
        Bn, T, F = x.shape
        nmels, H, layers, P = self.dims
        if F != nmels:
            raise RuntimeError("SpeechEmbedder: input has %d mel bins, model expects %d" % (F, nmels))
        lstm = self.LSTM_stack
        arrs = []
        keep = []
        for kind in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
            ptrs = (ctypes.c_void_p * layers)()
            for l in range(layers):
                t = _c(getattr(lstm, "%s_l%d" % (kind, l)))
                keep.append(t)
                ptrs[l] = t.data_ptr()
            arrs.append(ptrs)
        h_last = torch.empty((Bn, H), dtype=torch.float32, device=x.device)
        nb = _lib.query("ssv_lstm_fwd_workspace", Bn, T, F, H, layers)
        ws = _ws(nb, x.device)
        _lib.call("ssv_lstm_fwd", _p(x), arrs[0], arrs[1], arrs[2], arrs[3], _p(h_last), Bn, T, F, H, layers,
                  _p(ws), nb, _stream())
        e = torch.empty((Bn, P), dtype=torch.float32, device=x.device)
        nb2 = _lib.query("ssv_proj_l2norm_fwd_workspace", Bn, P)
        ws2 = _ws(nb2, x.device)
        _lib.call("ssv_proj_l2norm_fwd", _p(h_last), _p(_c(self.projection.weight)), _p(_c(self.projection.bias)),
                  _p(e), Bn, H, P, _p(ws2), nb2, _stream())
        return e


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
