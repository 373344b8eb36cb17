"""Free-running Text2Mel synthesis as a hipGraph replay (reference loop: synthesize.py:103-109, ordinary.py:59-65).

The reference calls ``melSyn`` once per frame with the growing prefix; every call costs ~60 small kernel launches and the
loop is launch-bound (~1 ms per frame for any batch).  The audio encoder and decoder are strictly causal and LayerNorm is
per column, so running them on a FIXED (B, F, frames) buffer whose future columns are still zero gives exactly the same
values in the columns already synthesised.  One fixed-shape step

    Q = audio_encoder(mel_in)  ->  attention column `col` (mask, softmax, arg-max; frame index on the device)
      ->  R = V A, decoder  ->  mel_in[:, :, col + 1] = Y[:, :, col];  col += 1

is captured once per (batch, text length, frames) and replayed ``frames`` times; K, V come from one eager text-encoder
call.  Same kernels per column as ``melSyn.forward`` in eval mode; the only difference from the step-by-step loop is
that very short prefixes (B*T < 128) run on the exact-fp32 GEMM kernels there and on the split-bf16 ones here (1e-5).
Measured (tools/bench_synth.py): 0.92 -> 0.57 ms per frame at batch 1, 0.90 -> 0.65 at batch 8, no gain at batch 32 --
each step still computes all ``frames`` columns.  ``IncrementalSynthesizer`` below computes one column per step instead
(0.19 / 0.25 / 0.30 ms per frame at batch 1 / 8 / 32).
"""
import torch

from . import ops


class GraphSynthesizer:
    """``run(text_id, spk_emb) -> (Y, A)`` with Y (B, F, frames), A (B, N, frames) as the reference loop returns them."""

    def __init__(self, model, batch, text_len, frames, device):
        if model.training:
            raise RuntimeError("GraphSynthesizer needs the model in eval mode")
        self.model, self.B, self.N, self.T = model, batch, text_len, frames
        self.addresses = _addresses(model)
        d, F = model.hidden_dim, model.audio_decoder.conv5.out_channels
        self.dev = device
        self.kv = torch.zeros((batch, 2 * d, text_len), device=device)
        self.spk = None
        self.mel_in = torch.zeros((batch, F, frames), device=device)
        self.A = torch.zeros((batch, text_len, frames), device=device)
        self.pma = torch.zeros((batch,), dtype=torch.int64, device=device)
        self.col = torch.zeros((1,), dtype=torch.int32, device=device)
        self.Y = None
        self.graph = None

    def _step(self):
        m = self.model
        Q = m.audio_encoder(self.mel_in, self.spk)
        ops.attention_step_dev(self.kv, Q, self.pma, self.A, self.col)
        self.Y = m.audio_decoder(ops.attention_apply(self.kv, self.A, Q, self.T))
        ops.synth_advance(self.Y, self.mel_in, self.col)

    def _capture(self):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s), torch.no_grad():
            self._step()                               # warm the allocator outside the capture
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self._step()

    @torch.no_grad()
    def run(self, text_id, spk_emb):
        B, N, T = self.B, self.N, self.T
        if tuple(text_id.shape) != (B, 1, N):
            raise RuntimeError("GraphSynthesizer was built for text ids of shape %s, got %s" % ((B, 1, N), tuple(text_id.shape)))
        kv = self.model.text_encoder.encode(text_id)                       # (B, 2d, N): K | V
        self.kv.copy_(kv)
        spk = spk_emb.to(self.dev).float()
        if self.spk is None:
            self.spk = spk.clone()
        else:
            self.spk.copy_(spk)
        if self.graph is None:
            self._capture()
        self.mel_in.zero_(); self.A.zero_(); self.pma.zero_(); self.col.zero_()
        for _ in range(T):
            self.graph.replay()
        return self.Y.clone(), self.A.clone()


_CACHE = {}            # small FIFO caches: a synthesizer owns frame-sized buffers and a captured graph
_CACHE_MAX = 4


def _remember(cache, key, make):
    g = cache.get(key)
    if g is None:
        while len(cache) >= _CACHE_MAX:
            cache.pop(next(iter(cache)))
        g = cache[key] = make()
    return g


def _addresses(model):
    """Where the model's parameters live.  A captured step holds these addresses: a model that was moved (``.cpu()`` and back,
    ``load_state_dict(assign=True)``) needs a new capture, its old one would read freed memory."""
    return tuple(p.data_ptr() for p in model.parameters())


def _cached(cache, key, model, make):
    g = _remember(cache, key, make)
    if g.model is not model or g.addresses != _addresses(model):      # an id() can be reused after the first model is gone
        cache.pop(key)
        g = _remember(cache, key, make)
    return g


def free_run(model, text_id, spk_emb, frames):
    """Drop-in for the step-by-step loop: cached GraphSynthesizer per (model, batch, text length, frames)."""
    key = (id(model), text_id.shape[0], text_id.shape[2], frames)
    g = _cached(_CACHE, key, model, lambda: GraphSynthesizer(model, text_id.shape[0], text_id.shape[2], frames, text_id.device))
    return g.run(text_id, spk_emb)


# ------------------------------------------------------------------------------------------------ column-incremental synthesis
class IncrementalSynthesizer:
    """The same loop with one NEW column per step instead of the whole prefix (include/ssv_hip.h, "Column-incremental
    synthesis").  The audio encoder and decoder are causal and LayerNorm acts per column, so the values of every layer at
    frames < t are final; step t computes column t only: 1x1 convs and causal k=3 convs as ``ssv_column_matvec`` (the k=3
    ones read frames t-d, t-2d from their (B, Tmax, C) input history), LayerNorm / highway gate on a length-1 sequence with
    the fused kernels of the full path, ``ssv_attention_column`` for the new attention frame.  ~55 launches per step, each a
    few microseconds, captured once per (batch, text length, frames) and replayed; work per step no longer grows with t.

    Values differ from the prefix loop only by fp32 summation order (plain fp32 dot products here, split-bf16 GEMMs there).
    """

    def __init__(self, model, batch, text_len, frames, device):
        if model.training:
            raise RuntimeError("IncrementalSynthesizer needs the model in eval mode")
        import ctypes
        from . import _lib
        self._lib, self._vp = _lib, ctypes.c_void_p
        self.model, self.B, self.N, self.T, self.dev = model, batch, text_len, frames, device
        self.addresses = _addresses(model)
        enc, dec = model.audio_encoder, model.audio_decoder
        self.d, self.F = model.hidden_dim, dec.conv5.out_channels
        B, d, F = batch, self.d, self.F
        z = lambda *s: torch.zeros(s, dtype=torch.float32, device=device)
        self.kv = z(B, 2 * d, text_len)
        self.mel_cur = z(B, F)
        self.Y = z(B, F, frames)
        self.A = z(B, text_len, frames)
        self.pma = torch.zeros((B,), dtype=torch.int64, device=device)
        self.t = torch.zeros((1,), dtype=torch.int32, device=device)
        self.s1, self.s2 = z(B, d), z(B, d)                         # fc1(spk), fc2(spk): models/TTSModel.py:174,179
        self.enc_hw = [enc.hci1.hc1, enc.hci1.hc2, enc.hci1.hc3, enc.hci1.hc4, enc.hci2.hc1, enc.hci2.hc2, enc.hci2.hc3, enc.hci2.hc4,
                       enc.hc1, enc.hc2]
        self.dec_hw = [dec.hci.hc1, dec.hci.hc2, dec.hci.hc3, dec.hci.hc4, dec.hc1, dec.hc2]
        for hc in self.enc_hw + self.dec_hw:
            if not (hc.causal and hc.kernel_size == 3 and hc.dimension == d):
                raise RuntimeError("IncrementalSynthesizer: unexpected highwayConv configuration")
        self.hist = [z(B, frames, d) for _ in self.enc_hw + self.dec_hw]
        self.pre, self.a, self.b = z(B, 2 * d), z(B, d), z(B, d)              # gate pre-activations / two ping-pong columns
        self.pre1 = z(B, max(d, F))                                            # 1x1 conv outputs before their LayerNorm
        self.rq = z(B, 2 * d)
        self.y_cur = z(B, F)
        self._wt = {}
        self.graph = None

    # ---- one-column operators on fixed buffers -------------------------------------------------------------------
    def _p(self, t):
        return self._vp(t.data_ptr())

    def _tap_major(self, conv):
        """(M, k, C) copy of a k = 3 conv weight in a buffer that lives as long as the synthesizer (the captured step holds its
        address); k = 1 weights are read in place.  ``_repack`` rewrites the copies at the start of every ``run``."""
        w = conv.weight
        if w.shape[2] == 1:
            return w
        ent = self._wt.get(id(conv))
        if ent is None:
            ent = self._wt[id(conv)] = (torch.empty((w.shape[0], w.shape[2], w.shape[1]), dtype=torch.float32, device=w.device), conv)
            ent[0].copy_(w.detach().permute(0, 2, 1))
        return ent[0]

    def _repack(self):
        """Bring the tap-major copies up to date with the live weights (one strided copy per k = 3 layer, 16 per run against
        ~50 launches per frame).  The synthesizer is cached per model and the validation pass of the trainers calls it on the
        model that is being trained; FusedAdam updates weights through raw pointers, so no version counter would tell."""
        for buf, conv in self._wt.values():
            buf.copy_(conv.weight.detach().permute(0, 2, 1))

    def _mv(self, conv, cur, out, hist=None, dilation=1, bias_b=None):
        M, C, k = conv.weight.shape
        w = self._tap_major(conv)
        self._lib.call("ssv_column_matvec", self._p(w), self._p(conv.bias), None if bias_b is None else self._p(bias_b), M,
                       self._p(cur), cur.stride(0), None if hist is None else self._p(hist), 0 if hist is None else hist.stride(0),
                       self.T, self._p(self.t), dilation, self._p(out), out.stride(0), self.B, C, M, k, ops._stream())

    def _ln(self, x, ln, y, act):
        C = ln.weight.shape[0]
        self._lib.call("ssv_column_ln_act", self._p(x), x.stride(0), self._p(ln.weight), self._p(ln.bias), self._p(y), y.stride(0),
                       self.B, C, act, ops._stream())

    def _cla(self, conv, ln, cur, out, act=0, bias_b=None):
        self._mv(conv, cur, self.pre1, bias_b=bias_b)
        self._ln(self.pre1, ln, out, act)

    def _highway(self, hc, hist, cur, out):
        d = self.d
        self._mv(hc.conv, cur, self.pre, hist=hist, dilation=hc.dilation)
        self._lib.call("ssv_column_gate", self._p(self.pre), self._p(cur), cur.stride(0), self._p(hc.ln1.weight), self._p(hc.ln1.bias),
                       self._p(hc.ln2.weight), self._p(hc.ln2.bias), self._p(out), out.stride(0), self.B, d, ops._stream())

    def _step(self):
        enc, dec, d = self.model.audio_encoder, self.model.audio_decoder, self.d
        a, b = self.a, self.b
        cond = enc.condition
        self._cla(enc.conv1, enc.ln1, self.mel_cur, a, act=1, bias_b=self.s1 if cond else None)
        self._cla(enc.conv2, enc.ln2, a, b, act=1)
        self._cla(enc.conv3, enc.ln3, b, a, bias_b=self.s2 if cond else None)
        cur, nxt, h = a, b, 0
        for hc in self.enc_hw:
            self._highway(hc, self.hist[h], cur, nxt)
            cur, nxt, h = nxt, cur, h + 1
        self._lib.call("ssv_attention_column", self._p(self.kv), self.kv.stride(0), self._p(cur), self._p(self.pma), self._p(self.A), self.T,
                       self._p(self.t), self._p(self.rq), self.B, d, self.N, ops._stream())
        self._cla(dec.conv1, dec.ln1, self.rq, a)
        cur, nxt = a, b
        for hc in self.dec_hw:
            self._highway(hc, self.hist[h], cur, nxt)
            cur, nxt, h = nxt, cur, h + 1
        self._cla(dec.conv2, dec.ln2, cur, nxt, act=1)
        self._cla(dec.conv3, dec.ln3, nxt, cur, act=1)
        self._cla(dec.conv4, dec.ln4, cur, nxt, act=1)
        self._cla(dec.conv5, dec.ln5, nxt, self.y_cur, act=2)
        self._lib.call("ssv_synth_column_advance", self._p(self.y_cur), self._p(self.Y), self._p(self.mel_cur), self._p(self.t),
                       self.B, self.F, self.T, ops._stream())

    def _capture(self):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s), torch.no_grad():
            self._step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self._step()

    @torch.no_grad()
    def run(self, text_id, spk_emb):
        B, N, T = self.B, self.N, self.T
        if tuple(text_id.shape) != (B, 1, N):
            raise RuntimeError("IncrementalSynthesizer was built for text ids of shape %s, got %s" % ((B, 1, N), tuple(text_id.shape)))
        enc = self.model.audio_encoder
        self.kv.copy_(self.model.text_encoder.encode(text_id))
        if enc.condition:
            spk = spk_emb.to(self.dev).float()
            self.s1.copy_(ops.conv1d(spk, enc.fc1.weight.unsqueeze(-1), enc.fc1.bias).reshape(B, self.d))
            self.s2.copy_(ops.conv1d(spk, enc.fc2.weight.unsqueeze(-1), enc.fc2.bias).reshape(B, self.d))
        if self.graph is None:
            self._capture()
        else:
            self._repack()
        self.mel_cur.zero_(); self.pma.zero_(); self.t.zero_(); self.A.zero_(); self.Y.zero_()
        for h in self.hist:
            h.zero_()
        for _ in range(T):
            self.graph.replay()
        return self.Y.clone(), self.A.clone()


_ICACHE = {}


def free_run_incremental(model, text_id, spk_emb, frames):
    """Drop-in for the step-by-step loop on the column-incremental path (cached per model / batch / text length / frames)."""
    key = (id(model), text_id.shape[0], text_id.shape[2], frames)
    g = _cached(_ICACHE, key, model, lambda: IncrementalSynthesizer(model, text_id.shape[0], text_id.shape[2], frames, text_id.device))
    return g.run(text_id, spk_emb)
