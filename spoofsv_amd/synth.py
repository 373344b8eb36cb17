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
Measured (tools/bench_synth.py): 0.99 -> 0.67 ms per frame at batch 1, 0.98 -> 0.75 at batch 8, no gain at batch 32 --
each step still computes all ``frames`` columns; a per-layer ring buffer (one column per step) is the next step.
"""
import torch

from . import ops


class GraphSynthesizer:
    """``run(text_id, spk_emb) -> (Y, A)`` with Y (B, F, frames), A (B, N, frames) as the reference loop returns them."""

    def __init__(self, model, batch, text_len, frames, device):
        if model.training:
            raise RuntimeError("GraphSynthesizer needs the model in eval mode")
        self.model, self.B, self.N, self.T = model, batch, text_len, frames
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


_CACHE = {}


def free_run(model, text_id, spk_emb, frames):
    """Drop-in for the step-by-step loop: cached GraphSynthesizer per (model, batch, text length, frames)."""
    key = (id(model), text_id.shape[0], text_id.shape[2], frames)
    g = _CACHE.get(key)
    if g is None:
        g = _CACHE[key] = GraphSynthesizer(model, text_id.shape[0], text_id.shape[2], frames, text_id.device)
    return g.run(text_id, spk_emb)
