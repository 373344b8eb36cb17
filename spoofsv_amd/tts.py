"""Host-side mirror of the reference's ``models/TTSModel.py`` module surface on the HIP hot path.

Same class names, constructor signatures, sub-module names (hence identical ``state_dict()`` keys and
tensor layouts, so the reference's checkpoints load unchanged) and the same ``forward`` contracts:

  melSyn(vocab_len, condition, spkemb_dim, textemb_dim=128, freq_bins=80, hidden_dim=256)
      .forward(melspec, textid, spkemb, K=None, V=None, A_last=None, pma=None)
          train mode            -> (Y, A)                      (models/TTSModel.py:263-273)
          eval mode, T == 1     -> (Y, A, pma, K, V)           (:276-298)
          eval mode, T  > 1     -> (Y, A, pma)                 (:299-300)
  SSRN(freq_bins, output_bins, ssrn_dim).forward(mel) -> linear spectrogram (:342-362)
  highwayConv(dimension, kernel_size, dilation, causal=False)  (:43-84; also used by the critics)

The ``nn.Conv1d`` / ``nn.LayerNorm`` / ``nn.Linear`` / ``nn.ConvTranspose1d`` children are parameter
containers only (they give the reference's initialisation order and key names); their stock
``forward`` is never called -- all compute goes through ``spoofsv_amd.ops`` into libssv_hip.so.
"""
import torch
import torch.nn as nn

from . import ops


class textEmbedding(nn.Module):
    def __init__(self, vocab_len, out_channels=128):
        super().__init__()
        self.vocab_len = vocab_len
        self.W = nn.Linear(in_features=vocab_len, out_features=out_channels)

    def forward(self, inputs):
        return ops.text_embed(inputs, self.W.weight, self.W.bias)


class highwayConv(nn.Module):
    def __init__(self, dimension, kernel_size, dilation, causal=False):
        super().__init__()
        self.dimension = dimension
        self.causal = causal
        self.kernel_size = kernel_size
        self.dilation = dilation
        self.pad = dilation * (kernel_size - 1) // 2
        self.conv = nn.Conv1d(in_channels=dimension, out_channels=2 * dimension, kernel_size=kernel_size,
                              padding=0 if causal else self.pad, dilation=dilation)
        self.ln1 = nn.LayerNorm(normalized_shape=dimension)
        self.ln2 = nn.LayerNorm(normalized_shape=dimension)

    def forward(self, inputs):
        return ops.highway_conv1d(inputs, self.conv.weight, self.conv.bias, self.ln1.weight, self.ln1.bias,
                                  self.ln2.weight, self.ln2.bias, self.kernel_size, self.dilation, self.causal)


class highwayDilationIncrement(nn.Module):
    def __init__(self, dimension, causal=False):
        super().__init__()
        self.hc1 = highwayConv(dimension=dimension, kernel_size=3, dilation=1, causal=causal)
        self.hc2 = highwayConv(dimension=dimension, kernel_size=3, dilation=3, causal=causal)
        self.hc3 = highwayConv(dimension=dimension, kernel_size=3, dilation=9, causal=causal)
        self.hc4 = highwayConv(dimension=dimension, kernel_size=3, dilation=27, causal=causal)

    def forward(self, inputs):
        return self.hc4(self.hc3(self.hc2(self.hc1(inputs))))


def _cla(x, conv, ln, s=None, act=0):
    return ops.pointwise_conv_ln_act(x, conv.weight, conv.bias, ln.weight, ln.bias, s, act)


def _no_cut(name, x):
    """Segment boundary hook of the forward pass.  The data-parallel step (train.SegmentedBackward) replaces it with a
    function that detaches ``x`` here, so that backward runs segment by segment and each segment's gradient bucket can be
    all-reduced while the next segment is still computing.  Without one the forward is a single tape."""
    return x


class textEncoder(nn.Module):
    def __init__(self, vocab_len, textemb_dim=128, hidden_dim=256):
        super().__init__()
        self.hidden_dim = hidden_dim
        self.textemb_layer = textEmbedding(vocab_len=vocab_len, out_channels=textemb_dim)
        self.conv1 = nn.Conv1d(in_channels=textemb_dim, out_channels=2 * hidden_dim, kernel_size=1)
        self.ln1 = nn.LayerNorm(normalized_shape=2 * hidden_dim)
        self.conv2 = nn.Conv1d(in_channels=2 * hidden_dim, out_channels=2 * hidden_dim, kernel_size=1)
        self.ln2 = nn.LayerNorm(normalized_shape=2 * hidden_dim)
        self.hci1 = highwayDilationIncrement(dimension=2 * hidden_dim)
        self.hci2 = highwayDilationIncrement(dimension=2 * hidden_dim)
        self.hc1 = highwayConv(dimension=2 * hidden_dim, kernel_size=3, dilation=1)
        self.hc2 = highwayConv(dimension=2 * hidden_dim, kernel_size=3, dilation=1)
        self.hc3 = highwayConv(dimension=2 * hidden_dim, kernel_size=1, dilation=1)
        self.hc4 = highwayConv(dimension=2 * hidden_dim, kernel_size=1, dilation=1)

    def encode(self, inputs, cut=_no_cut):
        """The un-split (B, 2*hidden, N) output; K is the first half, V the second (:138-139)."""
        x = self.textemb_layer(inputs)
        x = _cla(x, self.conv1, self.ln1, act=1)      # relu feeds conv2 (:130)
        x = _cla(x, self.conv2, self.ln2)
        x = cut("text_c0", self.hci1.hc1(x))                     # (hci1 spelled out: its first layer closes the LAST gradient bucket)
        x = cut("text_c1", self.hci1.hc4(self.hci1.hc3(self.hci1.hc2(x))))
        x = cut("text_c2", self.hci2(x))
        return self.hc4(self.hc3(self.hc2(self.hc1(x))))

    def forward(self, inputs):
        x = self.encode(inputs)
        return x[:, :self.hidden_dim, :], x[:, self.hidden_dim:, :]


class audioEncoder(nn.Module):
    def __init__(self, freq_bins, hidden_dim=256, condition=False, spkemb_dim=None):
        super().__init__()
        self.condition = condition
        if condition:
            self.fc1 = nn.Linear(in_features=spkemb_dim, out_features=hidden_dim)
            self.fc2 = nn.Linear(in_features=spkemb_dim, out_features=hidden_dim)
        self.conv1 = nn.Conv1d(in_channels=freq_bins, out_channels=hidden_dim, kernel_size=1)
        self.ln1 = nn.LayerNorm(normalized_shape=hidden_dim)
        self.conv2 = nn.Conv1d(in_channels=hidden_dim, out_channels=hidden_dim, kernel_size=1)
        self.ln2 = nn.LayerNorm(normalized_shape=hidden_dim)
        self.conv3 = nn.Conv1d(in_channels=hidden_dim, out_channels=hidden_dim, kernel_size=1)
        self.ln3 = nn.LayerNorm(normalized_shape=hidden_dim)
        self.hci1 = highwayDilationIncrement(dimension=hidden_dim, causal=True)
        self.hci2 = highwayDilationIncrement(dimension=hidden_dim, causal=True)
        self.hc1 = highwayConv(dimension=hidden_dim, kernel_size=3, dilation=3, causal=True)
        self.hc2 = highwayConv(dimension=hidden_dim, kernel_size=3, dilation=3, causal=True)

    def forward(self, inputs, spk=None, cut=_no_cut):
        s = p = None
        if self.condition:
            # nn.Linear on the (B, D, 1) speaker code == a 1x1 convolution over a length-1 sequence
            s = ops.conv1d(spk, self.fc1.weight.unsqueeze(-1), self.fc1.bias)
            p = ops.conv1d(spk, self.fc2.weight.unsqueeze(-1), self.fc2.bias)
        x = _cla(inputs, self.conv1, self.ln1, s, act=1)
        x = _cla(x, self.conv2, self.ln2, act=1)
        x = _cla(x, self.conv3, self.ln3, p)
        # (the two increments spelled out: the data-parallel step cuts the tape in step with the text encoder's cuts, so that
        # every backward segment has an audio part and a text part to run side by side on the two streams)
        h1, h2 = self.hci1, self.hci2
        x = cut("audio_c1", h1.hc2(h1.hc1(x)))
        x = cut("audio_c2", h2.hc2(h2.hc1(h1.hc4(h1.hc3(x)))))
        return self.hc2(self.hc1(h2.hc4(h2.hc3(x))))


class audioDecoder(nn.Module):
    def __init__(self, freq_bins, hidden_dim=256):
        super().__init__()
        self.conv1 = nn.Conv1d(in_channels=2 * hidden_dim, out_channels=hidden_dim, kernel_size=1)
        self.ln1 = nn.LayerNorm(normalized_shape=hidden_dim)
        self.hci = highwayDilationIncrement(dimension=hidden_dim, causal=True)
        self.hc1 = highwayConv(dimension=hidden_dim, kernel_size=3, dilation=1, causal=True)
        self.hc2 = highwayConv(dimension=hidden_dim, kernel_size=3, dilation=1, causal=True)
        self.conv2 = nn.Conv1d(in_channels=hidden_dim, out_channels=hidden_dim, kernel_size=1)
        self.ln2 = nn.LayerNorm(normalized_shape=hidden_dim)
        self.conv3 = nn.Conv1d(in_channels=hidden_dim, out_channels=hidden_dim, kernel_size=1)
        self.ln3 = nn.LayerNorm(normalized_shape=hidden_dim)
        self.conv4 = nn.Conv1d(in_channels=hidden_dim, out_channels=hidden_dim, kernel_size=1)
        self.ln4 = nn.LayerNorm(normalized_shape=hidden_dim)
        self.conv5 = nn.Conv1d(in_channels=hidden_dim, out_channels=freq_bins, kernel_size=1)
        self.ln5 = nn.LayerNorm(normalized_shape=freq_bins)

    def forward(self, inputs):
        x = _cla(inputs, self.conv1, self.ln1)
        x = self.hc2(self.hc1(self.hci(x)))
        x = _cla(x, self.conv2, self.ln2, act=1)
        x = _cla(x, self.conv3, self.ln3, act=1)
        x = _cla(x, self.conv4, self.ln4, act=1)
        return _cla(x, self.conv5, self.ln5, act=2)


class melSyn(nn.Module):
    def __init__(self, vocab_len, condition, spkemb_dim, textemb_dim=128, freq_bins=80, hidden_dim=256):
        super().__init__()
        self.hidden_dim = hidden_dim
        self._cut = _no_cut
        self.text_encoder = textEncoder(vocab_len=vocab_len, textemb_dim=textemb_dim, hidden_dim=hidden_dim)
        self.audio_encoder = audioEncoder(freq_bins=freq_bins, hidden_dim=hidden_dim, condition=condition, spkemb_dim=spkemb_dim)
        self.audio_decoder = audioDecoder(freq_bins=freq_bins, hidden_dim=hidden_dim)

    def forward(self, melspec, textid, spkemb, K=None, V=None, A_last=None, pma=None):
        T = melspec.shape[-1]
        if self.training:
            # The text encoder and the audio encoder are independent until the attention: run the text branch on a
            # second HIP stream so that its (small) kernels fill the CUs the audio branch leaves idle.  autograd replays
            # each branch's backward on the stream of its forward, so the backward pass overlaps the same way, and a
            # hipGraph capture records the fork/join as parallel branches.
            cur = torch.cuda.current_stream()
            side = _side_stream(melspec.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                kv = self.text_encoder.encode(textid, self._cut)
            Q = self.audio_encoder(melspec, spkemb, self._cut)
            cur.wait_stream(side)     # join; kv stays alive until backward, no record_stream (illegal under capture) needed
            RQ, A = ops.attention_train(kv, Q)
            return self.audio_decoder(self._cut("dec_in", RQ)), A

        # ---- synthesis step (models/TTSModel.py:275-300) ---------------------------------------
        d = self.hidden_dim
        with torch.no_grad():
            if T == 1:
                kv = self.text_encoder.encode(textid)
                K, V = kv[:, :d, :], kv[:, d:, :]
            else:
                kv = _as_kv(K, V)
            N = kv.shape[-1]
            Q = self.audio_encoder(melspec, spkemb)
            B = Q.shape[0]
            A = torch.empty((B, N, T), dtype=torch.float32, device=Q.device)
            if T > 1:
                A[:, :, :T - 1] = A_last                      # older columns are kept as they were (:289-290)
            nxt = ops.attention_step(kv, Q, pma, A, T - 1)    # mask + softmax + arg-max of the new column
            Y = self.audio_decoder(ops.attention_apply(kv, A, Q, T))
        if T == 1:
            return Y, A, nxt, K, V
        return Y, A, nxt


_SIDE = {}


def _side_stream(device):
    key = (device.type, device.index)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device)
    return _SIDE[key]


def _as_kv(K, V):
    """K and V as returned by the first step are the two halves of one (B, 2d, N) tensor; recover it
    without a copy when that is the case, otherwise concatenate."""
    d = K.shape[1]
    if (K.data_ptr() + 4 * d * K.shape[2] == V.data_ptr() and K.stride() == V.stride()
            and K.stride(2) == 1 and K.stride(1) == K.shape[2] and K.stride(0) == 2 * d * K.shape[2]):
        return torch.as_strided(K, (K.shape[0], 2 * d, K.shape[2]), K.stride(), K.storage_offset())
    return torch.cat((K, V), dim=1)


class upsampling(nn.Module):
    def __init__(self, ssrn_dim):
        super().__init__()
        self.deconv = nn.ConvTranspose1d(in_channels=ssrn_dim, out_channels=ssrn_dim, kernel_size=2, stride=2)
        self.hc1 = highwayConv(dimension=ssrn_dim, kernel_size=3, dilation=1)
        self.hc2 = highwayConv(dimension=ssrn_dim, kernel_size=3, dilation=3)

    def forward(self, inputs):
        x = ops.deconv1d_k2s2(inputs, self.deconv.weight, self.deconv.bias)
        return self.hc2(self.hc1(x))


class SSRN(nn.Module):
    def __init__(self, freq_bins, output_bins, ssrn_dim):
        super().__init__()
        self._cut = _no_cut
        self.conv1 = nn.Conv1d(in_channels=freq_bins, out_channels=ssrn_dim, kernel_size=1)
        self.ln1 = nn.LayerNorm(normalized_shape=ssrn_dim)
        self.hc1 = highwayConv(dimension=ssrn_dim, kernel_size=3, dilation=1)
        self.hc2 = highwayConv(dimension=ssrn_dim, kernel_size=3, dilation=3)
        self.ups1 = upsampling(ssrn_dim=ssrn_dim)
        self.ups2 = upsampling(ssrn_dim=ssrn_dim)
        self.conv2 = nn.Conv1d(in_channels=ssrn_dim, out_channels=2 * ssrn_dim, kernel_size=1)
        self.ln2 = nn.LayerNorm(normalized_shape=2 * ssrn_dim)
        self.hc3 = highwayConv(dimension=2 * ssrn_dim, kernel_size=3, dilation=1)
        self.hc4 = highwayConv(dimension=2 * ssrn_dim, kernel_size=3, dilation=1)
        self.conv3 = nn.Conv1d(in_channels=2 * ssrn_dim, out_channels=output_bins, kernel_size=1)
        self.ln3 = nn.LayerNorm(normalized_shape=output_bins)
        self.conv4 = nn.Conv1d(in_channels=output_bins, out_channels=output_bins, kernel_size=1)
        self.ln4 = nn.LayerNorm(normalized_shape=output_bins)
        self.conv5 = nn.Conv1d(in_channels=output_bins, out_channels=output_bins, kernel_size=1)
        self.ln5 = nn.LayerNorm(normalized_shape=output_bins)
        self.conv6 = nn.Conv1d(in_channels=output_bins, out_channels=output_bins, kernel_size=1)
        self.ln6 = nn.LayerNorm(normalized_shape=output_bins)

    def forward(self, inputs):
        x = _cla(inputs, self.conv1, self.ln1)
        x = self.hc2(self.hc1(x))
        x = self.ups2(self.ups1(x))
        x = self._cut("ssrn_mid", _cla(x, self.conv2, self.ln2))
        x = self._cut("ssrn_tail", self.hc4(self.hc3(x)))
        x = _cla(x, self.conv3, self.ln3)             # no ReLU between ln3 and conv4 (:355)
        x = _cla(x, self.conv4, self.ln4, act=1)
        x = _cla(x, self.conv5, self.ln5, act=1)
        return _cla(x, self.conv6, self.ln6, act=2)


# ------------------------------------------------------------------------------------------------ data-parallel plan
def _groups(module):
    """Parameter groups of ``module`` in gradient-arena order (gradarena.py): per fused operator the conv weight, then the
    block of small gradients its backward kernel emits as one array -- highwayConv: (ln1.w, ln1.b, ln2.w, ln2.b, conv.b);
    1x1 conv + LayerNorm pairs ``convN`` / ``lnN``: (ln.w, ln.b, conv.b).  Everything else one group per parameter."""
    groups, seen = [], set()

    def take(ps):
        ps = [p for p in ps if p is not None and id(p) not in seen]
        seen.update(id(p) for p in ps)
        if ps:
            groups.append(ps)
    for m in module.modules():
        if isinstance(m, highwayConv):
            take([m.conv.weight])
            take([m.ln1.weight, m.ln1.bias, m.ln2.weight, m.ln2.bias, m.conv.bias])
        else:
            for name, child in m.named_children():
                if name.startswith("conv") and isinstance(child, nn.Conv1d) and isinstance(getattr(m, "ln" + name[4:], None), nn.LayerNorm):
                    ln = getattr(m, "ln" + name[4:])
                    take([child.weight])
                    take([ln.weight, ln.bias, child.bias])
    for p in module.parameters():
        take([p])
    return groups


def ddp_plan(model):
    """``(plan, cuts)`` for ``train.SegmentedBackward``: gradient buckets in the order backward finishes them, and the names of
    the forward cuts (see ``_no_cut``) that end each backward segment.  Bucket i is complete once segment i has run.

    melSyn:  decoder | attention + audio top 4 + text hc1-4 | audio middle 4 + text hci2 | audio head + text hci1.hc2-4 |
             text hci1.hc1 + head.  Every middle segment ends at one cut in the text encoder AND one in the audio encoder, so its
             backward still runs the two branches side by side on the model's two streams.
    SSRN:    513-wide 1x1 tail | the two C=512 highway layers | everything at C=256 (upsampling, head)
    A cut entry is a list of cut names that end the same segment.  Sizing rule: the all-reduce of bucket i runs under segment
    i+1, so the LAST bucket (exposed) is kept small (7.6 MB) and the one before it has a last segment long enough to hide under
    (one C=512 highway layer + the head, ~0.3 ms)."""
    if hasattr(model, "ddp_plan"):
        return model.ddp_plan()
    if isinstance(model, melSyn):
        te, ae = model.text_encoder, model.audio_encoder
        G = lambda *mods: [g for m in mods for g in _groups(m)]
        named = lambda m, *names: [g for g in _groups(m) if all(any(p is q for n in names for q in getattr(m, n).parameters()) for p in g)]
        plan = [("audio_decoder", _groups(model.audio_decoder)),
                ("audio_top+text_top", G(ae.hc2, ae.hc1, ae.hci2.hc4, ae.hci2.hc3, te.hc4, te.hc3, te.hc2, te.hc1)),
                ("audio_mid+text_hci2", G(ae.hci2.hc2, ae.hci2.hc1, ae.hci1.hc4, ae.hci1.hc3, te.hci2)),
                ("audio_head+text_hci1.hc2-4", G(ae.hci1.hc2, ae.hci1.hc1) + named(ae, "conv3", "ln3", "conv2", "ln2", "conv1", "ln1") +
                 (named(ae, "fc1", "fc2") if ae.condition else []) + G(te.hci1.hc4, te.hci1.hc3, te.hci1.hc2)),
                ("text_hci1.hc1+head", G(te.hci1.hc1, te.textemb_layer) + named(te, "conv2", "ln2", "conv1", "ln1"))]
        cuts = [["dec_in"], ["text_c2", "audio_c2"], ["text_c1", "audio_c1"], ["text_c0"]]
    elif isinstance(model, SSRN):
        tail = nn.ModuleDict({k: getattr(model, k) for k in ("conv3", "ln3", "conv4", "ln4", "conv5", "ln5", "conv6", "ln6")})
        head = nn.ModuleDict({k: getattr(model, k) for k in ("conv1", "ln1", "hc1", "hc2", "ups1", "ups2", "conv2", "ln2")})
        plan = [("tail_513", _groups(tail)), ("hc_512", _groups(model.hc3) + _groups(model.hc4)), ("head_256", _groups(head))]
        cuts = [["ssrn_tail"], ["ssrn_mid"]]
    else:
        return [("all", _groups(model))], []
    have = {id(p) for _, gs in plan for g in gs for p in g}
    missing = [n for n, p in model.named_parameters() if id(p) not in have]
    if missing:
        raise RuntimeError("ddp_plan: parameters without a bucket: %s" % ", ".join(missing))
    return plan, cuts
