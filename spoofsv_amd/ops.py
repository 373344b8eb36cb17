"""Host-side operators: thin ``torch.autograd.Function`` wrappers over the C ABI of libssv_hip.so.

PyTorch is plumbing here (device memory, the current HIP stream, the autograd tape); every FLOP of
the hot path runs in the hand-written gfx950 kernels.  Tensors must live on a ROCm device -- there
is deliberately no CPU or stock-op fallback (``RuntimeError`` otherwise).

Layout convention: activations are (B, C, T) float32 with contiguous rows (stride(2) == 1,
stride(1) == T); the batch stride is free, so channel slices of a wider tensor are passed without
copies.  Each Function cites the reference code its forward replaces.
"""
import contextlib
import ctypes

import torch

from . import _lib, gradarena, resident

_F32 = torch.float32


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _dev(t, what="tensor"):
    if not t.is_cuda:
        raise RuntimeError("spoofsv_amd: %s is on %s; the HIP hot path needs a ROCm device tensor "
                           "(no CPU fallback exists)" % (what, t.device))
    return t


def _act3(t, what="activation"):
    """Return (tensor, batch_stride) with contiguous rows; copies only when the rows are not."""
    _dev(t, what)
    if t.dtype != _F32:
        t = t.float()
    if t.dim() != 3:
        raise RuntimeError("spoofsv_amd: %s must be (B, C, T), got %s" % (what, tuple(t.shape)))
    B, C, L = t.shape
    if L == 1:          # strides of a length-1 axis are arbitrary
        if t.stride(1) != 1 and C > 1:
            t = t.contiguous()
        return t, (t.stride(0) if B > 1 else C)
    if t.stride(2) != 1 or t.stride(1) != L or (B > 1 and t.stride(0) < C * L):
        t = t.contiguous()
    return t, (t.stride(0) if B > 1 else C * L)


def _c(t):
    """Dense float32 parameter/tensor."""
    _dev(t, "parameter")
    return t if (t.dtype == _F32 and t.is_contiguous()) else t.float().contiguous()


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


# ------------------------------------------------------------------------------------------- split-fp16 operand scales
# include/ssv_hip.h, "Operand scales (mode 2)": a kernel that reads an fp32 tensor as an MFMA operand needs a list of partial
# maxima of |x| (n entries per batch item).  The LayerNorm / gate kernels write one for their output as a by-product; it
# travels to the consumer as an attribute of the output tensor (with the tensor's version, so an in-place edit voids it).  A
# tensor without one (data, a view, the output of another kernel) gets an ``ssv_absmax`` launch.  Other arithmetic modes: None.
_AMAX_PIECES = 8


def _f16():
    return _lib.precision() == 2


_AMAX_ROWS = {}


def _amax_out(B, L, device):
    """Buffer for the scale list a LayerNorm / gate kernel writes for its (B, C, L) output (``ssv_amax_rows(L)`` per item)."""
    n = _AMAX_ROWS.get(L)
    if n is None:
        n = _AMAX_ROWS[L] = int(_lib.lib().ssv_amax_rows(L))
    return torch.empty((B, n), dtype=_F32, device=device)


def _tag(y, amax):
    y._ssv_amax = (amax, y._version)
    return y


def amax_of(x):
    """Scale list of ``x`` (B, C, L): the producer's, or computed here."""
    h = getattr(x, "_ssv_amax", None)
    if h is not None and h[1] == x._version and h[0].shape[0] == x.shape[0] and h[0].device == x.device:
        return h[0]
    xd, xbs = _act3(x.detach(), "operand")
    B, C, L = xd.shape
    out = torch.empty((B, _AMAX_PIECES), dtype=_F32, device=xd.device)
    _lib.call("ssv_absmax", _p(xd), xbs, B, C * L, _p(out), _AMAX_PIECES, _stream())
    return out               # NOT remembered on x: buffers that kernels update through raw pointers keep their version


def shift_right(mel):
    """Teacher-forcing input [0 | mel[:, :, :-1]] (train/ordinary.py:226) with its operand scale list as a by-product: ONE launch instead of
    torch.cat's copy kernel plus an ``ssv_absmax`` launch at the audio encoder's first convolution.  No gradient (the input is data)."""
    x, xbs = _act3(mel.detach(), "shift_right input")
    B, C, T = x.shape
    y = torch.empty((B, C, T), dtype=_F32, device=x.device)
    npb = min(C, 40)               # row ranges per item = workgroups per item (80 mel bins: two rows each; a launch of B * 40 small workgroups)
    amax = torch.empty((B, npb), dtype=_F32, device=x.device)
    _lib.call("ssv_shift_right_amax", _p(x), xbs, _p(y), B, C, T, _p(amax), npb, _stream())
    return _tag(y, amax)


def _an(a):
    """(pointer, entries per item) of a scale list or (None, 0)."""
    return (None, 0) if a is None else (_p(a), a.shape[1])


def _needs_grad(ctx):
    """True when autograd will call backward for this node (inside Function.forward grad mode is always
    off, so the tape's own bookkeeping is the reliable signal)."""
    return any(ctx.needs_input_grad)


# ------------------------------------------------------------------------------------------- deferred weight gradients
class DeferredWgrad:
    """Weight gradients of equal-shaped layers, collected during backward and computed by ONE launch per shape at ``flush``
    (include/ssv_hip.h, "Weight gradients of several equal-shaped conv layers in ONE launch"): nothing downstream in backward
    reads a dW, and a single layer must cut its reduction over (batch, time) into up to B slabs just to fill the chip.  While
    an instance is active (``with deferred:``), the fused operators run only the LayerNorm / gate backward and the data
    gradient, and queue a job; the step that owns the instance calls ``flush`` before anything reads the gradients (end of a
    backward segment: before the bucket's all-reduce / Adam).  Results differ from the immediate path only in the fp32
    summation order over the batch (other slab boundaries).

    Job tables travel host -> device through pinned buffers that belong to the instance: slot i serves the i-th launch of a
    step (``begin_step`` rewinds).  They are allocated during the eager warm-up iterations; a hipGraph capture re-uses them (a
    capture cannot allocate pinned memory, and the copy node reads the pinned table at every replay, so it must stay intact)."""

    def __init__(self):
        self.jobs = {}
        self.slots = []                # [pinned uint8 table, device table (eager launches), event after the last copy out of the pinned table,
                                       #  a capture's upload pending, device table of the CAPTURED launch (written once), that table in use]
        self.cursor = 0
        self.pending = set()           # addresses of the parameters whose gradient is queued and not yet flushed
        self._side = None              # stream for the table uploads of a capture (see flush)

    def __enter__(self):
        global _DEFER
        self._prev, _DEFER = _DEFER, self
        return self

    def __exit__(self, *exc):
        global _DEFER
        _DEFER = self._prev
        return False

    def begin_step(self):
        self.cursor = 0

    def finish_uploads(self):
        """Wait for the table uploads a capture started on the side stream (call after the capture, before the first replay)."""
        if self._side is not None:
            self._side.synchronize()
        for slot in self.slots:
            slot[3] = False

    def release_capture(self):
        """The captured step that read this instance's frozen job tables has been dropped (its hipGraphs destroyed): the tables may be
        rewritten by the next capture.  Call only when no replay of the old graphs can run any more (``TrainStep.release``)."""
        if self._side is not None:
            self._side.synchronize()
        for slot in self.slots:
            slot[3] = False
            slot[5] = False
        self.jobs = {}                 # (a capture that failed half way may have left its queue behind)
        self.pending.clear()

    @staticmethod
    def accepts(B, Cin, Cout, L, k, nblk, params=()):
        """``params``: the parameters whose gradients the job would deliver late.  A deferred gradient is handed to autograd
        BEFORE it is computed, which is only sound while autograd adopts the tensor as ``p.grad`` as it is: not when it would
        add it to an existing ``.grad`` or record the addition (create_graph) -- then the immediate path is taken."""
        if torch.is_grad_enabled() or any(p is not None and p.grad is not None for p in params):
            return False
        ok = nblk <= 768 and bool(_lib.lib().ssv_conv1d_bwd_weight_multi_ok(B, Cin, Cout, L, k))
        if _DEFER is not None:
            # A parameter used twice in one backward: p.grad is still None at its second use (AccumulateGrad runs after all users),
            # and autograd would ADD the first, not yet computed, deferred gradient to the second.  Nothing in the models here
            # shares a parameter; refuse loudly instead of computing a wrong sum.
            ptrs = [p.data_ptr() for p in params if p is not None]
            if any(q in _DEFER.pending for q in ptrs):
                raise RuntimeError("DeferredWgrad: a parameter is used by two operators of one backward pass; deferred (batched) "
                                   "weight gradients cannot be summed by autograd -- build the step with defer_wgrad=False")
            if ok:
                _DEFER.pending.update(ptrs)
        return ok

    def add(self, dy, dy_bs, x, x_bs, dw, part, pg, k, dilation, causal, n2, nblk, dy_amax=None, x_amax=None):
        B, Cin, L = x.shape
        key = (B, Cin, dy.shape[1], L, k, dy_bs, x_bs, n2, nblk, x.device)
        sh = (ctypes.c_int * 3)()
        _lib.call("ssv_conv_shifts", k, dilation, int(causal), sh)
        # dw and pg are what the caller hands to autograd, which adopts a returned gradient as ``p.grad`` only while nobody else
        # holds the tensor OBJECT or a view of it (otherwise it clones -- here: the not yet computed values).  The queue keeps
        # the address and the STORAGE alive, not the tensors.
        self.jobs.setdefault(key, []).append((dy, x, part, dw.data_ptr(), pg.data_ptr(), (dw.untyped_storage(), pg.untyped_storage()), tuple(sh),
                                              dy_amax, x_amax))

    def _slot(self, nbytes, dev):
        capturing = torch.cuda.is_current_stream_capturing()
        if self.cursor == len(self.slots):
            if capturing:
                raise RuntimeError("DeferredWgrad: run one eager iteration before capturing (job tables are pinned buffers, "
                                   "which cannot be allocated during a hipGraph capture)")
            cap = max(4096, nbytes)
            self.slots.append([torch.empty(cap, dtype=torch.uint8).pin_memory(), torch.empty(cap, dtype=torch.uint8, device=dev), None, False,
                               torch.empty(cap, dtype=torch.uint8, device=dev), False])
        slot = self.slots[self.cursor]
        if slot[0].numel() < nbytes or slot[1].device != dev:
            raise RuntimeError("DeferredWgrad: the sequence of weight-gradient launches changed between iterations")
        if capturing:
            # (HIP refuses an event synchronize from a capturing thread: "operation not permitted on an event last recorded in a
            # capturing stream", also for events of other streams.)  The pinned table is free unless an EARLIER capture's upload from
            # it is still in flight on the side stream: captures must be separated by finish_uploads().
            if slot[3]:
                raise RuntimeError("DeferredWgrad: call finish_uploads() after a capture before capturing again")
            slot[3] = True
        elif slot[2] is not None:
            slot[2].synchronize()          # the previous copy out of this pinned table has executed (launch stream, or a capture's side stream)
        self.cursor += 1
        return slot

    def flush(self):
        """One weight-gradient launch (+ one reduction launch) per queued shape, on the current stream."""
        for key, jobs in self.jobs.items():
            B, Cin, Cout, L, k, dy_bs, x_bs, n2, nblk, dev = key
            n = len(jobs)
            table = (_lib.WgradJob * n)()
            f16 = _f16()
            for t, (dy, x, part, dw_ptr, pg_ptr, _, sh, dy_amax, x_amax) in zip(table, jobs):
                t.dy, t.x, t.dw, t.part, t.pgrads = dy.data_ptr(), x.data_ptr(), dw_ptr, part.data_ptr(), pg_ptr
                t.shift[0], t.shift[1], t.shift[2] = sh
                if f16:
                    if dy_amax is None or x_amax is None:
                        raise RuntimeError("DeferredWgrad: a job was queued without operand scales in the split-fp16 mode")
                    t.dy_amax, t.x_amax, t.dy_namax, t.x_namax = dy_amax.data_ptr(), x_amax.data_ptr(), dy_amax.numel(), x_amax.numel()
            raw = bytes(table)
            slot = self._slot(len(raw), dev)
            slot[0][:len(raw)].copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))
            if torch.cuda.is_current_stream_capturing():
                # The table of a captured launch is CONSTANT (the addresses of the capture's own tensors): it is uploaded once, now,
                # on a stream outside the capture, instead of by a copy node that every replay would run in front of the launch
                # (11 such nodes of ~4 us per training step).  ``finish_uploads`` (after the capture) waits for it.
                # It goes into the slot's SECOND device table, which eager iterations never write: an eager iteration through the same
                # DeferredWgrad after the capture -- a logged or debug step -- rewinds the cursor and rewrites the slots' first tables,
                # and the replays must keep reading the captured addresses.  (Allocated by the eager warm-up, outside any capture: a
                # block from the capture's own pool may be the recycled memory of an earlier activation OF THE SAME CAPTURE, whose
                # producer would overwrite the table at every replay -- the upload below is not part of the captured order.)
                if slot[5]:
                    raise RuntimeError("DeferredWgrad: this instance already serves a captured step (its frozen job tables are in use); "
                                       "build a new step object for another capture")
                slot[5] = True
                dtable = slot[4]
                if self._side is None:
                    self._side = torch.cuda.Stream(device=dev)
                with torch.cuda.stream(self._side):
                    dtable.copy_(slot[0], non_blocking=True)
                    slot[2] = torch.cuda.Event()          # the pinned table may be refilled once this copy has executed
                    slot[2].record()
            else:
                dtable = slot[1]
                dtable.copy_(slot[0], non_blocking=True)
                slot[2] = torch.cuda.Event()
                slot[2].record()
            nb = _lib.query("ssv_conv1d_bwd_weight_multi_workspace", n, B, Cin, Cout, L, k)
            ws = _ws(nb, dev)
            max_shift = max(abs(v) for job in jobs for v in job[6])
            _lib.call("ssv_conv1d_bwd_weight_multi", _p(dtable), n, dy_bs, x_bs, B, Cin, Cout, L, k, max_shift, n2, nblk, _p(ws), nb, _stream())
        self.jobs = {}
        self.pending.clear()


_DEFER = None


# ------------------------------------------------------------------------------------------- highway
class HighwayConvFn(torch.autograd.Function):
    """highwayConv.forward, models/TTSModel.py:63-84 (conv -> 2x LayerNorm over channels -> gate)."""

    @staticmethod
    def forward(ctx, x, w, bias, g1, b1, g2, b2, k, dilation, causal, x_amax=None, y_amax=None):
        x, xbs = _act3(x, "highwayConv input")
        B, C, L = x.shape
        w, bias, g1, b1, g2, b2 = map(_c, (w, bias, g1, b1, g2, b2))
        if tuple(w.shape) != (2 * C, C, k):
            raise RuntimeError("highwayConv: weight %s does not match input channels %d" % (tuple(w.shape), C))
        train = _needs_grad(ctx)
        y = torch.empty((B, C, L), dtype=_F32, device=x.device)
        h = torch.empty((B, 2 * C, L), dtype=_F32, device=x.device)
        stats = torch.empty((B, 4, L), dtype=_F32, device=x.device) if train else None
        nb = _lib.query("ssv_highway_conv1d_fwd_workspace", B, C, L, k)
        ws = _ws(nb, x.device)
        _lib.call("ssv_highway_conv1d_fwd", _p(x), xbs, *_an(x_amax), _p(w), resident.lookup(w), _p(bias), _p(g1), _p(b1), _p(g2), _p(b2),
                  _p(h), _p(stats), _p(y), C * L, _p(y_amax), B, C, L, k, dilation, int(causal), _p(ws), nb, _stream())
        if train:
            ctx.save_for_backward(x, w, g1, b1, g2, b2, h, stats)
            ctx.x_amax = x_amax
            ctx.cfg = (k, dilation, int(causal))
            ctx.bias_ref = bias           # only its address is used (gradient-arena lookup); not needed by the kernels
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, g1, b1, g2, b2, h, stats = ctx.saved_tensors
        k, dilation, causal = ctx.cfg
        x, xbs = _act3(x)
        dy, dybs = _act3(dy, "highwayConv grad")
        B, C, L = x.shape
        dx = torch.empty((B, C, L), dtype=_F32, device=x.device)
        # parameter gradients go straight into the data-parallel gradient arena when the model has one (gradarena.py)
        dw = gradarena.grad_like(w)
        bias = ctx.bias_ref
        pg = gradarena.grad_block((g1, b1, g2, b2, bias), 6, C, x.device) if bias is not None else torch.empty((6, C), dtype=_F32, device=x.device)
        nblk = _lib.query("ssv_ln_partial_rows", B, L)
        if _DEFER is not None and _DEFER.accepts(B, C, 2 * C, L, k, nblk, (w, g1, b1, g2, b2, bias)):
            # LayerNorm / gate backward + data gradient now; the weight gradient joins the other layers of this shape at flush
            dh = torch.empty((B, 2 * C, L), dtype=_F32, device=x.device)
            part = torch.empty((nblk, 6 * C), dtype=_F32, device=x.device)
            f16 = _f16()
            dh_amax = _amax_out(B, L, x.device) if f16 else None
            x_amax = (ctx.x_amax if ctx.x_amax is not None else amax_of(x)) if f16 else None
            rows = _lib.query("ssv_ln_bwd_partial_rows", 1, B, C, L, int(f16))          # the rows the gate backward writes (one per tile of its kernel)
            nb = _lib.query("ssv_highway_conv1d_bwd_data_workspace", B, C, L, k)
            ws = _ws(nb, x.device)
            _lib.call("ssv_highway_conv1d_bwd_data", _p(dy), dybs, _p(x), xbs, _p(w), resident.lookup(w), _p(g1), _p(b1), _p(g2), _p(b2),
                      _p(h), _p(stats), _p(dx), C * L, _p(dh), _p(dh_amax), _p(part), B, C, L, k, dilation, causal, _p(ws), nb, _stream())
            _DEFER.add(dh, 2 * C * L, x, xbs, dw, part, pg, k, dilation, causal, 6 * C, rows, dh_amax, x_amax)
            return (dx, dw, pg[4:6].reshape(2 * C), pg[0], pg[1], pg[2], pg[3]) + (None,) * 5
        nb = _lib.query("ssv_highway_conv1d_bwd_workspace", B, C, L, k)
        ws = _ws(nb, x.device)
        _lib.call("ssv_highway_conv1d_bwd", _p(dy), dybs, _p(x), xbs, *_an(ctx.x_amax), _p(w), resident.lookup(w), _p(g1), _p(b1), _p(g2), _p(b2),
                  _p(h), _p(stats), _p(dx), C * L, _p(dw), _p(pg), B, C, L, k, dilation, causal,
                  _p(ws), nb, _stream())
        return (dx, dw, pg[4:6].reshape(2 * C), pg[0], pg[1], pg[2], pg[3]) + (None,) * 5


# ------------------------------------------------------------------------------------------- conv
def _conv_fwd(x, xbs, w, bias, bias_b, y, ybs, k, dilation, causal, x_amax=None):
    B, Cin, L = x.shape
    nb = _lib.query("ssv_conv1d_fwd_workspace", Cin, w.shape[0], k)
    ws = _ws(nb, x.device)
    _lib.call("ssv_conv1d_fwd", _p(x), xbs, *_an(x_amax), _p(w), resident.lookup(w), _p(bias), _p(bias_b), _p(y), ybs, None, B, Cin, w.shape[0], L,
              k, dilation, int(causal), _p(ws), nb, _stream())


def _conv_bwd_data(dy, dybs, w, Cin, L, k=1, dilation=1, causal=0, dy_amax=None):
    B, Cout = dy.shape[0], w.shape[0]
    dx = torch.empty((B, Cin, L), dtype=_F32, device=dy.device)
    nb = _lib.query("ssv_conv1d_bwd_data_workspace", Cin, Cout, k)
    ws = _ws(nb, dy.device)
    _lib.call("ssv_conv1d_bwd_data", _p(dy), dybs, *_an(dy_amax), _p(w), resident.lookup(w), None, _p(dx), Cin * L, B, Cin, Cout, L, k, dilation,
              int(causal), _p(ws), nb, _stream())
    return dx


def _conv_bwd_weight(dy, dybs, x, xbs, wshape, k=1, dilation=1, causal=0, out=None, dy_amax=None, x_amax=None):
    B, Cin, L = x.shape
    Cout = wshape[0]
    dw = out if out is not None else torch.empty(wshape, dtype=_F32, device=x.device)
    nb = _lib.query("ssv_conv1d_bwd_weight_workspace", B, Cin, Cout, L, k)
    ws = _ws(nb, x.device)
    _lib.call("ssv_conv1d_bwd_weight", _p(dy), dybs, *_an(dy_amax), _p(x), xbs, *_an(x_amax), _p(dw), B, Cin, Cout, L, k, dilation, int(causal),
              _p(ws), nb, _stream())
    return dw


def _sum_over_batch(rows, B, n, out=None):
    """rows: (B, n) dense -> (n,) summed in batch order."""
    if out is None:
        out = torch.empty((n,), dtype=_F32, device=rows.device)
    _lib.call("ssv_sum_slabs", _p(rows), _p(out), n, B, n, _stream())
    return out


class PointwiseConvLnActFn(torch.autograd.Function):
    """`ln(conv1x1(x) [+ s])` followed by relu / sigmoid / nothing.

    Replaces e.g. models/TTSModel.py:128-131 (text encoder), :173-180 (audio encoder, with the
    broadcast speaker term ``s`` = fc(spk), a (B, C, 1) tensor), :218-231, :343-361.  ``act``: 0 none,
    1 relu (the reference applies F.relu to this output when feeding the next conv), 2 sigmoid.
    """

    @staticmethod
    def forward(ctx, x, w, bias, gamma, beta, s, act, x_amax=None, y_amax=None):
        x, xbs = _act3(x, "conv input")
        B, Cin, L = x.shape
        w, bias, gamma, beta = map(_c, (w, bias, gamma, beta))
        Cout = w.shape[0]
        if w.shape[1] != Cin or w.shape[2] != 1:
            raise RuntimeError("pointwise conv: weight %s does not match input channels %d" % (tuple(w.shape), Cin))
        sb = None
        if s is not None:
            sb = _c(s.reshape(B, Cout))
        train = _needs_grad(ctx)
        pre = torch.empty((B, Cout, L), dtype=_F32, device=x.device)
        y = torch.empty((B, Cout, L), dtype=_F32, device=x.device)
        stats = torch.empty((B, 2, L), dtype=_F32, device=x.device) if train else None
        nb = _lib.query("ssv_pointwise_conv_ln_act_fwd_workspace", Cin, Cout)
        ws = _ws(nb, x.device)
        _lib.call("ssv_pointwise_conv_ln_act_fwd", _p(x), xbs, *_an(x_amax), _p(w), resident.lookup(w), _p(bias), _p(sb), _p(gamma), _p(beta),
                  _p(pre), _p(stats), _p(y), Cout * L, _p(y_amax), B, Cin, Cout, L, act, _p(ws), nb, _stream())
        if train:
            ctx.save_for_backward(x, w, gamma, beta, pre, stats)
            ctx.x_amax = x_amax
            ctx.act = act
            ctx.bias_ref = bias
            ctx.has_s = s is not None
            ctx.need_dx = ctx.needs_input_grad[0]
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, gamma, beta, pre, stats = ctx.saved_tensors
        x, xbs = _act3(x)
        dy, dybs = _act3(dy, "grad")
        B, Cin, L = x.shape
        Cout = w.shape[0]
        pg = gradarena.grad_block((gamma, beta, ctx.bias_ref), 3, Cout, x.device)
        dx = torch.empty((B, Cin, L), dtype=_F32, device=x.device) if ctx.need_dx else None
        dw = gradarena.grad_like(w)
        ds = torch.empty((B, Cout, 1), dtype=_F32, device=x.device) if ctx.has_s else None
        nblk = _lib.query("ssv_ln_partial_rows", B, L)
        if _DEFER is not None and _DEFER.accepts(B, Cin, Cout, L, 1, nblk, (w, gamma, beta, ctx.bias_ref)):
            dpre = torch.empty((B, Cout, L), dtype=_F32, device=x.device)
            part = torch.empty((nblk, 3 * Cout), dtype=_F32, device=x.device)
            f16 = _f16()
            dpre_amax = _amax_out(B, L, x.device) if f16 else None
            x_amax = (ctx.x_amax if ctx.x_amax is not None else amax_of(x)) if f16 else None
            rows = _lib.query("ssv_ln_bwd_partial_rows", 0, B, Cout, L, int(f16))
            nb = _lib.query("ssv_pointwise_conv_ln_act_bwd_data_workspace", B, Cin, Cout, L)
            ws = _ws(nb, x.device)
            _lib.call("ssv_pointwise_conv_ln_act_bwd_data", _p(dy), dybs, _p(w), resident.lookup(w), _p(gamma), _p(beta), _p(pre), _p(stats),
                      _p(dx), Cin * L, _p(ds), _p(dpre), _p(dpre_amax), _p(part), B, Cin, Cout, L, ctx.act, _p(ws), nb, _stream())
            _DEFER.add(dpre, Cout * L, x, xbs, dw, part, pg, 1, 1, 0, 3 * Cout, rows, dpre_amax, x_amax)
            return dx, dw, pg[2], pg[0], pg[1], ds, None, None, None
        nb = _lib.query("ssv_pointwise_conv_ln_act_bwd_workspace", B, Cin, Cout, L)
        ws = _ws(nb, x.device)
        _lib.call("ssv_pointwise_conv_ln_act_bwd", _p(dy), dybs, _p(x), xbs, *_an(ctx.x_amax), _p(w), resident.lookup(w), _p(gamma), _p(beta), _p(pre), _p(stats),
                  _p(dx), Cin * L, _p(dw), _p(pg), _p(ds), B, Cin, Cout, L, ctx.act, _p(ws), nb, _stream())
        return dx, dw, pg[2], pg[0], pg[1], ds, None, None, None


class Conv1dFn(torch.autograd.Function):
    """Plain Conv1d with bias (kernel 1 or 3).  With L == 1 it is nn.Linear on (B, D, 1) speaker codes,
    models/TTSModel.py:174,179 (`fc(spk.permute(0,2,1)).permute(0,2,1)`)."""

    @staticmethod
    def forward(ctx, x, w, bias, k, dilation, causal, x_amax=None):
        x, xbs = _act3(x, "conv input")
        B, Cin, L = x.shape
        w = _c(w)
        bias = _c(bias) if bias is not None else None
        Cout = w.shape[0]
        y = torch.empty((B, Cout, L), dtype=_F32, device=x.device)
        _conv_fwd(x, xbs, w, bias, None, y, Cout * L, k, dilation, causal, x_amax)
        if _needs_grad(ctx):
            ctx.save_for_backward(x, w)
            ctx.x_amax = x_amax
            ctx.cfg = (k, dilation, int(causal), ctx.needs_input_grad[0], bias is not None)
            ctx.bias_ref = bias
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        k, dilation, causal, need_dx, has_bias = ctx.cfg
        x, xbs = _act3(x)
        dy, dybs = _act3(dy, "grad")
        B, Cin, L = x.shape
        Cout = w.shape[0]
        dy_amax = amax_of(dy) if (_f16() and _bf3_shape(dy)) else None       # (the L = 1 speaker-code convs take the fp32 kernel)
        dx = _conv_bwd_data(dy, dybs, w, Cin, L, k, dilation, causal, dy_amax) if need_dx else None
        dw = _conv_bwd_weight(dy, dybs, x, xbs, tuple(w.shape), k, dilation, causal, out=gradarena.view(w), dy_amax=dy_amax, x_amax=ctx.x_amax)
        db = None
        if has_bias:
            db = gradarena.view(ctx.bias_ref)
            if db is None:
                db = torch.empty((Cout,), dtype=_F32, device=x.device)
            _lib.call("ssv_bias_grad", _p(dy), dybs, _p(db), B, Cout, L, _stream())
        return dx, dw, db, None, None, None, None


# ------------------------------------------------------------------------------------------- conv, any-order differentiable
# The WGAN-GP critic needs the gradient of a gradient (train/adversarial_wasserstein_gp.py:300-308).  A convolution is
# bilinear in (x, w), so its three kernels -- forward, data gradient, weight gradient -- are closed under differentiation:
# each is an autograd Function whose backward is written with the other two, which makes the op differentiable to any
# order on the HIP kernels alone.  The bias gradient is a plain torch sum (already differentiable).
# Split-fp16 operand scales (``*_amax``: the tensor's scale list or None): every tensor that enters these Functions as a GEMM
# operand has its list computed ONCE (``_dd_amax``: the producer's tag or one ssv_absmax launch) and handed to every product that
# reads it -- x serves the forward and the weight gradient, dy both gradients -- instead of one fallback launch per product.
def _tiny_conv(cin, cout):
    """Mirrors SSV_MIN_SPLIT_CHANNELS of csrc/api.hip: a convolution with fewer than 32 input or output channels runs all three of its
    products on the exact-fp32 kernels, which read no scale lists."""
    return min(int(cin), int(cout)) < 32


def _dd_amax(t, cin=None, cout=None):
    if t is None or not (_f16() and _bf3_shape(t)) or (cin is not None and _tiny_conv(cin, cout)):
        return None
    return amax_of(t)


_INPUT_ONLY = None        # None, or the set of parameter addresses whose gradients the current backward does not want


def _skip_param_grads(*params):
    """True when every given parameter belongs to the module ``input_grads_only`` was entered for."""
    if _INPUT_ONLY is None:
        return False
    return all(p is not None and p.data_ptr() in _INPUT_ONLY for p in params)


@contextlib.contextmanager
def input_grads_only(module):
    """Within the block the backward of the twice-differentiable operators produces the INPUT gradient only -- for the operators whose
    parameters belong to ``module`` (the critic).  The gradient penalty's first pass, ``torch.autograd.grad(outputs=D(x_mid),
    inputs=x_mid, create_graph=True)`` (train/adversarial_wasserstein_gp.py:303-304), asks for nothing else -- but a Python Function
    cannot see which of its gradients the engine will use (``ctx.needs_input_grad`` is fixed at forward time), so without this every
    convolution also ran its weight-gradient GEMM, slab reduction and bias row sums, and every LayerNorm / gate its
    parameter-gradient reductions, for results nobody reads.  The generator's backward runs inside the same block on generator
    iterations (the critic is the head of its tape): an operator over any OTHER parameter -- a generator layer that adopts ``conv1d_dd``
    / ``channel_ln_dd`` one day -- keeps all of its gradients."""
    global _INPUT_ONLY
    prev = _INPUT_ONLY
    _INPUT_ONLY = frozenset(p.data_ptr() for p in module.parameters())
    try:
        yield
    finally:
        _INPUT_ONLY = prev


class ConvFwdDD(torch.autograd.Function):
    """y = conv1d(x, w) + bias, kernel 1 or 3, "same" or causal zero padding (the bias is added by the kernel's epilogue)."""

    @staticmethod
    def forward(ctx, x, w, bias, k, dilation, causal, x_amax=None):
        x, xbs = _act3(x, "conv input")
        w = _c(w)
        bias = _c(bias) if bias is not None else None
        B, Cin, L = x.shape
        y = torch.empty((B, w.shape[0], L), dtype=_F32, device=x.device)
        _conv_fwd(x, xbs, w, bias, None, y, w.shape[0] * L, k, dilation, causal, x_amax)
        ctx.save_for_backward(x, w)
        ctx.x_amax = x_amax
        ctx.cfg = (k, dilation, causal, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        k, dilation, causal, has_bias = ctx.cfg
        dya = _dd_amax(dy, x.shape[1], w.shape[0])
        dx = ConvBwdDataDD.apply(dy, w, x.shape[1], k, dilation, causal, dya) if ctx.needs_input_grad[0] else None
        if _skip_param_grads(w):              # (input_grads_only: the gradient penalty's first pass, the critic under a generator iteration)
            return dx, None, None, None, None, None, None
        dw = ConvBwdWeightDD.apply(dy, x, k, dilation, causal, dya, ctx.x_amax) if ctx.needs_input_grad[1] else None
        db = BiasGradFn.apply(dy) if (has_bias and ctx.needs_input_grad[2]) else None
        return dx, dw, db, None, None, None, None


class BiasGradFn(torch.autograd.Function):
    """db[c] = sum_{b,t} dy[b,c,t] on the HIP row-sum kernels (fixed summation order); linear in dy."""

    @staticmethod
    def forward(ctx, dy):
        dy, dybs = _act3(dy, "grad")
        B, C, L = dy.shape
        rows = torch.empty((B, C), dtype=_F32, device=dy.device)
        _lib.call("ssv_rowsum", _p(dy), dybs, _p(rows), B, C, L, _stream())
        ctx.dims = (B, C, L)
        return _sum_over_batch(rows, B, C)

    @staticmethod
    def backward(ctx, g):
        B, C, L = ctx.dims
        return g.view(1, C, 1).expand(B, C, L)


class ConvBwdDataDD(torch.autograd.Function):
    """dx = conv1d_transpose(dy, w): linear in dy and in w."""

    @staticmethod
    def forward(ctx, dy, w, Cin, k, dilation, causal, dy_amax=None):
        dy, dybs = _act3(dy, "grad")
        w = _c(w)
        dx = _conv_bwd_data(dy, dybs, w, Cin, dy.shape[2], k, dilation, causal, dy_amax)
        ctx.save_for_backward(dy, w)
        ctx.dy_amax = dy_amax
        ctx.cfg = (k, dilation, causal)
        return dx

    @staticmethod
    def backward(ctx, ddx):
        dy, w = ctx.saved_tensors
        k, dilation, causal = ctx.cfg
        xa = _dd_amax(ddx, w.shape[1], w.shape[0])
        g_dy = ConvFwdDD.apply(ddx, w, None, k, dilation, causal, xa) if ctx.needs_input_grad[0] else None
        g_w = ConvBwdWeightDD.apply(dy, ddx, k, dilation, causal, ctx.dy_amax, xa) if ctx.needs_input_grad[1] else None
        return g_dy, g_w, None, None, None, None, None


class ConvBwdWeightDD(torch.autograd.Function):
    """dw = sum_{b,t} dy x: linear in dy and in x."""

    @staticmethod
    def forward(ctx, dy, x, k, dilation, causal, dy_amax=None, x_amax=None):
        dy, dybs = _act3(dy, "grad")
        x, xbs = _act3(x, "conv input")
        dw = _conv_bwd_weight(dy, dybs, x, xbs, (dy.shape[1], x.shape[1], k), k, dilation, causal, dy_amax=dy_amax, x_amax=x_amax)
        ctx.save_for_backward(dy, x)
        ctx.amax = (dy_amax, x_amax)
        ctx.cfg = (k, dilation, causal)
        return dw

    @staticmethod
    def backward(ctx, ddw):
        dy, x = ctx.saved_tensors
        k, dilation, causal = ctx.cfg
        g_dy = ConvFwdDD.apply(x, ddw, None, k, dilation, causal, ctx.amax[1]) if ctx.needs_input_grad[0] else None
        g_x = ConvBwdDataDD.apply(dy, ddw, x.shape[1], k, dilation, causal, ctx.amax[0]) if ctx.needs_input_grad[1] else None
        return g_dy, g_x, None, None, None, None, None


def conv1d_dd(x, w, bias=None, k=1, dilation=1, causal=False):
    """Conv1d (kernel 1 or 3) differentiable to any order on the HIP kernels; the bias is added in the kernel's epilogue."""
    return ConvFwdDD.apply(x, w, bias, k, dilation, bool(causal), _dd_amax(x, w.shape[1], w.shape[0]))


# ------------------------------------------------------------------------------------------- critics: dropout / leaky-ReLU / pooling / penalty
# models/discriminator.py:24-41 between the convolutions and LayerNorms, and train/adversarial_wasserstein_gp.py:305-308.  Each
# op is (piecewise) linear, so its backward is "multiply by the saved factor" / "the adjoint pool", which is again one of these
# ops: differentiable to any order on the HIP kernels (csrc/critic.hip).
class MulConstFn(torch.autograd.Function):
    """y = x * d with d a constant (a saved derivative factor or an injected dropout mask)."""

    @staticmethod
    def forward(ctx, x, d):
        x, d = _c(_dev(x)), _c(d)
        if x.shape != d.shape:
            raise RuntimeError("mul_const: shapes %s and %s differ" % (tuple(x.shape), tuple(d.shape)))
        y = torch.empty_like(x)
        _lib.call("ssv_mul", _p(x), _p(d), _p(y), x.numel(), _stream())
        ctx.save_for_backward(d)
        return y

    @staticmethod
    def backward(ctx, gy):
        (d,) = ctx.saved_tensors
        return MulConstFn.apply(gy, d), None


_DROP_CTR = {}        # device -> 1-element int64 tensor: number of dropout masks drawn so far (advanced on the stream by the kernel)


class ActDropoutFn(torch.autograd.Function):
    """y = dropout(leaky_relu(x, slope), p): one kernel draws the mask (Philox keyed by the device-side call counter, so a
    replayed hipGraph gets a fresh mask each time), applies both and saves the combined factor for the backward."""

    @staticmethod
    def forward(ctx, x, slope, p):
        x = _c(_dev(x, "critic activation"))
        y, d = torch.empty_like(x), torch.empty_like(x)
        ctr = None
        if p > 0:
            key = (x.device.type, x.device.index)
            if key not in _DROP_CTR:
                _DROP_CTR[key] = torch.zeros(1, dtype=torch.int64, device=x.device)
            ctr = _DROP_CTR[key]
        _lib.call("ssv_act_dropout_fwd", _p(x), _p(y), _p(d), x.numel(), float(slope), float(p), _p(ctr), int(torch.cuda.initial_seed()) & 0xFFFFFFFF, _stream())
        ctx.save_for_backward(d)
        return y

    @staticmethod
    def backward(ctx, gy):
        (d,) = ctx.saved_tensors
        return MulConstFn.apply(gy, d), None, None


class AvgPoolFn(torch.autograd.Function):
    """nn.AvgPool1d(kernel_size=k) on (B, C, L) (k = L: the AdaptiveAvgPool1d(1) of discriminator.py:38)."""

    @staticmethod
    def forward(ctx, x, k):
        x = _c(_dev(x, "pool input"))
        B, C, L = x.shape
        y = torch.empty((B, C, L // k), dtype=_F32, device=x.device)
        _lib.call("ssv_avgpool1d_fwd", _p(x), _p(y), B * C, L, k, _stream())
        ctx.cfg = (L, k)
        return y

    @staticmethod
    def backward(ctx, gy):
        L, k = ctx.cfg
        return AvgPoolBwdFn.apply(gy, L, k), None


class AvgPoolBwdFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gy, L, k):
        gy = _c(_dev(gy, "grad"))
        B, C, Lo = gy.shape
        dx = torch.empty((B, C, L), dtype=_F32, device=gy.device)
        _lib.call("ssv_avgpool1d_bwd", _p(gy), _p(dx), B * C, L, k, _stream())
        ctx.k = k
        return dx

    @staticmethod
    def backward(ctx, g):
        return AvgPoolFn.apply(g, ctx.k), None, None


class GradPenaltyFn(torch.autograd.Function):
    """mean_b lam * (||g_b||_2 - 1)^2 over per-sample gradients g (B, ...), train/adversarial_wasserstein_gp.py:305-308."""

    @staticmethod
    def forward(ctx, g, lam):
        g = _c(_dev(g, "penalty gradient"))
        B = g.shape[0]
        n = g.numel() // B
        loss = torch.empty((1,), dtype=_F32, device=g.device)
        coef = torch.empty((B,), dtype=_F32, device=g.device)
        nb = _lib.query("ssv_grad_penalty_workspace", B, n)
        ws = _ws(nb, g.device)
        _lib.call("ssv_grad_penalty_fwd", _p(g), _p(loss), _p(coef), B, n, float(lam), _p(ws), nb, _stream())
        ctx.save_for_backward(g, coef)
        return loss

    @staticmethod
    def backward(ctx, gout):
        g, coef = ctx.saved_tensors
        B = g.shape[0]
        dg = torch.empty_like(g)
        _lib.call("ssv_grad_penalty_bwd", _p(g), _p(coef), _p(_c(gout)), _p(dg), B, g.numel() // B, _stream())
        return dg, None


def mul_const(x, d):
    return MulConstFn.apply(x, d)


def act_dropout(x, slope=1.0, p=0.0):
    """dropout(leaky_relu(x, slope), p); slope = 1 and p = 0 is the identity."""
    if slope == 1.0 and p == 0.0:
        return x
    y = ActDropoutFn.apply(x, slope, p)
    h = getattr(x, "_ssv_amax", None)
    if h is not None and h[1] == x._version and 0.0 <= slope <= 1.0 and p <= 0.25:
        # |leaky_relu(x)| <= |x| and the keep mask scales by 1 / (1 - p) <= 4/3: x's scale list bounds y within the factor-of-two
        # headroom the split-fp16 scaling leaves (max |x| 2^e < 2^15 against fp16's 65504), so the convolution that reads y
        # needs no ssv_absmax launch of its own
        _tag(y, h[0])
    return y


def avg_pool1d(x, k):
    return AvgPoolFn.apply(x, int(k))


def grad_penalty(g, lam):
    return GradPenaltyFn.apply(g, lam)[0]


# ------------------------------------------------------------------------------------------- critics: LayerNorm / gate, twice differentiable
# models/discriminator.py:24-41 (LayerNorm over channels between the critic's convs) and models/TTSModel_dropout.py:63-84
# (the highway gate), for the WGAN-GP critics.  The gradient penalty (train/adversarial_wasserstein_gp.py:300-308) takes the
# gradient of the critic's input gradient, so each op is a pair of Functions: the forward, whose backward is itself a
# Function (the first-order HIP backward kernel) with a hand-written backward of its own (ssv_*_bwd2).  Dropout stays a
# separate torch op between them (a mask product, differentiable as it is).
def _no_third_order(*gs):
    for g in gs:
        if g is not None:
            raise RuntimeError("spoofsv_amd: gradients through the parameter-gradient outputs of a critic backward are not "
                               "implemented (third-order use); only the input-gradient path of the gradient penalty is")


class ChannelLnDD(torch.autograd.Function):
    """y = LayerNorm over channels of a (B, C, T) tensor (no activation), any use up to second order."""

    @staticmethod
    def forward(ctx, x, gamma, beta, y_amax=None):
        x, xbs = _act3(x, "LayerNorm input")
        B, C, L = x.shape
        gamma, beta = _c(gamma), _c(beta)
        y = torch.empty((B, C, L), dtype=_F32, device=x.device)
        stats = torch.empty((B, 2, L), dtype=_F32, device=x.device)
        nb = _lib.query("ssv_channel_ln_act_fwd_workspace", B, C, L)
        ws = _ws(nb, x.device)
        _lib.call("ssv_channel_ln_act_fwd", _p(x), xbs, _p(gamma), _p(beta), _p(y), C * L, _p(y_amax), _p(stats), B, C, L, 0, _p(ws), nb, _stream())
        ctx.save_for_backward(x, gamma, beta, stats)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, gamma, beta, stats = ctx.saved_tensors
        return ChannelLnBwdDD.apply(gy, x, gamma, beta, stats) + (None,)


class ChannelLnBwdDD(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gy, x, gamma, beta, stats):
        x, xbs = _act3(x)
        gy, gybs = _act3(gy, "grad")
        B, C, L = x.shape
        dx = torch.empty((B, C, L), dtype=_F32, device=x.device)
        pg = None if _skip_param_grads(gamma, beta) else torch.empty((3, C), dtype=_F32, device=x.device)      # (None: the partial rows are not summed)
        nb = _lib.query("ssv_channel_ln_act_bwd_workspace", B, C, L)
        ws = _ws(nb, x.device)
        _lib.call("ssv_channel_ln_act_bwd", _p(gy), gybs, _p(x), xbs, _p(stats), _p(gamma), _p(beta), _p(dx), C * L, _p(pg),
                  B, C, L, 0, _p(ws), nb, _stream())
        ctx.save_for_backward(gy, x, gamma, stats)
        ctx.set_materialize_grads(False)
        if pg is None:
            return dx, None, None
        return dx, pg[0], pg[1]

    @staticmethod
    def backward(ctx, v, v_dg, v_db):
        _no_third_order(v_dg, v_db)
        gy, x, gamma, stats = ctx.saved_tensors
        if v is None:
            return None, None, None, None, None
        x, xbs = _act3(x)
        gy, gybs = _act3(gy)
        v, vbs = _act3(v, "grad")
        B, C, L = x.shape
        d_gy = torch.empty((B, C, L), dtype=_F32, device=x.device)
        d_x = torch.empty((B, C, L), dtype=_F32, device=x.device)
        dgamma = torch.empty((C,), dtype=_F32, device=x.device)
        nb = _lib.query("ssv_channel_ln_bwd2_workspace", B, C, L)
        ws = _ws(nb, x.device)
        _lib.call("ssv_channel_ln_bwd2", _p(v), vbs, _p(gy), gybs, _p(x), xbs, _p(stats), _p(gamma), _p(d_gy), C * L, _p(d_x), C * L,
                  _p(dgamma), B, C, L, _p(ws), nb, _stream())
        return d_gy, d_x, dgamma, None, None


class HighwayGateDD(torch.autograd.Function):
    """y = sigmoid(LN1(h[:, :C])) * LN2(h[:, C:]) + (1 - sigmoid(LN1(h[:, :C]))) * x, any use up to second order."""

    @staticmethod
    def forward(ctx, h, x, g1, b1, g2, b2, y_amax=None):
        x, xbs = _act3(x, "gate input")
        B, C, L = x.shape
        h = _dev(h).float().contiguous()
        if tuple(h.shape) != (B, 2 * C, L):
            raise RuntimeError("highway gate: h %s does not match x %s" % (tuple(h.shape), tuple(x.shape)))
        g1, b1, g2, b2 = map(_c, (g1, b1, g2, b2))
        y = torch.empty((B, C, L), dtype=_F32, device=x.device)
        stats = torch.empty((B, 4, L), dtype=_F32, device=x.device)
        _lib.call("ssv_highway_gate_fwd", _p(h), _p(x), xbs, _p(g1), _p(b1), _p(g2), _p(b2), _p(stats), _p(y), C * L, _p(y_amax), B, C, L, _stream())
        ctx.save_for_backward(h, x, g1, b1, g2, b2, stats)
        return y

    @staticmethod
    def backward(ctx, gy):
        return HighwayGateBwdDD.apply(gy, *ctx.saved_tensors) + (None,)


class HighwayGateBwdDD(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gy, h, x, g1, b1, g2, b2, stats):
        x, xbs = _act3(x)
        gy, gybs = _act3(gy, "grad")
        B, C, L = x.shape
        dh = torch.empty((B, 2 * C, L), dtype=_F32, device=x.device)
        dx = torch.empty((B, C, L), dtype=_F32, device=x.device)
        pg = None if _skip_param_grads(g1, b1, g2, b2) else torch.empty((6, C), dtype=_F32, device=x.device)
        nb = _lib.query("ssv_highway_gate_bwd_workspace", B, C, L)
        ws = _ws(nb, x.device)
        _lib.call("ssv_highway_gate_bwd", _p(gy), gybs, _p(x), xbs, _p(g1), _p(b1), _p(g2), _p(b2), _p(h), _p(stats), _p(dh), _p(dx),
                  C * L, _p(pg), B, C, L, _p(ws), nb, _stream())
        ctx.save_for_backward(gy, h, x, g1, b1, g2, b2, stats)
        ctx.set_materialize_grads(False)
        if pg is None:
            return dh, dx, None, None, None, None
        return dh, dx, pg[0], pg[1], pg[2], pg[3]

    @staticmethod
    def backward(ctx, vh, vx, *rest):
        _no_third_order(*rest)
        gy, h, x, g1, b1, g2, b2, stats = ctx.saved_tensors
        if vh is None and vx is None:
            return (None,) * 8
        x, xbs = _act3(x)
        gy, gybs = _act3(gy)
        B, C, L = x.shape
        vh = torch.zeros_like(h) if vh is None else _dev(vh).float().contiguous()
        vx = torch.zeros_like(x) if vx is None else vx
        vx, vxbs = _act3(vx, "grad")
        d_gy = torch.empty((B, C, L), dtype=_F32, device=x.device)
        d_h = torch.empty((B, 2 * C, L), dtype=_F32, device=x.device)
        d_x = torch.empty((B, C, L), dtype=_F32, device=x.device)
        pg = torch.empty((4, C), dtype=_F32, device=x.device)
        nb = _lib.query("ssv_highway_gate_bwd2_workspace", B, C, L)
        ws = _ws(nb, x.device)
        _lib.call("ssv_highway_gate_bwd2", _p(vh), _p(vx), vxbs, _p(gy), gybs, _p(h), _p(x), xbs, _p(stats), _p(g1), _p(b1), _p(g2), _p(b2),
                  _p(d_gy), C * L, _p(d_h), _p(d_x), C * L, _p(pg), B, C, L, _p(ws), nb, _stream())
        return d_gy, d_h, d_x, pg[0], pg[1], pg[2], pg[3], None


def channel_ln_dd(x, gamma, beta):
    if not (_f16() and _bf3_shape(x)):
        return ChannelLnDD.apply(x, gamma, beta)
    ya = _amax_out(x.shape[0], x.shape[2], x.device)
    return _tag(ChannelLnDD.apply(x, gamma, beta, ya), ya)


def highway_gate_dd(h, x, g1, b1, g2, b2):
    if not (_f16() and _bf3_shape(x)):
        return HighwayGateDD.apply(h, x, g1, b1, g2, b2)
    ya = _amax_out(x.shape[0], x.shape[2], x.device)
    return _tag(HighwayGateDD.apply(h, x, g1, b1, g2, b2, ya), ya)


# ------------------------------------------------------------------------------------------- embedding
class TextEmbedFn(torch.autograd.Function):
    """textEmbedding.forward, models/TTSModel.py:25-35: one-hot + Linear == column gather + bias."""

    @staticmethod
    def forward(ctx, ids, w, bias):
        _dev(ids, "text ids")
        ids = ids.long().contiguous()
        B, one, N = ids.shape
        w, bias = _c(w), _c(bias)
        E, V = w.shape
        y = torch.empty((B, E, N), dtype=_F32, device=w.device)
        _lib.call("ssv_text_embed_fwd", _p(ids), _p(w), _p(bias), _p(y), B, N, E, V, _stream())
        if _needs_grad(ctx):
            ctx.save_for_backward(ids)
            ctx.dims = (B, N, E, V)
            ctx.refs = (w, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        (ids,) = ctx.saved_tensors
        B, N, E, V = ctx.dims
        dy = _c(dy)
        dw = gradarena.grad_like(ctx.refs[0])
        db = gradarena.grad_like(ctx.refs[1])
        _lib.call("ssv_text_embed_bwd", _p(ids), _p(dy), _p(dw), _p(db), B, N, E, V, _stream())
        return None, dw, db


# ------------------------------------------------------------------------------------------- attention
class AttentionTrainFn(torch.autograd.Function):
    """models/TTSModel.py:266-270.  kv: (B, 2d, N) text-encoder output (K = first half, V = second,
    :138-139); q: (B, d, T).  Returns (cat(R, Q) (B, 2d, T), A (B, N, T))."""

    @staticmethod
    def forward(ctx, kv, q):
        kv, kvbs = _act3(kv, "K|V")
        q, qbs = _act3(q, "Q")
        B, d2, N = kv.shape
        d, T = q.shape[1], q.shape[2]
        if d2 != 2 * d:
            raise RuntimeError("attention: K|V has %d channels, Q has %d" % (d2, d))
        a = torch.empty((B, N, T), dtype=_F32, device=q.device)
        rq = torch.empty((B, 2 * d, T), dtype=_F32, device=q.device)
        k_ptr = ctypes.c_void_p(kv.data_ptr())
        v_ptr = ctypes.c_void_p(kv.data_ptr() + 4 * d * N)
        _lib.call("ssv_attention_train_fwd_rq", k_ptr, v_ptr, kvbs, _p(q), qbs, _p(a), _p(rq), 2 * d * T, B, d, N, T, _stream())
        if _needs_grad(ctx):
            ctx.save_for_backward(kv, q, a)
        return rq, a

    @staticmethod
    def backward(ctx, drq, da_ext):
        kv, q, a = ctx.saved_tensors
        kv, kvbs = _act3(kv)
        q, qbs = _act3(q)
        B, d2, N = kv.shape
        d, T = q.shape[1], q.shape[2]
        if drq is None:
            drq = torch.zeros((B, 2 * d, T), dtype=_F32, device=q.device)
        drq = _c(drq)
        da_ext = _c(da_ext) if da_ext is not None else None
        dkv = torch.empty((B, 2 * d, N), dtype=_F32, device=q.device)
        dq = torch.empty((B, d, T), dtype=_F32, device=q.device)
        nb = _lib.query("ssv_attention_train_bwd_workspace", B, d, N, T)
        ws = _ws(nb, q.device)
        off_q = 4 * d * T
        _lib.call("ssv_attention_train_bwd", _p(drq), 2 * d * T, _p(da_ext), ctypes.c_void_p(drq.data_ptr() + off_q), 2 * d * T,
                  ctypes.c_void_p(kv.data_ptr()), ctypes.c_void_p(kv.data_ptr() + 4 * d * N), kvbs, _p(q), qbs, _p(a),
                  ctypes.c_void_p(dkv.data_ptr()), ctypes.c_void_p(dkv.data_ptr() + 4 * d * N), 2 * d * N, _p(dq), d * T,
                  B, d, N, T, _p(ws), nb, _stream())
        return dkv, dq


def attention_step(kv, q, pma, a_buf, col):
    """One synthesis step of models/TTSModel.py:281-291: writes attention column ``col`` of ``a_buf``
    (B, N, Tcap) from the last query frame and returns the int64 arg-max positions (B,)."""
    kv, kvbs = _act3(kv, "K|V")
    q, qbs = _act3(q, "Q")
    B, d2, N = kv.shape
    d, T = q.shape[1], q.shape[2]
    pma = pma.long().contiguous()
    out = torch.empty((B,), dtype=torch.int64, device=q.device)
    q_last = ctypes.c_void_p(q.data_ptr() + 4 * (T - 1))
    _lib.call("ssv_attention_step", ctypes.c_void_p(kv.data_ptr()), kvbs, q_last, qbs, T, _p(pma), _p(a_buf),
              a_buf.shape[2], col, None, _p(out), B, d, N, _stream())
    return out


def attention_step_dev(kv, q, pma, a_buf, col_dev):
    """attention_step with the frame index on the device (``col_dev``: int32 tensor of one element) and ``pma`` (int64,
    (B,)) updated in place: the form a captured fixed-shape synthesis step uses (spoofsv_amd/synth.py)."""
    kv, kvbs = _act3(kv, "K|V")
    q, qbs = _act3(q, "Q")
    B, d2, N = kv.shape
    d, T = q.shape[1], q.shape[2]
    _lib.call("ssv_attention_step", ctypes.c_void_p(kv.data_ptr()), kvbs, _p(q), qbs, T, _p(pma), _p(a_buf),
              a_buf.shape[2], 0, _p(col_dev), _p(pma), B, d, N, _stream())


def synth_advance(y, mel_in, col_dev):
    """Feed the frame just synthesised back as the next input column and advance the device-side frame counter."""
    B, F, T = y.shape
    _lib.call("ssv_synth_advance", _p(y), _p(mel_in), _p(col_dev), B, F, T, _stream())


def attention_apply(kv, a_buf, q, T):
    """cat(V @ A[:, :, :T], Q), models/TTSModel.py:293-294, for inference (no tape)."""
    kv, kvbs = _act3(kv, "K|V")
    q, qbs = _act3(q, "Q")
    B, d2, N = kv.shape
    d = q.shape[1]
    rq = torch.empty((B, 2 * d, T), dtype=_F32, device=q.device)
    _lib.call("ssv_attention_apply", ctypes.c_void_p(kv.data_ptr() + 4 * d * N), kvbs, _p(a_buf), a_buf.shape[2],
              _p(rq), 2 * d * T, B, d, N, T, _stream())
    _lib.call("ssv_copy_rows", _p(q), qbs, ctypes.c_void_p(rq.data_ptr() + 4 * d * T), 2 * d * T, B, d * T, _stream())
    return rq


# ------------------------------------------------------------------------------------------- deconv
class DeconvK2S2Fn(torch.autograd.Function):
    """nn.ConvTranspose1d(C, C, kernel_size=2, stride=2), models/TTSModel.py:309,314."""

    @staticmethod
    def forward(ctx, x, w, bias, x_amax=None, y_amax=None):
        x, xbs = _act3(x, "deconv input")
        B, Cin, L = x.shape
        w, bias = _c(w), _c(bias)
        Cout = w.shape[1]
        if w.shape[0] != Cin or w.shape[2] != 2:
            raise RuntimeError("deconv: weight %s does not match input channels %d / kernel 2" % (tuple(w.shape), Cin))
        y = torch.empty((B, Cout, 2 * L), dtype=_F32, device=x.device)
        nb = _lib.query("ssv_deconv1d_k2s2_fwd_workspace", Cin, Cout)
        ws = _ws(nb, x.device)
        # one product over 2 Cout rows on the planes of the 1x1 weight w.view(Cin, 2 Cout, 1) (resident when an optimizer keeps them), which also leaves
        # y's operand-scale list (y_amax, (B, 64)) for the highway layer that follows (before: two stride-2 products, a per-call weight split, an ssv_absmax over y)
        _lib.call("ssv_deconv1d_k2s2_fwd", _p(x), xbs, *_an(x_amax), _p(w), resident.lookup(w.view(Cin, 2 * Cout, 1)), _p(bias), _p(y), Cout * 2 * L,
                  _p(y_amax), 64, B, Cin, Cout, L, _p(ws), nb, _stream())
        if _needs_grad(ctx):
            ctx.save_for_backward(x, w)
            ctx.bias_ref = bias
            ctx.x_amax = x_amax           # (a saved tensor comes back without its Python attributes: the list travels beside it)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        x, xbs = _act3(x)
        dy, dybs = _act3(dy, "grad")
        B, Cin, L = x.shape
        Cout = w.shape[1]
        dx = torch.empty((B, Cin, L), dtype=_F32, device=x.device)
        dw = gradarena.grad_like(w)
        db = gradarena.grad_like(ctx.bias_ref)
        nb = _lib.query("ssv_deconv1d_k2s2_bwd_workspace", B, Cin, Cout)
        ws = _ws(nb, x.device)
        # A ConvTranspose1d(k = 2, s = 2) is the 1x1 convolution u = W2 x, W2 = w.view(Cin, 2 Cout), followed by y(b, o, 2t + j) = u(b, 2o + j, t).  With
        # dy de-interleaved row by row into du (B, 2 Cout, L) -- ONE kernel that also leaves the scale list -- the backward is that convolution's:
        # dx = ONE forward 1x1 product of du with the weight w.view(Cin, 2 Cout, 1) (its planes are resident like any conv weight's), dw = ONE k = 1
        # weight gradient over 2 Cout channels that lands in the weight's own (Cin, Cout, 2) layout, db = row sums of dy.  (Until round 6: two
        # stride-2 data-gradient products behind a per-call weight scan + split, two weight gradients and a torch copy that permuted them.)
        split = _lib.precision() >= 1 and B * L >= 256 and L >= 8 and dybs == Cout * 2 * L
        if split:
            f16 = _f16()                  # the scale lists are read in the split-fp16 mode only
            du = torch.empty((B, 2 * Cout, L), dtype=_F32, device=x.device)
            du_am = torch.empty((B, 64), dtype=_F32, device=x.device) if f16 else None
            _lib.call("ssv_deinterleave2_rows_amax", _p(dy), dybs, _p(du), B, Cout, L, _p(du_am), 64, _stream())
            w2 = w.view(Cin, 2 * Cout, 1)
            _conv_fwd(du, 2 * Cout * L, w2, None, None, dx, Cin * L, 1, 1, 0, du_am)
            x_am = (ctx.x_amax if ctx.x_amax is not None else amax_of(x)) if f16 else None
            _conv_bwd_weight(x, xbs, du, 2 * Cout * L, (Cin, 2 * Cout, 1), 1, 1, 0, dw.view(Cin, 2 * Cout, 1), x_am, du_am)
            _lib.call("ssv_bias_grad", _p(dy), dybs, _p(db), B, Cout, 2 * L, _stream())
            return dx, dw, db, None, None
        _lib.call("ssv_deconv1d_k2s2_bwd", _p(dy), dybs, None, 0, _p(x), xbs, _p(w), _p(dx), Cin * L, _p(dw), _p(db),
                  B, Cin, Cout, L, _p(ws), nb, _stream())
        return dx, dw, db, None, None


# ------------------------------------------------------------------------------------------- losses
class SpecLossFn(torch.autograd.Function):
    """train/ordinary.py:230-231 / :249-250: returns a 2-vector (mean |gt - y|, binary divergence).

    ``seed``: the (2,) gradient vector the caller promises to seed this output's backward with (a training step knows it before the
    forward runs: a constant).  Forward and backward then share ONE pass over (y, gt) (``ssv_spec_losses_fwd_bwd``) and the backward
    hands out the stored dy -- provided it is really called with that tensor, unchanged; anything else takes the separate backward kernel."""

    @staticmethod
    def forward(ctx, y, gt, seed=None):
        y, gt = _c(y), _c(gt)
        if y.shape != gt.shape:
            raise RuntimeError("spec loss: prediction %s vs target %s" % (tuple(y.shape), tuple(gt.shape)))
        n = y.numel()
        out = torch.empty((2,), dtype=_F32, device=y.device)
        nb = _lib.query("ssv_spec_losses_workspace", n)
        ws = _ws(nb, y.device)
        ctx.fused = None
        if seed is not None and ctx.needs_input_grad[0]:
            if seed.dtype != _F32 or seed.numel() != 2 or seed.device != y.device or not seed.is_contiguous():
                raise RuntimeError("spec loss: the promised gradient seed must be a contiguous float32 2-vector on the prediction's device")
            dy = torch.empty_like(y)
            _lib.call("ssv_spec_losses_fwd_bwd", _p(y), _p(gt), n, _p(seed), _p(out), _p(dy), _p(ws), nb, _stream())
            ctx.fused = (dy, seed.data_ptr(), seed._version)
            ctx.seed_ref = seed
        else:
            _lib.call("ssv_spec_losses_fwd", _p(y), _p(gt), n, _p(out), _p(ws), nb, _stream())
        ctx.save_for_backward(y, gt)
        return out

    @staticmethod
    def backward(ctx, gout):
        y, gt = ctx.saved_tensors
        f = ctx.fused
        if f is not None and gout.data_ptr() == f[1] and ctx.seed_ref._version == f[2] and gout.is_contiguous():
            return f[0], None, None
        gout = _c(gout)
        dy = torch.empty_like(y)
        _lib.call("ssv_spec_losses_bwd", _p(y), _p(gt), y.numel(), _p(gout), _p(dy), _stream())
        return dy, None, None


class GuidedAttLossFn(torch.autograd.Function):
    """train/ordinary.py:232-234: sum(A * W[:N, :T]) / (B*N*T) as a 1-vector."""

    @staticmethod
    def forward(ctx, a, gaw):
        a, gaw = _c(a), _c(gaw)
        B, N, T = a.shape
        if gaw.shape[0] < N or gaw.shape[1] < T:
            raise RuntimeError("guided attention: weight %s smaller than attention %s" % (tuple(gaw.shape), tuple(a.shape)))
        out = torch.empty((1,), dtype=_F32, device=a.device)
        nb = _lib.query("ssv_guided_att_loss_workspace", B, N, T)
        ws = _ws(nb, a.device)
        _lib.call("ssv_guided_att_loss_fwd", _p(a), _p(gaw), gaw.shape[1], _p(out), B, N, T, _p(ws), nb, _stream())
        ctx.save_for_backward(gaw)
        ctx.dims = (B, N, T)
        return out

    @staticmethod
    def backward(ctx, gout):
        (gaw,) = ctx.saved_tensors
        B, N, T = ctx.dims
        gout = _c(gout)
        da = torch.empty((B, N, T), dtype=_F32, device=gaw.device)
        _lib.call("ssv_guided_att_loss_bwd", _p(gaw), gaw.shape[1], _p(gout), _p(da), B, N, T, _stream())
        return da, None


# ------------------------------------------------------------------------------------------- functional
def _bf3_shape(x):
    """The conv kernels run their MFMA arithmetic only from B * L >= 128 on (speaker codes, L = 1, take the fp32 kernel)."""
    return x.dim() == 3 and x.shape[0] * x.shape[2] >= 128


def highway_conv1d(x, w, bias, g1, b1, g2, b2, k, dilation, causal):
    if not (_f16() and _bf3_shape(x)):
        return HighwayConvFn.apply(x, w, bias, g1, b1, g2, b2, k, dilation, causal)
    ya = _amax_out(x.shape[0], x.shape[2], x.device)
    return _tag(HighwayConvFn.apply(x, w, bias, g1, b1, g2, b2, k, dilation, causal, amax_of(x), ya), ya)


RELU_TAP = None      # diagnostics / tests: a list that receives (y > 0) of every fused-ReLU output, in call order


def pointwise_conv_ln_act(x, w, bias, gamma, beta, s=None, act=0):
    if not (_f16() and _bf3_shape(x)):
        y = PointwiseConvLnActFn.apply(x, w, bias, gamma, beta, s, act)
    else:
        ya = _amax_out(x.shape[0], x.shape[2], x.device)
        y = _tag(PointwiseConvLnActFn.apply(x, w, bias, gamma, beta, s, act, amax_of(x), ya), ya)
    if RELU_TAP is not None and act == 1:
        RELU_TAP.append(y.detach() > 0)
    return y


def conv1d(x, w, bias, k=1, dilation=1, causal=False):
    if not (_f16() and _bf3_shape(x)):
        return Conv1dFn.apply(x, w, bias, k, dilation, causal)
    return Conv1dFn.apply(x, w, bias, k, dilation, causal, amax_of(x))


def text_embed(ids, w, bias):
    return TextEmbedFn.apply(ids, w, bias)


def attention_train(kv, q):
    return AttentionTrainFn.apply(kv, q)


def deconv1d_k2s2(x, w, bias):
    if not (_f16() and _bf3_shape(x)):
        return DeconvK2S2Fn.apply(x, w, bias)
    ya = torch.empty((x.shape[0], 64), dtype=_F32, device=x.device)
    return _tag(DeconvK2S2Fn.apply(x, w, bias, amax_of(x), ya), ya)


def spec_losses(y, gt):
    out = SpecLossFn.apply(y, gt)
    return out[0], out[1]


def spec_losses_vec(y, gt, seed=None):
    """(l1, binary divergence) as ONE 2-vector: a training step seeds its backward with a constant gradient vector for it instead of
    summing two selected scalars (each select's backward is a zeros + a scatter + an add of two 2-vectors: tiny launches in a row between
    the end of the forward and the start of the backward).  ``seed``: that vector, when the caller already has it (see SpecLossFn)."""
    return SpecLossFn.apply(y, gt, seed)


def guided_att_loss_vec(a, gaw):
    return GuidedAttLossFn.apply(a, gaw)


def guided_att_loss(a, gaw):
    return GuidedAttLossFn.apply(a, gaw)[0]
