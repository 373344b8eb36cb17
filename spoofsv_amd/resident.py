"""Resident pre-split conv weights (include/ssv_hip.h, "Resident pre-split weights").

The split-bf16 conv kernels read weights as bf16 hi/lo planes in MFMA fragment order.  Splitting a weight is one
small launch per conv call (forward order for the forward, transposed order for the data gradient) -- about a
hundred launches per training step.  A ``ResidentWeights`` keeps one plane buffer per conv weight and refreshes
all of them with ONE launch (``ssv_conv_pack_multi``); ``FusedAdam`` does that right after its update, so the next
step's convolutions find current planes and skip their own split.

Validity is checked per call and is conservative: planes are used only while the weight tensor still has the
address, shape and autograd version (``Tensor._version``) it had when the planes were written.  Any in-place torch
op on the weight (``load_state_dict``, ``init``) bumps the version and the conv falls back to splitting the weight
itself until the next refresh.  Writers that go around the version counter (``p.data`` aliases, collectives) must
call ``invalidate``.
"""
import ctypes

import numpy as np
import torch

from . import _lib

_REG = {}            # weight data_ptr -> _Entry


class _Entry:
    __slots__ = ("param", "planes", "version", "shape", "ptr", "owner")


def lookup(w):
    """ctypes pointer to current planes of weight ``w`` (a (Cout, Cin, k) tensor), or None."""
    e = _REG.get(w.data_ptr())
    if e is None or e.version != w._version or e.shape != tuple(w.shape):
        return None
    return e.ptr


def invalidate(params=None):
    """Forget the planes of ``params`` (all registered weights when None); they are rebuilt by the next refresh."""
    if params is None:
        for e in _REG.values():
            e.version = -1
        return
    for p in params:
        e = _REG.get(p.data_ptr())
        if e is not None:
            e.version = -1


def eligible(p):
    return p.is_cuda and p.dtype == torch.float32 and p.dim() == 3 and p.shape[2] in (1, 2, 3) and p.is_contiguous()


def pack_shape(p):
    """The (Cout, Cin, k) conv weight whose planes are kept for parameter ``p``: itself, or -- for the (Cin, Cout, 2) weight of a
    ConvTranspose1d(k = 2, s = 2) -- the 1x1 weight ``p.view(Cin, 2 Cout, 1)`` that its data gradient is a forward convolution with
    (ops.DeconvK2S2Fn.backward looks the planes up under that view: same address, same version counter)."""
    return (p.shape[0], 2 * p.shape[1], 1) if p.shape[2] == 2 else tuple(p.shape)


class ResidentWeights:
    """Plane buffers + the device job table for a fixed list of conv weights."""

    def __init__(self, params):
        self.params = [p for p in params if eligible(p)]
        self._key = None
        self._jobs = None
        self._nblocks = 0

    def _build(self):
        key = tuple(p.data_ptr() for p in self.params)
        if key == self._key:
            return
        for e in [e for e in _REG.values() if any(e.param is p for p in self.params)]:
            _REG.pop(e.param.data_ptr(), None)
        n = len(self.params)
        dev = self.params[0].device
        sizes = [int(_lib.query("ssv_conv_pack_bytes", *pack_shape(p))) for p in self.params]
        self._planes = [torch.empty(s, dtype=torch.uint8, device=dev) for s in sizes]
        vp = ctypes.c_void_p
        w = (vp * n)(*[p.data_ptr() for p in self.params])
        pl = (vp * n)(*[t.data_ptr() for t in self._planes])
        ci = lambda k: (ctypes.c_int * n)(*[pack_shape(p)[k] for p in self.params])
        jobs = (_lib.PackJob * (2 * n))()
        nblocks = _lib.lib().ssv_conv_pack_plan(n, w, pl, ci(0), ci(1), ci(2), jobs)
        if nblocks < 0:
            raise RuntimeError("ssv_conv_pack_plan: " + _lib.lib().ssv_last_error().decode())
        raw = np.frombuffer(bytes(jobs), dtype=np.uint8).copy()
        host = torch.from_numpy(raw).pin_memory()
        self._jobs = torch.empty(raw.size, dtype=torch.uint8, device=dev)
        self._jobs.copy_(host, non_blocking=False)
        self._nblocks, self._njobs = nblocks, 2 * n
        self._ws_bytes = int(_lib.query("ssv_conv_pack_multi_workspace", 2 * n))
        self._ws = torch.empty(max(self._ws_bytes, 256), dtype=torch.uint8, device=dev)      # the weights' partial maxima (split-fp16)
        for p, t in zip(self.params, self._planes):
            e = _Entry()
            e.param, e.planes, e.version, e.shape, e.ptr = p, t, -1, pack_shape(p), ctypes.c_void_p(t.data_ptr())
            e.owner = self
            _REG[p.data_ptr()] = e
        self._key = key

    def refresh(self, stream):
        """Re-split every weight on ``stream`` (one launch) and mark the planes current."""
        if not self.params:
            return
        self._build()
        _lib.call("ssv_conv_pack_multi", ctypes.c_void_p(self._jobs.data_ptr()), self._njobs, self._nblocks,
                  ctypes.c_void_p(self._ws.data_ptr()), self._ws_bytes, stream)
        for p in self.params:
            e = _REG.get(p.data_ptr())
            if e is not None and e.owner is self:
                e.version = p._version


_FROZEN = {}         # id(module) -> ResidentWeights of an inference model


def ensure(module, stream):
    """Inference helper: keep the conv weights of ``module`` resident.  The first call builds the planes; later calls cost
    one version check per weight and re-split (one launch) only if a weight was modified since (``load_state_dict``, init)."""
    # A weight has ONE set of planes.  When the module's weights already belong to a training optimizer's ResidentWeights
    # (FusedAdam re-splits those after every update -- inside the captured hipGraph when the step is replayed from one),
    # use and, if stale, refresh THOSE: a second set registered here would be written once and then go stale behind the
    # version check, because FusedAdam updates weights through raw pointers.
    mine = [p for p in module.parameters() if eligible(p)]
    owners = []
    for p in mine:
        e = _REG.get(p.data_ptr())
        if e is None or e.param is not p or e.owner is None:
            owners = None
            break
        if not any(o is e.owner for o in owners):
            owners.append(e.owner)
    if owners and not any(o is _FROZEN.get(id(module)) for o in owners):
        if any(lookup(p) is None for p in mine):
            for o in owners:
                o.refresh(stream)
        return owners[0]
    rw = _FROZEN.get(id(module))
    if rw is None or [id(p) for p in rw.params] != [id(p) for p in mine]:
        while len(_FROZEN) >= 8:                 # a handful of inference models at most; older plane sets are dropped
            old = _FROZEN.pop(next(iter(_FROZEN)))
            for p in old.params:
                e = _REG.get(p.data_ptr())
                if e is not None and e.param is p:
                    _REG.pop(p.data_ptr())
        rw = _FROZEN[id(module)] = ResidentWeights(list(module.parameters()))
    if any(lookup(p) is None for p in rw.params):
        rw.refresh(stream)
    return rw
