"""Flat gradient arena: every parameter gradient of one model lives in ONE fp32 buffer, bucket after bucket.

Why: the data-parallel exchange (the reference's nn.DataParallel reduce, train/ordinary.py:165-173) all-reduces
a few large flat messages over RCCL/xGMI.  Packing 214 gradient tensors into a flat bucket after backward costs a copy of
the whole gradient and delays the first collective to the end of backward.  Here the backward kernels WRITE their parameter
gradients straight into the arena (``ops`` asks ``view`` / ``block`` for the destination instead of allocating), autograd
adopts those views as ``p.grad``, and a bucket -- a contiguous arena range ordered by when its gradients become final --
can be handed to the collective as soon as the backward segment that produces it has been enqueued.

A *plan* is ``[(bucket name, [group, ...]), ...]`` with ``group`` a list of parameters laid out adjacently in that order:
the LayerNorm / bias gradients of one fused operator come out of the kernel as one ``(rows, C)`` block, so their slots must
be neighbours (``block``).  Group starts are 16-byte aligned (the Adam kernel's vector path).
"""
import weakref

import torch

_SLOTS = {}      # parameter data_ptr -> (arena weakref, offset, numel, parameter weakref)


def _entry(t):
    e = _SLOTS.get(t.data_ptr())
    if e is None or e[2] != t.numel():          # same storage start and size (nn.Linear weights arrive with a trailing unit axis)
        return None
    arena, param = e[0](), e[3]()
    if arena is None or param is None or param.data_ptr() != t.data_ptr():
        # the arena or the parameter it was built for is gone (its memory may belong to another tensor by now), or the
        # parameter was moved: the slot is stale
        _SLOTS.pop(t.data_ptr(), None)
        return None
    g = param.grad
    if g is not None and g.data_ptr() == arena.flat.data_ptr() + 4 * e[1]:
        # The slot already HOLDS a gradient of this parameter (a backward without zero_grad(set_to_none=True) before it).  A
        # kernel writing there again would clobber what p.grad aliases and autograd would then add the slot to itself.  Hand
        # out no slot: the operator allocates a fresh tensor and autograd accumulates it into p.grad -- i.e. into the arena.
        return None
    if e[1] in arena.claimed:
        # The slot was handed to another user of this parameter EARLIER IN THE SAME BACKWARD: autograd's AccumulateGrad runs only
        # after all users of a leaf have delivered, so p.grad is still None and the test above cannot see it.  The mark is set
        # by view() / block() and cleared by the parameter's post-accumulate hook.  Same answer: no slot, autograd sums.
        return None
    return arena, e[1], e[2]


def view(w, claim=True):
    """The arena slot of parameter ``w`` (looked up by address and shape) as a tensor of ``w``'s shape, or None.  ``claim``: the
    caller is a backward operator about to write the gradient there (the slot is not handed out again before autograd has
    accumulated it); False only looks."""
    e = _entry(w)
    if e is None:
        return None
    arena, off, n = e
    if claim:
        arena.claimed.add(off)
    return arena.flat[off:off + n].view(w.shape)


def claimed(w):
    """True when ``w`` has an arena slot that a backward operator already took in the backward in progress."""
    e = _SLOTS.get(w.data_ptr())
    arena = e[0]() if e is not None else None
    return arena is not None and e[1] in arena.claimed


def block(params, rows, cols, claim=True):
    """One ``(rows, cols)`` view covering the slots of ``params`` when they are adjacent in this order and fill it exactly,
    else None.  ``claim`` as in ``view``."""
    first = _entry(params[0])
    if first is None:
        return None
    arena, off0, _ = first
    off = off0
    offs = []
    for p in params:
        e = _entry(p)
        if e is None or e[0] is not arena or e[1] != off:
            return None
        offs.append(off)
        off += e[2]
    if off - off0 != rows * cols:
        return None
    if claim:
        arena.claimed.update(offs)
    return arena.flat[off0:off].view(rows, cols)


def grad_like(w):
    v = view(w)
    return v if v is not None else torch.empty_like(w)


def grad_block(params, rows, cols, device):
    v = block(params, rows, cols)
    return v if v is not None else torch.empty((rows, cols), dtype=torch.float32, device=device)


def _remove_hooks(hooks):
    while hooks:
        hooks.pop().remove()


class GradArena:
    def __init__(self, plan, device=None, align=4):
        self.names, self.ranges, self.slots = [], [], {}
        self.params = []
        self.claimed = set()       # offsets of slots handed to a backward operator and not yet accumulated by autograd
        self._hooks = []
        off = 0
        seen = set()
        for name, groups in plan:
            start = off
            for g in groups:
                off = (off + align - 1) // align * align
                for p in g:
                    if id(p) in seen:
                        raise RuntimeError("GradArena: parameter listed twice in the plan")
                    seen.add(id(p))
                    if p.dtype != torch.float32 or not p.is_contiguous():
                        raise RuntimeError("GradArena needs dense float32 parameters")
                    self.slots[id(p)] = (off, p.numel())
                    self.params.append(p)
                    off += p.numel()
            off = (off + 63) // 64 * 64            # buckets start on 256-byte boundaries
            self.names.append(name)
            self.ranges.append((start, off))
        dev = device if device is not None else self.params[0].device
        self.flat = torch.zeros(max(off, 1), dtype=torch.float32, device=dev)
        ref = weakref.ref(self)

        def unclaim(o):
            # autograd rejects any hook result that is not None, also once the arena is gone: always return None
            def hook(_p):
                a = ref()
                if a is not None:
                    a.claimed.discard(o)
                return None
            return hook

        for p in self.params:
            o, n = self.slots[id(p)]
            _SLOTS[p.data_ptr()] = (ref, o, n, weakref.ref(p))
            if p.requires_grad:
                # AccumulateGrad has run for p (all of its users in this backward have delivered): its slot may be handed out again
                self._hooks.append(p.register_post_accumulate_grad_hook(unclaim(o)))
        # an arena that is dropped without release() must not leave its hooks on the parameters (they would pile up with
        # every new arena over the same model): the finalizer owns the handle list, not the arena
        self._finalizer = weakref.finalize(self, _remove_hooks, self._hooks)

    def release(self):
        """Detach the arena from its parameters: hooks removed, slots forgotten.  Idempotent."""
        _remove_hooks(self._hooks)
        for p in self.params:
            e = _SLOTS.get(p.data_ptr())
            if e is not None and e[0]() is self:
                _SLOTS.pop(p.data_ptr())

    def bucket(self, i):
        a, b = self.ranges[i]
        return self.flat[a:b]

    def bucket_of(self, p):
        off = self.slots[id(p)][0]
        for i, (a, b) in enumerate(self.ranges):
            if a <= off < b:
                return i
        raise KeyError("parameter is not in the arena")

    def slot(self, p):
        o, n = self.slots[id(p)]
        return self.flat[o:o + n].view(p.shape)

    @torch.no_grad()
    def adopt(self, params=None):
        """Make every ``p.grad`` the arena slot: gradients an operator produced elsewhere are copied in (one small copy each
        -- the fused operators of ``ops`` never take this path), missing gradients become zero slots."""
        moved = 0
        for p in (self.params if params is None else params):
            o, n = self.slots[id(p)]
            dst = self.flat[o:o + n].view(p.shape)
            g = p.grad
            if g is None:
                dst.zero_()
                p.grad = dst
            elif g.data_ptr() != dst.data_ptr():
                dst.copy_(g)
                p.grad = dst
                moved += 1
            self.claimed.discard(o)        # (a backward that never reached AccumulateGrad must not leave the slot marked)
        return moved
