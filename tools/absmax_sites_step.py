#!/usr/bin/env python3
"""Diagnostic (GPU box): which call sites of one ORDINARY training iteration (Text2Mel and SSRN, split-fp16, eager, B = 8) still compute an
operand's scale list with an ssv_absmax launch (ops.amax_of fallback), and which kernel produced the tensor that had no list.
(tools/absmax_sites.py does the same for the WGAN-GP iterations.)"""
import collections, inspect, os, sys, traceback
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spoofsv_amd import ops, train
from spoofsv_amd.tts import SSRN, melSyn

srcs, sites = collections.Counter(), collections.Counter()
PHASE = "warm"


def _tagging(cls, which):
    fn = getattr(cls, which)

    def wrapped(*a, **k):
        out = fn(*a, **k)
        for t in (out if isinstance(out, tuple) else (out,)):
            if isinstance(t, torch.Tensor):
                try:
                    t._ssv_src = "%s.%s" % (cls.__name__, which)
                except Exception:
                    pass
        return out
    setattr(cls, which, staticmethod(wrapped))


for _, cls in inspect.getmembers(ops, inspect.isclass):
    if issubclass(cls, torch.autograd.Function) and cls is not torch.autograd.Function:
        _tagging(cls, "forward"); _tagging(cls, "backward")
orig_amax = ops.amax_of


def amax_of(x):
    h = getattr(x, "_ssv_amax", None)
    if not (h is not None and h[1] == x._version and h[0].shape[0] == x.shape[0]):
        st = traceback.extract_stack(limit=8)[:-1]
        key = " <- ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in reversed(st) if "spoofsv_amd" in f.filename)[:150]
        srcs[(PHASE, getattr(x, "_ssv_src", "torch / unknown") + " " + str(tuple(x.shape[1:])), key)] += 1
    return orig_amax(x)


ops.amax_of = amax_of
# every C-ABI call of the iteration, by entry and call site (the small reduction / row-sum launches hide behind these)
from spoofsv_amd import _lib
calls = collections.Counter()
orig_call = _lib.call
WATCH = ("ssv_conv1d_bwd_weight", "ssv_rowsum", "ssv_reduce_partial_rows", "ssv_reduce_slabs", "ssv_pointwise_conv_bwd_weight", "ssv_deconv1d_k2s2_bwd_weight")


def call(name, *a):
    key = ""
    if name in WATCH:
        st = traceback.extract_stack(limit=7)[:-1]
        key = " <- ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in reversed(st) if "spoofsv_amd" in f.filename)[:120]
        key += "  ints=" + ",".join(str(v) for v in a if isinstance(v, int) and 0 < v < 100000)[:60]
    calls[(PHASE, name, key)] += 1
    return orig_call(name, *a)


_lib.call = call
ops._lib.call = call
dev = "cuda:0"
for kind in ("text2mel", "ssrn"):
    torch.manual_seed(0)
    if kind == "text2mel":
        model = melSyn(34, True, 200, 128, 80, 256)
        data = list(train.synthetic_text2mel_batch(8, 186, 325, seed=0, device=dev)); gaw = train.guided_attention_mat(186, 325, device=dev)
    else:
        model = SSRN(80, 513, 256)
        data = list(train.synthetic_ssrn_batch(8, 325, seed=0, device=dev)); gaw = None
    model.apply(train.init_weights); model.to(dev).train()
    opt = train.FusedAdam(model.parameters(), 2e-4, (0.5, 0.9), 1e-6, capturable=True)
    opt.refresh_resident_weights()
    st = train.TrainStep(kind, model, opt, data, gaw, None, graph=False, defer_wgrad=True)
    PHASE = kind + " warm"; st()
    PHASE = kind; st()
    torch.cuda.synchronize()
for ph in ("text2mel", "ssrn"):
    rows = [(k, v) for k, v in srcs.items() if k[0] == ph]
    print("== %s: %d ssv_absmax launches per iteration" % (ph, sum(v for _, v in rows)))
    for (p, src, key), v in sorted(rows, key=lambda kv: -kv[1]):
        print("   %3d  %-50s %s" % (v, src, key))
for ph in ("text2mel", "ssrn"):
    print("== %s: C-ABI calls of one iteration" % ph)
    for (p, name, key), v in sorted(((k, v) for k, v in calls.items() if k[0] == ph), key=lambda kv: (kv[0][1], kv[0][2])):
        print("   %3d  %-44s %s" % (v, name, key))
