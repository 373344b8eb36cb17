#!/usr/bin/env python3
"""Diagnostic (GPU box): which call sites still compute an operand's scale list with an ssv_absmax launch (ops.amax_of fallback) in one eager
generator iteration and one critic iteration of the WGAN-GP trainer (split-fp16 mode), Text2Mel and SSRN, B = 8."""
import collections, os, sys, traceback
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spoofsv_amd import _lib, ops, train
from spoofsv_amd.critic import linDisc, melDisc
from spoofsv_amd.tts import SSRN, melSyn
sites = collections.Counter()
orig_call = _lib.call
def call(name, *a):
    if name == "ssv_absmax":
        st = traceback.extract_stack(limit=9)[:-1]
        key = " <- ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in reversed(st) if "spoofsv_amd" in f.filename)[:160]
        sites[(PHASE, key)] += 1
    return orig_call(name, *a)
_lib.call = call
ops._lib.call = call
# who PRODUCED the tensor whose list is missing: every autograd Function of ops tags what its forward / backward returns
import inspect
def _tagging(cls, which):
    fn = getattr(cls, which)
    def wrapped(*a, **k):
        out = fn(*a, **k)
        for t in (out if isinstance(out, tuple) else (out,)):
            if isinstance(t, torch.Tensor):
                try: t._ssv_src = "%s.%s" % (cls.__name__, which)
                except Exception: pass
        return out
    setattr(cls, which, staticmethod(wrapped))
for _, cls in inspect.getmembers(ops, inspect.isclass):
    if issubclass(cls, torch.autograd.Function) and cls is not torch.autograd.Function:
        _tagging(cls, "forward"); _tagging(cls, "backward")
orig_amax = ops.amax_of
def amax_of(x):
    h = getattr(x, "_ssv_amax", None)
    if not (h is not None and h[1] == x._version and h[0].shape[0] == x.shape[0]):
        srcs[(PHASE, getattr(x, "_ssv_src", "torch / unknown") + " " + str(tuple(x.shape[1:])))] += 1
    return orig_amax(x)
srcs = collections.Counter()
ops.amax_of = amax_of
dev = "cuda:0"
for kind in ("text2mel", "ssrn"):
    torch.manual_seed(0)
    if kind == "text2mel":
        model, disc = melSyn(34, True, 200, 128, 80, 256), melDisc(80, 128)
        data = train.synthetic_text2mel_batch(8, 186, 325, seed=0, device=dev); gaw = train.guided_attention_mat(186, 325, device=dev)
    else:
        model, disc = SSRN(80, 513, 256), linDisc(513, 128)
        data = train.synthetic_ssrn_batch(8, 325, seed=0, device=dev); gaw = None
    model.apply(train.init_weights); disc.apply(train.init_weights); model.to(dev).train(); disc.to(dev).train()
    og = train.FusedAdam(model.parameters(), 2e-4, (0.5, 0.9), 1e-6, capturable=True); od = train.FusedAdam(disc.parameters(), 2e-4, (0.5, 0.9), 1e-6, capturable=True)
    og.refresh_resident_weights(); od.refresh_resident_weights()
    st = train.AdversarialGraphStep(kind, model, disc, og, od, data, gaw, 10.0, None, None, graph=False)
    PHASE = kind + " warm"; st.g_step(); st.d_step()
    PHASE = kind + " G"; st.g_step()
    PHASE = kind + " D"; st.d_step()
    torch.cuda.synchronize()
for ph in sorted(set(k[0] for k in sites)):
    if "warm" in ph: continue
    tot = sum(v for k, v in sites.items() if k[0] == ph)
    print("== %s: %d ssv_absmax launches" % (ph, tot))
    for (p, key), v in sorted(sites.items(), key=lambda kv: -kv[1]):
        if p == ph: print("   %3d  %s" % (v, key))
print("\nproducers of the tensors that had no list (phase, producer, shape per item):")
for (ph, key), v in sorted(srcs.items(), key=lambda kv: (kv[0][0], -kv[1])):
    if "warm" not in ph: print("   %-12s %3d  %s" % (ph, v, key))
