#!/usr/bin/env python3
"""Diagnostic (GPU box): which Python call sites issue the SMALL launches of the ordinary training step -- torch's fill / copy / add
kernels, hipMemcpy and ssv_absmax -- so that each can be removed or folded into a neighbouring kernel.
One eager Text2Mel + SSRN iteration under torch.profiler (with_stack); prints, per operator name, the call sites inside this
repository (innermost first) and how often each fired.   usage: python tools/small_launch_sites.py [text2mel|ssrn]"""
import collections, os, sys
import torch
from torch.profiler import ProfilerActivity, profile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spoofsv_amd import ops, train
from spoofsv_amd.tts import SSRN, melSyn

WATCH = ("aten::fill_", "aten::zero_", "aten::copy_", "aten::add", "aten::add_", "aten::mul", "aten::cat", "aten::sum", "aten::clone",
         "aten::contiguous", "aten::zeros", "aten::zeros_like", "aten::full_like", "aten::ones_like", "aten::stack", "aten::div", "aten::neg")


def site(ev):
    frames = [f for f in (ev.stack or []) if ("/spoofsv_amd/" in f or "/bench.py" in f or "/tools/" in f)]
    return " <- ".join(f.split("/")[-1].strip() for f in frames[:3]) or "(autograd engine / no repo frame)"


def main():
    kinds = sys.argv[1:] or ["text2mel", "ssrn"]
    dev = torch.device("cuda", 0)
    B = 32
    for kind in kinds:
        torch.manual_seed(0)
        if kind == "text2mel":
            model = melSyn(34, True, 200, textemb_dim=128, freq_bins=80, hidden_dim=256)
            batch = train.synthetic_text2mel_batch(B, N=186, T=325, seed=1)
            gaw = train.guided_attention_mat(186, 325).to(dev)
        else:
            model = SSRN(80, 513, 256)
            batch = train.synthetic_ssrn_batch(B, T=325, seed=1)
            gaw = None
        model.apply(train.init_weights)
        model = model.to(dev).train()
        opt = train.FusedAdam(model.parameters(), 2e-4, (0.5, 0.9), 1e-6, capturable=True)
        opt.refresh_resident_weights()
        st = train.TrainStep(kind, model, opt, [b.to(dev) for b in batch], gaw, None, graph=False, defer_wgrad=True)
        for _ in range(2):
            st()
        torch.cuda.synchronize()
        by = collections.defaultdict(collections.Counter)
        import traceback
        from torch.utils._python_dispatch import TorchDispatchMode

        class Sites(TorchDispatchMode):
            def __torch_dispatch__(self, func, types, args=(), kwargs=None):
                name = str(func).replace(".default", "")
                if any(k in name for k in ("fill", "zero", "copy", "add", "mul", "cat", "sum", "clone", "full", "ones", "stack", "div", "neg")):
                    fr = [f for f in traceback.extract_stack() if ("/spoofsv_amd/" in f.filename or "/bench.py" in f.filename)]
                    shp = [tuple(a.shape) for a in args if isinstance(a, torch.Tensor)][:2]
                    where = " <- ".join("%s:%d %s" % (os.path.basename(f.filename), f.lineno, f.name) for f in reversed(fr[-3:])) or "(no repo frame)"
                    by[name]["%s  %s" % (where, shp)] += 1
                return func(*args, **(kwargs or {}))

        from spoofsv_amd import _lib
        orig_call = _lib.call

        def call(name, *a):
            if name == "ssv_absmax":
                fr = [f for f in traceback.extract_stack() if "/spoofsv_amd/" in f.filename]
                by["ssv_absmax (python)"][" <- ".join("%s:%d %s" % (os.path.basename(f.filename), f.lineno, f.name) for f in reversed(fr[-5:-1])) + "  n=%d" % a[3]] += 1
            return orig_call(name, *a)
        _lib.call = ops._lib.call = call
        with Sites():
            st()
            torch.cuda.synchronize()
        _lib.call = ops._lib.call = orig_call
        print("==== %s: one eager iteration" % kind)
        for name in sorted(by):
            print("%s: %d" % (name, sum(by[name].values())))
            for s_, n in by[name].most_common(12):
                print("   %4d  %s" % (n, s_))


if __name__ == "__main__":
    main()
