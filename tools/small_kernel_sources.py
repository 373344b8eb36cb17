#!/usr/bin/env python3
"""Diagnostic (GPU box): which Python lines issue the tiny torch ops (fills, copies, element-wise) of one eager training step."""
import os, sys, collections, traceback
import torch
from torch.utils._python_dispatch import TorchDispatchMode
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

class Spy(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.cnt = collections.Counter()
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(k in name for k in ("copy_", "fill_", "zero_", "zeros", "ones", "add.", "mul.", "stack", "clone", "full", "cat", "sum", "div")):
            big = any(isinstance(a, torch.Tensor) and a.is_cuda for a in args)
            fr = [f for f in traceback.extract_stack() if ("spoofsv_amd" in f.filename or f.filename.endswith("bench.py"))]
            where = "%s:%d %s" % (os.path.basename(fr[-1].filename), fr[-1].lineno, fr[-1].line) if fr else "(no package frame)"
            shape = tuple(args[0].shape) if args and isinstance(args[0], torch.Tensor) else ()
            self.cnt[(name, where[:100], str(shape)[:30], big)] += 1
        return func(*args, **(kwargs or {}))

dev = torch.device("cuda:0")
for kind in ("text2mel", "ssrn"):
    tr = bench.Trainer(kind, 32, dev, 0, 1, False)
    for _ in range(2): tr.step()
    torch.cuda.synchronize()
    spy = Spy()
    with spy:
        tr.step()
    torch.cuda.synchronize()
    print("==", kind)
    for (nm, where, shape, big), c in spy.cnt.most_common(40):
        print("%3d  %-28s %-28s %s" % (c, nm, shape, where))
