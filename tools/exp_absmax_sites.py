"""Which operands of one Text2Mel + SSRN step have no producer-written scale list (each costs an ssv_absmax launch)?  GPU box."""
import collections, sys, traceback
import torch
sys.path.insert(0, ".")
import bench
from spoofsv_amd import ops

sites = collections.Counter()
orig = ops.amax_of
def spy(x):
    h = getattr(x, "_ssv_amax", None)
    if not (h is not None and h[1] == x._version and h[0].shape[0] == x.shape[0]):
        fr = [f for f in traceback.extract_stack()[:-1] if "spoofsv_amd" in f.filename][-3:]
        sites[(tuple(x.shape), " < ".join(f"{f.name}:{f.lineno}" for f in reversed(fr)))] += 1
    return orig(x)
ops.amax_of = spy
dev = torch.device("cuda:0")
for kind in ("text2mel", "ssrn"):
    t = bench.Trainer(kind, 32, dev, 0, 1, use_graph=False)
    t.prepare()
    t.step(); torch.cuda.synchronize()
    sites.clear()
    t.step(); torch.cuda.synchronize()
    print("==", kind, sum(sites.values()), "absmax launches per eager step")
    for (shape, where), n in sorted(sites.items(), key=lambda kv: -kv[1]):
        print(f"  {n:3d}  {shape}  {where}")
