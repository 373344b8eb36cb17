"""Which operands have no producer-written scale list (each costs an ssv_absmax launch)?  GPU box.
   python tools/exp_absmax_sites.py [step|adv]  -- one Text2Mel + SSRN step, or the WGAN-GP iterations of SSRN + linDisc (counted while they are captured)."""
import collections, sys, traceback
import torch
sys.path.insert(0, ".")
import bench
from spoofsv_amd import ops, train

sites = collections.Counter()
orig = ops.amax_of
def spy(x):
    h = getattr(x, "_ssv_amax", None)
    if not (h is not None and h[1] == x._version and h[0].shape[0] == x.shape[0]):
        fr = [f for f in traceback.extract_stack()[:-1] if "spoofsv_amd" in f.filename][-4:]
        sites[(tuple(x.shape), " < ".join(f"{f.name}:{f.lineno}" for f in reversed(fr)))] += 1
    return orig(x)
ops.amax_of = spy
dev = torch.device("cuda:0")
def report(title):
    print("==", title, sum(sites.values()), "absmax launches from ops.amax_of")
    for (shape, where), n in sorted(sites.items(), key=lambda kv: -kv[1]):
        print(f"  {n:3d}  {shape}  {where}")
    sites.clear()
if (sys.argv[1] if len(sys.argv) > 1 else "step") == "step":
    for kind in ("text2mel", "ssrn"):
        t = bench.Trainer(kind, 32, dev, 0, 1, use_graph=False)
        t.prepare()
        t.step(); torch.cuda.synchronize()
        sites.clear()
        t.step(); torch.cuda.synchronize()
        report(kind + " eager step")
else:
    from spoofsv_amd.critic import linDisc
    from spoofsv_amd.tts import SSRN
    torch.manual_seed(1234)
    model, disc = SSRN(80, 513, 256), linDisc(513, 128)
    data = train.synthetic_ssrn_batch(32, 325, seed=0, device=dev)
    model.apply(train.init_weights); disc.apply(train.init_weights)
    model.to(dev).train(); disc.to(dev).train()
    og = train.FusedAdam(model.parameters(), 2e-4, (0.5, 0.9), 1e-6, capturable=True)
    od = train.FusedAdam(disc.parameters(), 2e-4, (0.5, 0.9), 1e-6, capturable=True)
    stepper = train.AdversarialGraphStep("ssrn", model, disc, og, od, data, None, 10.0, None, None)
    stepper.g_step(); torch.cuda.synchronize(); report("first g_step (eager warm-up + capture)")
    stepper.d_step(); torch.cuda.synchronize(); report("first d_step (eager warm-up + capture)")
