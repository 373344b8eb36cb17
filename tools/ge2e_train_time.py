#!/usr/bin/env python3
"""GE2E config 5 training iteration (880 x 120 x 40: forward keeping every frame, loss, backward, clip, SGD) timed alone, with the loss after three
iterations as a results check between builds.   python tools/ge2e_train_time.py [reps]
Under rocprofv3 --kernel-trace --stats the same command gives the per-kernel table of the iteration."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spoofsv_amd.ge2e import SpeechEmbedder, GE2ELoss
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = "cuda:0"
torch.manual_seed(0)
m = SpeechEmbedder().to(dev).train()
L = GE2ELoss(dev)
x = torch.randn(880, 120, 40).to(dev)
opt = torch.optim.SGD([{"params": m.parameters()}, {"params": L.parameters()}], lr=0.01)


def train_step():
    opt.zero_grad(set_to_none=True)
    ls = L(m(x).reshape(88, 10, 256))
    ls.backward()
    torch.nn.utils.clip_grad_norm_(m.parameters(), 3.0)
    torch.nn.utils.clip_grad_norm_(L.parameters(), 1.0)
    opt.step()
    return ls


l0 = train_step()
g = torch.cat([p.grad.flatten() for p in m.parameters()]).double()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    ls = train_step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print("ge2e train iteration %.2f ms  (first loss %.6f, grad l2 of iteration 0 %.9e, loss after %d more %.6f)  force=%s"
      % (dt * 1e3, float(l0), float(g.norm()), reps, float(ls), os.environ.get("SSV_NNB_FORCE", "")))
