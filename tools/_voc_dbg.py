import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from spoofsv_amd import ops, resident, _lib
from spoofsv_amd.vocoder import Vocoder, _p
from oracle import vocoder_oracle as vo
torch.manual_seed(0)
for (Co, Ci) in ((1024, 1026), (1026, 1024), (1024, 1024), (513, 1026), (1024, 513), (1024, 1056), (1024, 1040)):
    x = torch.randn(2, Ci, 25, device="cuda"); w = torch.randn(Co, Ci, 1, device="cuda")
    y = torch.empty(2, Co, 25, device="cuda")
    ops._conv_fwd(x, Ci * 25, w, None, None, y, Co * 25, 1, 1, 0)
    ref = torch.einsum("oc,bct->bot", w[:, :, 0].double(), x.double()).float()
    e1 = float((y - ref).abs().max() / ref.abs().max())
    rw = resident.ResidentWeights([w]); rw.refresh(ops._stream())
    ops._conv_fwd(x, Ci * 25, w, None, None, y, Co * 25, 1, 1, 0)
    print(Co, Ci, "plain", e1, "resident", float((y - ref).abs().max() / ref.abs().max()))
v = Vocoder(1024, 256)
T = 25
fr = torch.randn(2, 1024, T, device="cuda")
y = torch.empty(2, 256 * (T - 1), device="cuda")
env = v._inv_env(T)
_lib.call("ssv_ola_signal", _p(fr), _p(env), _p(y), 2, 1024, T, 256, ops._stream())
f = fr.cpu().numpy().astype(np.float64); e = env.cpu().numpy().astype(np.float64)
o = np.zeros((2, 1024 + 256 * (T - 1)))
for t in range(T): o[:, t * 256:t * 256 + 1024] += f[:, :, t]
o = (o * e)[:, 512:-512]
print("ola_signal", np.abs(y.cpu().numpy() - o).max())
