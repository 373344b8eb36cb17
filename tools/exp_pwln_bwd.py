"""Fused 1x1 conv + LayerNorm backward (pwln_bwd_kernel) against the two-launch form, same inputs.  GPU box."""
import os, sys, ctypes
import torch
sys.path.insert(0, ".")
from spoofsv_amd import ops, _lib, resident
dev = torch.device("cuda:0")
def run(B, Cin, Cout, L, act, fused):
    os.environ["SSV_PWLN_BWD"] = "1" if fused else "0"
    _lib.lib().ssv_reload_tuning()
    torch.manual_seed(5)
    x = torch.randn(B, Cin, L, device=dev, requires_grad=True)
    w = (torch.randn(Cout, Cin, 1, device=dev) / Cin ** 0.5).requires_grad_()
    bias = torch.randn(Cout, device=dev).requires_grad_()
    gamma = (1 + 0.1 * torch.randn(Cout, device=dev)).requires_grad_()
    beta = (0.1 * torch.randn(Cout, device=dev)).requires_grad_()
    rw = resident.ResidentWeights([w]); rw.refresh(torch.cuda.current_stream().cuda_stream)
    dfr = ops.DeferredWgrad()
    outs = []
    for it in range(2):                       # the second pass runs with the job tables allocated
        for t in (x, w, bias, gamma, beta): t.grad = None
        dfr.begin_step()
        y = ops.pointwise_conv_ln_act(x, w, bias, gamma, beta, None, act)
        gy = torch.randn(B, Cout, L, device=dev, generator=torch.Generator(device=dev).manual_seed(7))
        with dfr:
            y.backward(gy)
        dfr.flush()
        torch.cuda.synchronize()
    return [t.grad.double().cpu() for t in (x, w, bias, gamma, beta)]
def rel(a, b): return float((a - b).norm() / b.norm().clamp_min(1e-30))
for (B, Cin, Cout, L, act) in [(2, 256, 256, 325, 1), (2, 128, 512, 186, 1), (2, 512, 512, 186, 0), (2, 256, 512, 650, 1), (2, 512, 513, 1300, 1), (2, 513, 513, 1300, 1), (2, 513, 513, 1299, 2), (3, 256, 256, 70, 2)]:
    try:
        f = run(B, Cin, Cout, L, act, True); u = run(B, Cin, Cout, L, act, False)
        print(f"{Cin:4d}->{Cout:4d} L={L:5d} act={act}: dx {rel(f[0],u[0]):.2e} dw {rel(f[1],u[1]):.2e} db {rel(f[2],u[2]):.2e} dgamma {rel(f[3],u[3]):.2e} dbeta {rel(f[4],u[4]):.2e}", flush=True)
    except Exception as e:
        print(Cin, Cout, L, "failed:", repr(e)[:300], flush=True)
