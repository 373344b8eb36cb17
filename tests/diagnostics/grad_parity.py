#!/usr/bin/env python3
"""Diagnostic (GPU box): per-parameter gradient error of the full-size training step vs the float64 CPU oracle evaluated on the
HIP path's own ReLU sides, every arithmetic mode, in model order (the numbers behind
tests/test_gpu_parity.py::test_bench_workload_full_size_training_step_vs_oracle).  GP_KINDS=ssrn GP_MODES=f16x2 narrow it down."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import spoofsv_amd
from _golden import rel_err, rel_l2
import test_gpu_parity as T
from spoofsv_amd import ops, train
B = int(os.environ.get("GP_B", "8"))
for kind in os.environ.get("GP_KINDS", "text2mel,ssrn").split(","):
    o = T._bench_workload_oracle(kind, B)
    for prec in os.environ.get("GP_MODES", "f16x2,bf16x3,fp32").split(","):
        spoofsv_amd.set_precision(prec)
        m = o["model"].to("cuda:0").train()
        for p in m.parameters(): p.grad = None
        ops.RELU_TAP = []
        if kind == "text2mel":
            mel, text, spk = [b.to("cuda:0") for b in o["batch"]]
            Y, A = m(train.shift_right(mel), text, spk)
            l = train.text2mel_losses(Y, A, mel, o["gaw"].to("cuda:0"))
            print(kind, prec, "A: max %.2e l2 %.2e" % (rel_err(A, o["outs"]["A"]), rel_l2(A, o["outs"]["A"])))
        else:
            mel, lin = [b.to("cuda:0") for b in o["batch"]]
            Y = m(mel); l = ops.spec_losses(Y, lin)
        sides, ops.RELU_TAP = [t.cpu() for t in ops.RELU_TAP] + [(Y.detach() > (mel if kind == "text2mel" else lin)).cpu()], None
        print(kind, prec, "Y: max %.2e l2 %.2e" % (rel_err(Y, o["outs"]["Y"]), rel_l2(Y, o["outs"]["Y"])), "losses", [abs(float(a.detach()) - b) for a, b in zip(l, o["losses"])])
        sum(l).backward(); torch.cuda.synchronize()
        flips = [int((a != b[0]).sum()) for a, b in zip(sides, o["kinks64"])]
        exact = o["grads64"] if sum(flips) == 0 else T._oracle_pass(o, torch.float64, force=sides)[2]
        print(kind, prec, "kink sides (ReLU layers, then the L1 loss) differing from float64:", flips)
        for k, p in m.named_parameters():
            print("  %-40s on HIP sides %.2e   plain float64 %.2e   float32 oracle vs float64 %.2e" % (
                k, rel_l2(p.grad, exact[k]), rel_l2(p.grad, o["grads64"][k]), rel_l2(o["grads"][k], o["grads64"][k])), flush=True)
        m.cpu()
