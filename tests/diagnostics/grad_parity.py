#!/usr/bin/env python3
"""Diagnostic (GPU box): per-parameter gradient error of the full-size training step vs the CPU oracle, both arithmetic modes
(the numbers behind tests/test_gpu_parity.py::test_bench_workload_full_size_training_step_vs_oracle)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import spoofsv_amd
from _golden import rel_err, rel_l2
import test_gpu_parity as T
from spoofsv_amd import ops, train
B = int(os.environ.get("GP_B", "8"))
for kind in ("text2mel", "ssrn"):
    o = T._bench_workload_oracle(kind, B)
    for prec in ("bf16x3", "fp32"):
        spoofsv_amd.set_precision(prec)
        m = o["model"].to("cuda:0").train()
        for p in m.parameters(): p.grad = None
        if kind == "text2mel":
            mel, text, spk = [b.to("cuda:0") for b in o["batch"]]
            Y, A = m(train.shift_right(mel), text, spk)
            l = train.text2mel_losses(Y, A, mel, o["gaw"].to("cuda:0"))
            print(kind, prec, "A: max %.2e l2 %.2e" % (rel_err(A, o["outs"]["A"]), rel_l2(A, o["outs"]["A"])))
        else:
            mel, lin = [b.to("cuda:0") for b in o["batch"]]
            Y = m(mel); l = ops.spec_losses(Y, lin)
        print(kind, prec, "Y: max %.2e l2 %.2e" % (rel_err(Y, o["outs"]["Y"]), rel_l2(Y, o["outs"]["Y"])), "losses", [abs(float(a) - b) for a, b in zip(l, o["losses"])])
        sum(l).backward(); torch.cuda.synchronize()
        errs = sorted(((rel_l2(p.grad, o["grads"][k]), rel_err(p.grad, o["grads"][k]), k) for k, p in m.named_parameters()), reverse=True)
        print(kind, prec, "vs float32 oracle, worst 8 (l2, max):", ["%s %.2e %.2e" % (k, a, b) for a, b, k in errs[:8]])
        print(kind, prec, "median l2 %.2e, n>1e-3: %d of %d, n>5e-4: %d" % (errs[len(errs) // 2][0], sum(e[0] > 1e-3 for e in errs), len(errs), sum(e[0] > 5e-4 for e in errs)), flush=True)
        e64 = sorted(((rel_l2(p.grad, o["grads64"][k]), rel_l2(o["grads"][k], o["grads64"][k]), k) for k, p in m.named_parameters()), reverse=True)
        print(kind, prec, "vs float64 oracle (hip, float32-oracle), worst 6:", ["%s %.2e %.2e" % (k, a, b) for a, b, k in e64[:6]],
              "median hip %.2e ref %.2e" % (e64[len(e64) // 2][0], sorted(e[1] for e in e64)[len(e64) // 2]), flush=True)
        m.cpu()
