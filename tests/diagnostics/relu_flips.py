#!/usr/bin/env python3
"""Diagnostic (GPU box): does an arithmetic mode put any activation on the other side of a ReLU than the exact-fp32 mode does?
Full-size Text2Mel at B = 8 (the workload of test_bench_workload_full_size_training_step_vs_oracle): for every ReLU layer, the
number of elements whose sign differs from the fp32 mode's, and the magnitude of the pre-activation there."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import spoofsv_amd
from spoofsv_amd import ops, train
from spoofsv_amd.tts import melSyn

torch.manual_seed(1234)
m = melSyn(34, True, 200, textemb_dim=128, freq_bins=80, hidden_dim=256)
m.apply(train.init_weights)
m = m.cuda().train()
mel, text, spk = [b.cuda() for b in train.synthetic_text2mel_batch(8, 186, 325, seed=0)]
rec = {}
orig = ops.pointwise_conv_ln_act
def spy(x, w, bias, gamma, beta, s=None, act=0):
    y = orig(x, w, bias, gamma, beta, s, act)
    if act == 1:
        rec.setdefault(cur[0], []).append(y.detach().clone())
    return y
ops.pointwise_conv_ln_act = spy
import spoofsv_amd.tts as tts
cur = [None]
for prec in ("fp32", "f16x2", "bf16x3"):
    spoofsv_amd.set_precision(prec)
    cur[0] = prec
    with torch.no_grad():
        m(train.shift_right(mel), text, spk)
    torch.cuda.synchronize()
ref = rec["fp32"]
for prec in ("f16x2", "bf16x3"):
    for i, (a, b) in enumerate(zip(rec[prec], ref)):
        flip = (a > 0) != (b > 0)
        n = int(flip.sum())
        mag = float(torch.maximum(a, b)[flip].max()) if n else 0.0
        print("%-6s relu layer %d %s: %d of %d signs differ from fp32 (largest value there %.2e; layer rms %.2e)" % (
            prec, i, tuple(a.shape), n, a.numel(), mag, float(b.pow(2).mean().sqrt())), flush=True)
