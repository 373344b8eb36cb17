#!/usr/bin/env python3
"""WGAN-GP cycle timing (1 G + 5 D iterations, hipGraph replay) for profiling: python tools/bench_adversarial.py [text2mel|ssrn] [batch]"""
import sys
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import contextlib
import os
import torch
import bench
if os.environ.get("AB_FULL_GRADS") == "1":     # A/B: the gradient penalty's first pass with every parameter gradient computed (the round-3 behaviour)
    from spoofsv_amd import ops
    ops.input_grads_only = lambda module: contextlib.nullcontext()

kind = sys.argv[1] if len(sys.argv) > 1 else "ssrn"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
ms = bench.adversarial_cycle_ms(kind, batch, torch.device("cuda", 0), cycles=3)
print("%s adversarial: %.3f ms per iteration (cycle of 6 = %.2f ms)" % (kind, ms, 6 * ms), flush=True)
