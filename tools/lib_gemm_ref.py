#!/usr/bin/env python3
"""Reference point (GPU box): what the vendor library (hipBLASLt through torch.matmul) reaches on plain bf16 GEMMs of the hot
shapes -- one bf16 product, operands already bf16 and resident.  The split-bf16 path needs three such products per fp32 product
plus the split itself, so (library time x 3) is what a library-based implementation of the same arithmetic would cost."""
import torch
dev = "cuda:0"
def t(fn, n=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, (M, K, N, B) in {"conv fwd C=256->512 L=325 k=3 as GEMM (B=32 batched)": (512, 768, 325, 32),
                           "conv fwd C=512->1024 L=1300 k=3 (B=32 batched)": (1024, 1536, 1300, 32),
                           "LSTM step 3072 x 880 x 1536": (3072, 1536, 880, 1),
                           "LSTM step, 2 layers batched": (3072, 1536, 880, 2)}.items():
    a = torch.randn(B, M, K, device=dev).bfloat16(); x = torch.randn(B, K, N, device=dev).bfloat16()
    o = torch.empty(B, M, N, device=dev, dtype=torch.bfloat16)
    us = t(lambda: torch.bmm(a, x, out=o))
    fl = 2.0 * M * K * N * B
    print("%-55s %8.1f us  %7.1f TFLOP/s (one bf16 product)   x3 = %8.1f us -> %6.1f algorithmic TFLOP/s" % (name, us, fl / us / 1e6, 3 * us, fl / (3 * us) / 1e6), flush=True)
