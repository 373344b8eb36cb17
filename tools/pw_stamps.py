#!/usr/bin/env python3
"""Diagnostic (GPU box, library built with -DSSV_PW_STAMP): where a workgroup of the fused 1x1 conv + LayerNorm kernel (gemm_pwln_kernel)
spends its cycles -- prologue | chunk loop | pre stores + column sums | variance | y stores -- and when the workgroups of a launch enter and leave."""
import ctypes, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spoofsv_amd import ops, _lib
B = 32
for (Cin, Cout, L, act) in ((513, 513, 1300, 1), (256, 512, 1300, 0), (256, 256, 325, 1)):
    x = torch.randn(B, Cin, L, device="cuda")
    w = torch.randn(Cout, Cin, 1, device="cuda") * 0.05
    bias = torch.randn(Cout, device="cuda") * 0.1
    gam = torch.rand(Cout, device="cuda") + 0.5
    bet = torch.randn(Cout, device="cuda") * 0.3
    with torch.no_grad():
        for _ in range(3):
            y = ops.pointwise_conv_ln_act(x, w, bias, gam, bet, None, act)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (1024 * 8))()
    rc = _lib.lib().ssv_debug_pw_stamps(buf)
    nwg = min(1024, B * ((L + 63) // 64))
    rows = [[buf[i * 8 + k] for k in range(8)] for i in range(nwg) if buf[i * 8 + 7]]
    t0 = min(r[6] for r in rows)
    print("%d -> %d, L = %d (rc %d, %d workgroups of %d stamped)" % (Cin, Cout, L, rc, len(rows), B * ((L + 63) // 64)))
    rows.sort(key=lambda r: r[6])
    # rounds: workgroups grouped by entry time (a new round starts when an entry is more than 3 us after the previous one)
    groups, cur = [], [rows[0]]
    for r in rows[1:]:
        if (r[6] - cur[-1][6]) > 300 and len(cur) >= 32: groups.append(cur); cur = [r]
        else: cur.append(r)
    groups.append(cur)
    med = lambda v: statistics.median(v)
    for g in groups:
        ph = [med([r[k + 1] - r[k] for r in g]) for k in range(5)]
        print("  %4d workgroups entering %.1f .. %.1f us: resident %.1f us (median) | cycles: prologue %d | chunk loop %d | pre stores + sums %d | variance %d | y stores %d | clock %.2f GHz" % (
            len(g), (g[0][6] - t0) / 100.0, (g[-1][6] - t0) / 100.0, med([(r[7] - r[6]) / 100.0 for r in g]), ph[0], ph[1], ph[2], ph[3], ph[4],
            med([(r[5] - r[0]) / ((r[7] - r[6]) * 10.0) for r in g if r[7] > r[6]])))
    print("  last exit %.1f us after the first entry" % ((max(r[7] for r in rows) - t0) / 100.0))
