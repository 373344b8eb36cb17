#!/usr/bin/env python3
"""Scan a hipcc --save-temps gfx950 .s file for GEMM loops whose wait counts have collapsed: an `s_waitcnt vmcnt(0 | 1)` between a barrier and the
first MFMAs behind it while loads were issued shortly before that barrier -- the wave then waits for the prefetch it has just issued (one exposed
memory round trip per chunk) instead of only for the operands of this chunk.  hipcc's wait bookkeeping merges over all paths: a prefetch written in two
forms under a test, or a load inside a waterfall loop, is what usually causes it.
usage: hipcc -O3 --offload-arch=gfx950 --save-temps -c x.hip ; python tools/isa_loop_waits.py x-hip-amdgcn-amd-amdhsa-gfx950.s [name filter]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)s_endpgm", txt, re.S | re.M):
    name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void ", "")
    if flt not in name:
        continue
    L = [l.strip() for l in m.group(2).split("\n")]
    is_load = lambda t: re.match(r"(buffer|global|flat)_load", t) is not None
    hits = []
    # loop headers: labels that a LATER branch jumps back to; the loads "before" a header are those in front of that back edge
    labels = {t[:-1].split(":")[0]: i for i, t in enumerate(L) if re.match(r"\.LBB\d+_\d+:", t)}
    back = {}
    for i, t in enumerate(L):
        b = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", t)
        if b and b.group(1) in labels and labels[b.group(1)] < i:
            back[labels[b.group(1)]] = i
    for i, t in enumerate(L):
        header = i in back
        if not (t.startswith("s_barrier") or header):
            continue
        src = back[i] if header else i
        loads_before = sum(1 for q in L[max(0, src - 120):src] if is_load(q))
        # the first vmcnt wait within the next 40 instructions, if MFMAs follow within 60
        mf = [j for j in range(i + 1, min(len(L), i + 60)) if L[j].startswith("v_mfma")]
        if not mf:
            continue
        for j in range(i + 1, min(len(L), mf[0] + 4)):
            w = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", L[j])
            if w:
                if int(w.group(1)) <= 1 and loads_before >= 8:
                    hits.append((i, int(w.group(1)), loads_before, "loop header" if header else "barrier"))
                break
    if hits:
        print("%-52s %d place(s) where MFMAs wait on vmcnt(<=1) with a prefetch in flight: %s" % (name[:52], len(hits), ", ".join("%s at line %d (%d loads before)" % (h[3], h[0], h[2]) for h in hits[:4])))
