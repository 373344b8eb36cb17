#!/usr/bin/env python3
"""Diagnostic: in a rocprofv3 kernel trace (csv) of bench.py, list what runs before / after the tiny torch kernels of ONE step.
usage: trace_neighbours.py <rocprof output dir> [substring ...]"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
pats = sys.argv[2:] or ["FillFunctor", "copyBuffer", "elementwise"]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("embed_bwd_kernel")]
seg = rows[marks[-2]:marks[-1]]                       # one whole step (both models), from the replayed part
print(len(seg), "kernels in the step")
short = lambda r: r["Kernel_Name"].replace("void ", "").replace("at::native::", "")[:46]
cnt = collections.Counter()
for i, r in enumerate(seg):
    if any(p in r["Kernel_Name"] for p in pats):
        cnt[(short(r), short(seg[i - 1]) if i else "", short(seg[i + 1]) if i + 1 < len(seg) else "")] += 1
for k, v in sorted(cnt.items(), key=lambda kv: -kv[1]):
    print("%3d  %-46s after %-46s before %s" % (v, k[0], k[1], k[2]))
