#!/usr/bin/env python3
"""Per replayed training step (Adam to Adam+2 in a rocprofv3 kernel trace): every kernel name with its launches and microseconds per step,
small ones first -- what the launches that are too short to show in a time-ranked table add up to.   usage: step_kernels.py <trace dir>"""
import collections, csv, glob, re, sys
d = sys.argv[1]
hits = glob.glob(d + "/*/*kernel_trace.csv") + glob.glob(d + "/*kernel_trace.csv")
rows = [(float(r["Start_Timestamp"]), float(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size_X", r.get("Grid_Size", "?"))) for r in csv.DictReader(open(hits[0]))]
rows.sort()
adam = [i for i, r in enumerate(rows) if r[2].startswith("adam_multi_kernel")]
steps = []
for a, b in zip(adam[:-2:2], adam[2::2]):
    seg = rows[a + 1:b + 1]
    if len(seg) >= 200:
        steps.append(seg)
steps = steps[len(steps) // 2:]            # the replayed ones
med = sorted(len(sg) for sg in steps)[len(steps) // 2]
steps = [sg for sg in steps if len(sg) == med]      # whole steps only (a pair of Adam launches can also straddle two phases of the benchmark)
n = len(steps)
cnt, tot, torch_k = collections.Counter(), collections.Counter(), collections.Counter()
for seg in steps:
    for s_, e_, k, gsz in seg:
        if k.startswith('void at::native') or 'rocclr' in k or k.startswith('at::native'): torch_k[(re.sub(r'\(.*', '', k)[:90], gsz)] += 1
        k = re.sub(r"\(.*", "", re.sub(r"^void ", "", k))[:70]
        cnt[k] += 1; tot[k] += (e_ - s_) / 1e3
print("%d replayed steps; per step: %.0f kernels, %.1f us of kernel time" % (n, sum(cnt.values()) / n, sum(tot.values()) / n))
small = 0.0
for k in sorted(cnt, key=lambda k: tot[k] / cnt[k]):
    avg = tot[k] / cnt[k]
    if avg < 12.0: small += tot[k] / n
    print("%7.1f us avg  %5.1f launches/step  %8.1f us/step  %s" % (avg, cnt[k] / n, tot[k] / n, k))
print("kernels under 12 us on average: %.1f us per step" % small)
print("torch / runtime kernels per step, by grid size (threads):")
for (k, gsz), c in sorted(torch_k.items(), key=lambda kv: -kv[1]):
    print("  %5.1f /step  grid %-10s %s" % (c / n, gsz, k))
