#!/usr/bin/env python3
"""One replayed training step of a rocprofv3 kernel trace of bench.py in time order: start offset, duration, stream (queue), kernel, grid --
to see which launches a small kernel sits between.   usage: step_timeline.py <trace dir>"""
import csv, glob, re, sys
d = sys.argv[1]
hits = glob.glob(d + "/*/*kernel_trace.csv") + glob.glob(d + "/*kernel_trace.csv")
rows = list(csv.DictReader(open(hits[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adam_multi_kernel")]
# the replayed benchmark steps: Adam .. Adam segments of >= 100 kernels from the trace's middle (the bench's other arms follow); one of each kind
segs = [rows[a + 1:b + 1] for a, b in zip(adam[:-1], adam[1:]) if b - a >= 100]
segs = segs[len(segs) // 3: 2 * len(segs) // 3]
seen = {}
for sg in segs:
    seen.setdefault(len(sg), sg)
for n, seg in sorted(seen.items()):
  print("==== a segment of %d kernels (Adam to Adam)" % n)
  t0 = int(seg[0]["Start_Timestamp"])
  for r in seg:
    k = re.sub(r"\(.*", "", re.sub(r"^void ", "", r["Kernel_Name"]))[:64]
    if True:
      print("%9.1f us  %7.1f us  q%-3s %-64s grid %sx%sx%s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
          r.get("Queue_Id", "?"), k, r.get("Grid_Size_X", "?"), r.get("Grid_Size_Y", "?"), r.get("Grid_Size_Z", "?")))
