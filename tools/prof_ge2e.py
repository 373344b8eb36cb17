#!/usr/bin/env python3
"""GE2E config 5 forward (880 x 120 x 40) under rocprofv3 --kernel-trace: how much of the wall time of the 122 wavefront steps is kernel time and how
much lies BETWEEN the launches (what a persistent wavefront kernel could remove at best).   rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/prof_ge2e.py
then   python3 tools/prof_ge2e.py --summarize DIR"""
import csv, glob, os, sys
if len(sys.argv) > 2 and sys.argv[1] == "--summarize":
    f = (glob.glob(sys.argv[2] + "/*/*kernel_trace.csv") + glob.glob(sys.argv[2] + "/*kernel_trace.csv"))[0]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
    rows.sort()
    lstm = [r for r in rows if "gemm_nn_bf3_kernel<1, 2, 8, 1" in r[2]]
    # forwards: runs of wavefront launches (one forward = T + layers - 1 = 122 launches); take the last complete one
    n = 122
    last = lstm[-n:]
    dur = [e - s for s, e, _ in last]
    gaps = [last[i + 1][0] - last[i][1] for i in range(n - 1)]
    span = last[-1][1] - last[0][0]
    print("last forward: %d wavefront launches, span %.3f ms; kernel time %.3f ms (avg %.1f us, median %.1f us); between launches %.3f ms (avg %.2f us, max %.1f us)"
          % (n, span / 1e6, sum(dur) / 1e6, sum(dur) / n / 1e3, sorted(dur)[n // 2] / 1e3, sum(gaps) / 1e6, sum(gaps) / len(gaps) / 1e3, max(gaps) / 1e3))
    sys.exit(0)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spoofsv_amd.ge2e import SpeechEmbedder
torch.manual_seed(0)
m = SpeechEmbedder().to("cuda").eval()
x = torch.randn(880, 120, 40, device="cuda")
with torch.no_grad():
    for _ in range(4):
        e = m(x)
torch.cuda.synchronize()
print("ok", float(e.norm()))
