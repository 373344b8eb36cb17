#!/bin/bash
# per-kernel table of config 5's forward (tools/prof_ge2e.py: 4 forwards of 880 x 120 x 40 on fixed weights) -> gpurun_out/ge2e/fwd_kernel_stats.csv
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/ge2e
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pgf && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pgf -- python3 $R/tools/prof_ge2e.py > /tmp/pgf.log 2>&1
cp $(ls /tmp/pgf/*/*kernel_stats.csv | head -1) $R/gpurun_out/ge2e/fwd_kernel_stats.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$R/gpurun_out/ge2e/fwd_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("== forward: %.2f ms of kernels per call (4 calls traced)" % (tot/4e6))
for r in rows[:14]:
    print("%8.1f us x %6.1f = %7.2f ms/call  %s"%(float(r['AverageNs'])/1e3,int(r['Calls'])/4,float(r['TotalDurationNs'])/4e6,r['Name'][:80]))
PY
