#!/bin/bash
# per-kernel table of the GE2E training iteration (tools/ge2e_train_time.py under rocprofv3 --kernel-trace --stats) -> gpurun_out/ge2e/train_kernel_stats_$1.csv
tag=${1:-tree}; lib=spoofsv_amd/libssv_hip.so; [ $tag != tree ] && lib=spoofsv_amd/csrc/build/ab/libssv_hip_$tag.so
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/ge2e
export SSV_HIP_LIB=$R/$lib
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pg_$tag && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pg_$tag -- python3 $R/tools/ge2e_train_time.py 3 > /tmp/pg_$tag.log 2>&1
cp $(ls /tmp/pg_$tag/*/*kernel_stats.csv | head -1) $R/gpurun_out/ge2e/train_kernel_stats_$tag.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$R/gpurun_out/ge2e/train_kernel_stats_$tag.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("== $tag: %.2f ms of kernels per iteration" % (tot/4e6))
for r in rows[:16]:
    print("%8.1f us x %5d = %7.2f ms/iter  %s"%(float(r['AverageNs'])/1e3,int(r['Calls'])//4,float(r['TotalDurationNs'])/4e6,r['Name'][:80]))
PY
