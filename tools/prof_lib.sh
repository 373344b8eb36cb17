#!/bin/bash
# In-step kernel profile of one library build:  tools/prof_lib.sh TAG [LIB.so] [MODE]   (GPU box; writes gpurun_out/r3/prof_TAG.txt)
# The reliable way to judge a kernel variant: the rocprofv3 per-kernel averages of the replayed training step.
tag=$1; lib=${2:-}; mode=${3:-f16x2}
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
[ -n "$lib" ] && export SSV_HIP_LIB=$R/$lib
rm -rf /tmp/prof_$tag
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-ge2e --no-adversarial --no-fp32 --no-roofline --no-stock --precision $mode > $R/gpurun_out/r3/prof_$tag.json 2> $R/gpurun_out/r3/prof_$tag.err)
python3 tools/summarize_prof.py /tmp/prof_$tag > gpurun_out/r3/prof_$tag.txt
python3 tools/step_kernels.py /tmp/prof_$tag > gpurun_out/r3/steps_$tag.txt 2>&1
tail -1 gpurun_out/r3/prof_$tag.txt | cut -c1-120
