#!/bin/bash
# In-step kernel profile of bench.py under extra environment settings:  tools/prof_env.sh TAG [VAR=value ...]   (GPU box)
# -> gpurun_out/${SSV_PROF_DIR:-r5}/prof_TAG.txt (tools/summarize_prof.py table) and prof_TAG.json (the bench line of the profiled run).
# The reliable judge of a kernel variant: per-kernel rocprofv3 averages inside the replayed training step.
tag=$1; shift
mkdir -p gpurun_out/${SSV_PROF_DIR:-r5}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for kv in "$@"; do export "$kv"; done
rm -rf /tmp/prof_$tag /tmp/shapes_$tag.tsv
export SSV_SHAPE_LOG=/tmp/shapes_$tag.tsv
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-ge2e --no-adversarial --no-fp32 --no-roofline --no-stock > $R/gpurun_out/${SSV_PROF_DIR:-r5}/prof_$tag.json 2> $R/gpurun_out/${SSV_PROF_DIR:-r5}/prof_$tag.err)
python3 tools/summarize_prof.py /tmp/prof_$tag --shapes /tmp/shapes_$tag.tsv > gpurun_out/${SSV_PROF_DIR:-r5}/prof_$tag.txt
grep -m1 "median replayed step" gpurun_out/${SSV_PROF_DIR:-r5}/prof_$tag.txt | cut -c1-160
