#!/bin/bash
# Profiles of the benchmark for round $ROUND (default 5; GPU box): kernel trace with per-kernel roofline columns, the PMC passes the guide prescribes
# (FETCH_SIZE and WRITE_SIZE in separate passes; SQ counters in a third), and the adversarial cycle's kernel trace.
# Output: gpurun_out/r$ROUND/profiles/*.txt -- tools/install_profiles.py copies what is to be judged into profiles/.
set -u
export TMPDIR=/tmp
ROUND=${ROUND:-5}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r$ROUND/profiles
mkdir -p $O
B="python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-ge2e --no-adversarial --no-fp32 --no-roofline --no-stock"
run() { tag=$1; shift; rm -rf /tmp/p_$tag; (cd /tmp && timeout -k 10 400 rocprofv3 "$@" --output-format csv -d /tmp/p_$tag -- $B > $O/$tag.json 2> $O/$tag.err) || echo "$tag: rc=$?"; }
rm -f /tmp/shapes.tsv
SSV_SHAPE_LOG=/tmp/shapes.tsv run trace --kernel-trace --stats
run fetch --kernel-trace --pmc FETCH_SIZE
run write --kernel-trace --pmc WRITE_SIZE
run sq --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAVES
cp /tmp/shapes.tsv $O/shapes.tsv 2>/dev/null
python3 $R/tools/summarize_prof.py /tmp/p_trace /tmp/p_fetch /tmp/p_write --shapes /tmp/shapes.tsv --sq /tmp/p_sq > $O/bench_f16x2.txt 2> $O/summarize.err
python3 $R/tools/step_kernels.py /tmp/p_trace > $O/step_kernels.txt 2>> $O/summarize.err
# the adversarial cycle (config 3)
rm -rf /tmp/p_adv
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_adv -- python3 $R/tools/bench_adversarial.py > $O/adv.json 2> $O/adv.err) || echo "adv rc=$?"
python3 $R/tools/summarize_prof.py /tmp/p_adv > $O/adversarial.txt 2>> $O/summarize.err
head -30 $O/bench_f16x2.txt
# the headline kernel alone, for roofline.traffic
for c in FETCH_SIZE WRITE_SIZE; do rm -rf /tmp/p_h$c; (cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/p_h$c -- python3 $R/tools/pmc_headline.py > /dev/null 2> $O/headline_$c.err) || echo "headline $c rc=$?"; done
python3 $R/tools/pmc_headline.py --json /tmp/p_hFETCH_SIZE /tmp/p_hWRITE_SIZE profiles/round${ROUND}_bench_kernel_stats.txt > $O/traffic_f16x2.json 2>> $O/summarize.err
# the headline kernel's counter rows themselves (one line per launch), so the figure can be recomputed from a committed file
for c in FETCH_SIZE WRITE_SIZE; do f=$(ls /tmp/p_h$c/*/*counter_collection.csv /tmp/p_h$c/*counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && (head -1 "$f"; grep "gemm_nn_bf3_kernel<3, 1, 7" "$f") > $O/headline_$c.csv; done
cat $O/traffic_f16x2.json
