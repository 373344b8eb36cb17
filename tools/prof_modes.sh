set -e
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in f16x2 bf16x3; do
  rm -rf /tmp/prof_$mode
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$mode -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-ge2e --no-adversarial --no-fp32 --no-roofline --no-stock --precision $mode > $R/gpurun_out/r3/prof_$mode.json 2> $R/gpurun_out/r3/prof_$mode.err)
  python3 tools/summarize_prof.py /tmp/prof_$mode > gpurun_out/r3/prof_$mode.txt
done
head -45 gpurun_out/r3/prof_f16x2.txt
