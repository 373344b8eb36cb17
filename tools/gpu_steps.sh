#!/bin/bash
# Run GPU steps one after the other on the box, each under its own timeout, logs under gpurun_out/<tag>/.
# A failing test does not stop the sequence; a step that had to be KILLED (timeout) does: no further GPU step after a hang.
#   tools/gpu_steps.sh <tag> "<seconds> <name> <command...>" ...
tag=$1; shift
out=gpurun_out/$tag
mkdir -p "$out"
for spec in "$@"; do
  secs=${spec%% *}; rest=${spec#* }; name=${rest%% *}; cmd=${rest#* }
  echo "== $name (limit ${secs}s): $cmd" | tee -a "$out/steps.log"
  t0=$(date +%s)
  timeout -k 10 "$secs" bash -c "$cmd" > "$out/$name.log" 2>&1
  rc=$?
  echo "== $name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a "$out/steps.log"
  tail -n 4 "$out/$name.log"
  if grep -q "Memory access fault" "$out/$name.log"; then echo "== $name: GPU memory access fault: stopping" | tee -a "$out/steps.log"; exit 1; fi
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "== $name was killed at its limit: stopping" | tee -a "$out/steps.log"; exit 1; fi
done
exit 0
