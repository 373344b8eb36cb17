#!/bin/bash
# In-step effect of forcing the conv tile of one problem shape (SSV_NNB_FORCE="kt:M:N=wm,nt"): ms per step of bench.py per setting.
# usage (GPU box): tools/sweep_force.sh "3:512:325=1,11" "3:512:325=1,12;3:256:325=1,12" ...   ("" = the cost model's choice)
mkdir -p gpurun_out/r3
for f in "" "$@" ""; do
  SSV_NNB_FORCE="$f" timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-ge2e --no-adversarial --no-fp32 --no-roofline 2>/dev/null | \
    python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-44s step %.3f  text2mel %.3f  ssrn %.3f' % ('$f' or '(default)', d['ms_per_step'], d['config']['text2mel_ms'], d['config']['ssrn_ms']))"
done
