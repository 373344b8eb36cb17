#!/bin/bash
# config 5 (python bench.py --ge2e: forward + loss on fixed weights, with the weight split, one training iteration, distance from the oracle) under several
# library builds, one box:  tools/ge2e_fwd_ab.sh NAME ...   (NAME = a build under spoofsv_amd/csrc/build/ab, or "tree")
mkdir -p gpurun_out/ge2e
for n in "$@"; do
  lib=spoofsv_amd/csrc/build/ab/libssv_hip_$n.so; [ $n = tree ] && lib=spoofsv_amd/libssv_hip.so
  SSV_HIP_LIB=$lib timeout -k 10 200 python bench.py --ge2e 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$n: %.2f ms  %.0f utt/s  (%.2f with the weight split)  train %.2f ms  err %.3e  loss %.4f' % (d['ms'], d['value'], d['ms_with_weight_split'], d['train_iteration']['ms'], d['rel_err_vs_cpu_oracle'], d['loss']))"
done | tee -a gpurun_out/ge2e/fwd_ab.txt
