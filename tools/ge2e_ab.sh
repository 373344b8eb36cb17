#!/bin/bash
# GE2E training iteration under several library builds, one box:  tools/ge2e_ab.sh NAME ...   (NAME = a build under spoofsv_amd/csrc/build/ab, or "tree")
mkdir -p gpurun_out/ge2e
for n in "$@"; do
  lib=spoofsv_amd/csrc/build/ab/libssv_hip_$n.so; [ $n = tree ] && lib=spoofsv_amd/libssv_hip.so
  echo -n "$n: "; SSV_HIP_LIB=$lib timeout -k 10 150 python tools/ge2e_train_time.py 5 2>&1 | grep "train iteration" | cut -c1-150
done | tee -a gpurun_out/ge2e/ab.txt
