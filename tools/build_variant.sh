#!/bin/bash
# Build a tuning variant of libssv_hip.so with extra compiler flags:  tools/build_variant.sh NAME -DSSV_XYZ=1 ...
# -> spoofsv_amd/csrc/build/ab/libssv_hip_NAME.so (git-ignored; travels to the GPU box).  Use with SSV_HIP_LIB=<path>.
set -e
name=$1; shift
cd "$(dirname "$0")/../spoofsv_amd/csrc"
mkdir -p build/ab/$name
for f in api gemm_nn gemm_nt pack conv_nn wgrad_nt wgrad_nt3r pwln norm norm_pers attn attn_fused misc lstm vocoder synth critic; do
  extra=""; [ $f = norm_pers ] && extra="-fno-slp-vectorize"
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $extra "$@" -c $f.hip -o build/ab/$name/$f.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 build/ab/$name/*.o -o build/ab/libssv_hip_$name.so
echo built build/ab/libssv_hip_$name.so
