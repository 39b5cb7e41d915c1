#!/bin/bash
# same-box A/B of two builds of the library: tools/ab_lib.sh <python script + args>   (old = lvio_fusion_amd/liblvf_hip_old.so)
for k in 1 2 3; do
  echo "old: $(LVF_LIB_PATH=$PWD/lvio_fusion_amd/liblvf_hip_old.so timeout 120 python "$@" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr\|amdgpu.ids" | tr '\n' ' ')"
  echo "new: $(timeout 120 python "$@" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr\|amdgpu.ids" | tr '\n' ' ')"
done
