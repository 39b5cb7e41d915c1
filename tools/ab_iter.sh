#!/bin/bash
# same-box A/B of the configs[3] iteration (old = lvio_fusion_amd/liblvf_hip_old.so)
for k in 1 2 3; do
  echo "old: $(LVF_LIB_PATH=$PWD/lvio_fusion_amd/liblvf_hip_old.so timeout 200 python tools/run_batch.py 20 1,8 2>&1 | grep -o "device loop, 1 window: [0-9.]* ms/it\|batch of 8: tables=1 [0-9.]* ms" | tr '\n' ' ')"
  echo "new: $(timeout 200 python tools/run_batch.py 20 1,8 2>&1 | grep -o "device loop, 1 window: [0-9.]* ms/it\|batch of 8: tables=1 [0-9.]* ms" | tr '\n' ' ')"
done
LVF_CHOL_TIMING=1 timeout 100 python tools/one_iteration.py 2>&1 | grep "chol step" | tail -4
