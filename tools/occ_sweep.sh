#!/bin/bash
# Runs ON THE GPU BOX: the scan-match frame and the loop-closure candidates under different grid-pyramid occupancy targets
for occ in 16 32 64 128 256; do
  echo "occ $occ: $(LVF_KNN_OCC=$occ timeout 120 python tools/run_scan_match.py 30 2>&1 | grep -o "'ms_per_frame': [0-9.]*")  $(LVF_KNN_OCC=$occ timeout 120 python tools/knn_time.py 20 2>&1 | grep ms | tr '\n' ' ')"
done
