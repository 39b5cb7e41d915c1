"""Runs the configs[2] scan-match frame N times (for rocprofv3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from lvio_fusion_amd import api, synthetic as syn
ctx = api.Context(0)
c3 = syn.config3_icp()
print(bench.scan_match_frame(api, syn, ctx, c3, reps=int(sys.argv[1]) if len(sys.argv) > 1 else 20))
