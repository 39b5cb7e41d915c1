"""Runs N LM iterations of the config-4 window (for rocprofv3)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from lvio_fusion_amd import api, synthetic as syn
ctx = api.Context(0)
print(json.dumps(bench.full_window(api, syn, ctx, iters=int(sys.argv[1]) if len(sys.argv) > 1 else 20, ids_by_birth="--ids-by-birth" in sys.argv)))
