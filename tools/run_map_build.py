"""Builds the configs[2] map index N times from a device-resident cloud (for rocprofv3)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lvio_fusion_amd import api, synthetic as syn
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
which = sys.argv[2] if len(sys.argv) > 2 else "config3"
ctx = api.Context(0)
if which == "config3":
    c3 = syn.config3_icp(); pts = c3["map"]; thr = c3["thr_ground"]
else:
    cand = syn.config5_candidates(1)[0]; print({k: getattr(v, "shape", v) for k, v in cand.items()}); pts = [v for v in cand.values() if getattr(v, "ndim", 0) == 2 and v.shape[0] > 1000][0]; thr = 4.0
cloud = api.Cloud(ctx, np.ascontiguousarray(pts[:, :3], np.float32))
m = api.Map(ctx, cloud, thr); m.close()
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    m = api.Map(ctx, cloud, thr); m.close()
ctx.synchronize()
print("points", len(pts), "ms per build", 1e3 * (time.perf_counter() - t0) / n)
