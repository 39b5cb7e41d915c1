"""One scan-match frame (2 x {associate, <= 4 LM iterations}) on a voxel-filtered scan of realistic size (a few thousand points):
the regime of the reference's live Mapping::Optimize, where every launch sits on its latency floor."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lvio_fusion_amd import api, synthetic as syn
ctx = api.Context(0)
c3 = syn.config3_icp()
step = int(sys.argv[1]) if len(sys.argv) > 1 else 20          # keep every step-th query point
qg = c3["query"][c3["query_ground"]][::step]; qs = c3["query"][~c3["query_ground"]][::step]
mg = c3["map"][c3["map_ground"]]; ms = c3["map"][~c3["map_ground"]]
opt = api.scan_match_options(0.2, outer_iterations=1, prior_weight=0.0)
mpg, mps = api.Map(ctx, mg, opt.thr_ground), api.Map(ctx, ms, opt.thr_surf)
scg, scs = api.Scan(ctx, qg), api.Scan(ctx, qs)
for _ in range(3):
    api.scan_match(mpg, scg, mps, scs, c3["map_pose"], c3["pose0"], opt)
ctx.synchronize()
reps = 50
t0 = time.perf_counter()
for _ in range(reps):
    r = api.scan_match(mpg, scg, mps, scs, c3["map_pose"], c3["pose0"], opt)
dt = (time.perf_counter() - t0) / reps
print("scan-match frame, %d ground + %d surf points: %.3f ms" % (len(qg), len(qs), 1e3 * dt))
