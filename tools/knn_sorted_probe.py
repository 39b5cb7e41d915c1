"""Does spatially sorting the query scan speed up k_knn3?  (probe for a device-side pre-sort at lvf_scan_create)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lvio_fusion_amd import api, synthetic as syn
ctx = api.Context(0)
c3 = syn.config3_icp()
mp = api.Map(ctx, c3["map"], c3["thr_ground"])
def run(q, label):
    sc = api.Scan(ctx, q)
    for thr in (c3["thr_ground"], c3["thr_surf"]):
        for _ in range(3): api.knn3(mp, sc, c3["pose0"], thr)
        ctx.synchronize(); ctx.timer_begin()
        for _ in range(20): api.knn3(mp, sc, c3["pose0"], thr)
        ctx.timer_end(); print(label, thr, "ms", ctx.timer_ms() / 20)
    sc.close()
q = c3["query"]
run(q, "scan order")
for cell in (0.25, 0.5, 1.0, 2.0):
    key = np.floor(q[:, :3] / cell).astype(np.int64); key -= key.min(0)
    d = key.max(0) + 1
    order = np.argsort((key[:, 2] * d[1] + key[:, 1]) * d[0] + key[:, 0], kind="stable")
    run(q[order], f"cell-sorted {cell}")
rng = np.random.default_rng(0)
run(q[rng.permutation(len(q))], "shuffled")
