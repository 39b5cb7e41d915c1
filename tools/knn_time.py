import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lvio_fusion_amd import api, synthetic as syn
import time
ctx = api.Context(0); c3 = syn.config3_icp()
for thr in (c3["thr_ground"], c3["thr_surf"]):
    t0=time.perf_counter(); mp = api.Map(ctx, c3["map"], thr); ctx.synchronize(); tb=time.perf_counter()-t0
    sc = api.Scan(ctx, c3["query"])
    api.knn3(mp, sc, c3["pose0"], thr); ctx.synchronize()
    ctx.timer_begin()
    for _ in range(20): api.knn3(mp, sc, c3["pose0"], thr)
    ctx.timer_end(); ms = ctx.timer_ms()/20
    print("thr %.1f ms %.4f Mpairs/s %.0f  (map build %.2f ms)" % (thr, ms, 100000/ms/1e3, tb*1e3))
