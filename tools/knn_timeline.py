"""Per-wave start / end stamps of the association kernel (STATS variant): how full the chip is over the launch.  tools/knn_timeline.py"""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lvio_fusion_amd import api, _lib, synthetic
ctx = api.Context()
c3 = synthetic.config3_icp()
mp = api.Map(ctx, c3["map"], c3["thr_ground"]); sc = api.Scan(ctx, c3["query"])
BIN = float(os.environ.get("BIN_US", "10"))
Q = len(c3["query"]); pose = np.asarray(c3["pose0"], np.float64)
for rep in range(2):
    stats = np.zeros((Q, 6), np.int32); lv = np.zeros((8, 4), np.float32); nl = C.c_int()
    api._chk(ctx.L.lvf_knn3_debug_stats2(mp.h, sc.h, pose.ctypes.data_as(_lib.c_double_p), 4.0, stats.ctypes.data_as(_lib.c_int_p), 6, lv.ctypes.data_as(_lib.c_float_p), C.byref(nl)))
t0 = stats[:, 4].astype(np.int64); t1 = stats[:, 5].astype(np.int64)
# one wave = 8 consecutive queries
w0 = t0[::8]; w1 = t1[::8]; cand = stats[:, 0].reshape(-1, 8).sum(1) if Q % 8 == 0 else None
base = w0.min(); w0 = (w0 - base) * 0.01; w1 = (w1 - base) * 0.01     # us (100 MHz)
print("waves", len(w0), "span us", w1.max(), " wave duration us: mean %.2f p50 %.2f p90 %.2f p99 %.2f max %.2f" % (np.mean(w1 - w0), *np.percentile(w1 - w0, [50, 90, 99]), (w1 - w0).max()))
edges = np.arange(0, w1.max() + BIN, BIN)
for a in edges:
    act = ((w0 < a + BIN) & (w1 > a)).sum()
    started = ((w0 >= a) & (w0 < a + BIN)).sum()
    print("  t %5.0f..%5.0f us: waves alive %5d  started %5d" % (a, a + BIN, act, started))
if cand is not None:
    d = w1 - w0
    order = np.argsort(cand)
    for q in (0.1, 0.5, 0.9, 0.99):
        k = order[int(q * (len(order) - 1))]
        print("  wave at candidates quantile %.2f: %d candidates, %.2f us" % (q, cand[k], d[k]))
    print("  corr(duration, candidates) %.3f ; us per 1000 candidates (fit) %.3f + %.3f" % (np.corrcoef(d, cand)[0, 1], *np.polyfit(cand / 1000.0, d, 1)))

# where in the launch the slow waves sit (wave index deciles)
nw = len(d)
for k in range(10):
    sl = slice(k * nw // 10, (k + 1) * nw // 10)
    print("  waves %5d..%5d: mean duration %.1f us, mean candidates %.0f, start %.1f..%.1f" % (sl.start, sl.stop, d[sl].mean(), cand[sl].mean(), w0[sl].min(), w0[sl].max()))
