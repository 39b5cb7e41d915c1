"""Distribution of the association's per-query work (candidates, cell look-ups, last level, shells): tools/knn_stats.py"""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lvio_fusion_amd import api, _lib, synthetic

def main():
    ctx = api.Context()
    c3 = synthetic.config3_icp()
    mpts, spts = c3["map"], c3["query"]
    mp = api.Map(ctx, mpts, c3["thr_ground"]); sc = api.Scan(ctx, spts)
    Q = len(spts)
    for name, thr in (("ground", c3["thr_ground"]), ("surf", c3["thr_surf"])):
        pose = np.asarray(c3["pose0"], np.float64)
        stats = np.zeros((Q, 6), np.int32); lv = np.zeros((8, 4), np.float32); nl = C.c_int()
        api._chk(ctx.L.lvf_knn3_debug_stats2(mp.h, sc.h, pose.ctypes.data_as(_lib.c_double_p), float(thr), stats.ctypes.data_as(_lib.c_int_p), 6,
                                            lv.ctypes.data_as(_lib.c_float_p), C.byref(nl)))
        print(name, "Q", Q, "M", len(mpts), "levels", nl.value, lv[:nl.value].tolist())
        for k, nm in enumerate(("candidates", "lookups", "level", "shells")):
            v = stats[:, k]
            print("  %-10s mean %.1f  p10 %d p50 %d p90 %d p99 %d max %d" % (nm, v.mean(), *np.percentile(v, [10, 50, 90, 99]).astype(int), v.max()))
        print("  shells hist", np.bincount(stats[:, 3])[:12].tolist(), " level hist", np.bincount(stats[:, 2]).tolist())

if __name__ == "__main__":
    main()
