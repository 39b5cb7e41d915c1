"""Diagnostic: distribution of per-query work in the grid 3-NN (config 3)."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lvio_fusion_amd import api, synthetic as syn, _lib
c3 = syn.config3_icp()
ctx = api.Context(0)
for thr in (c3["thr_ground"], c3["thr_surf"]):
    mp = api.Map(ctx, c3["map"], thr); sc = api.Scan(ctx, c3["query"])
    Q = sc.Q
    st = np.zeros((Q, 4), np.int32); lv = np.zeros((8, 4), np.float32); nl = C.c_int()
    pose = np.ascontiguousarray(c3["pose0"])
    api._chk(ctx.L.lvf_knn3_debug_stats(mp.h, sc.h, pose.ctypes.data_as(_lib.c_double_p), float(thr), st.ctypes.data_as(_lib.c_int_p), lv.ctypes.data_as(_lib.c_float_p), C.byref(nl)))
    print("thr", thr, "levels", lv[:nl.value].tolist())
    for k, name in enumerate(("candidates", "lookups", "level", "shells")):
        v = st[:, k]
        print(f"  {name:10s} mean {v.mean():9.1f} p50 {np.percentile(v,50):7.0f} p90 {np.percentile(v,90):7.0f} p99 {np.percentile(v,99):8.0f} max {v.max():8d}")
    w = st.reshape(-1, 64, 4) if Q % 64 == 0 else st[:Q // 64 * 64].reshape(-1, 64, 4)
    print("  per-wave max candidates mean", w[:, :, 0].max(1).mean(), " per-wave max lookups mean", w[:, :, 1].max(1).mean())
    print("  level histogram", np.bincount(st[:, 2], minlength=8).tolist())
    mp.close(); sc.close()
