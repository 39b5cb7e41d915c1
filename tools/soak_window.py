"""Soak run of the persistent window: a long trajectory replayed tick by tick (add keyframe + IMU + landmarks + observations, slide, solve,
outlier gate every few ticks).  Reports the tick time, the census bounds and that nothing grows without bound."""
import os, sys, time, resource
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lvio_fusion_amd import api, synthetic as syn

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
W = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cfg = syn.config4_window(n_kf=N, n_lm=40 * N, n_prewindow=0, seed=777, imu_samples=4)
ctx = api.Context(0)
pre = api.preintegrate_or_none(ctx, cfg)
win = api.Window(ctx, cfg["cam0"], cfg["cam1"], baseline=syn.baseline())
opt = api.default_solver_options(); opt.max_num_iterations = 3
tc, tf = cfg["tc"], cfg["tf"]
births = {}
for i, (l, k) in enumerate(zip(tc["lm_idx"], tc["kf_idx"])):
    births.setdefault(int(k), []).append(i)
obs_by_kf = {}
for i, k in enumerate(tf["kf2_idx"]):
    obs_by_kf.setdefault(int(k), []).append(i)
ticks, worst, known_max, removed_total = [], 0.0, 0, 0
rss60 = 0
rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
for t in range(N):
    win.add_keyframe(1000 + t, cfg["poses"][t], cfg["w_kf"][t])
    win.set_imu(1000 + t, cfg["vel"][t], cfg["ba"][t], cfg["bg"][t], pre[t - 1] if t > 0 else None)
    for i in births.get(t, []):
        win.add_landmark(int(tc["lm_idx"][i]), 1000 + t, tc["left_ob"][i], tc["right_ob"][i], cfg["inv_depth"][tc["lm_idx"][i]])
    for i in obs_by_kf.get(t, []):
        win.add_observation(int(tf["lm_idx"][i]), 1000 + t, tf["ob"][i])
    win.slide(1000 + max(0, t - W + 1))
    t0 = time.perf_counter()
    s = win.solve(opt)
    dt = time.perf_counter() - t0
    assert np.isfinite(s.final_cost) and s.final_cost <= s.initial_cost * (1 + 1e-9), (t, s.initial_cost, s.final_cost)
    if t % 7 == 3:
        removed_total += win.reject_outliers(10.0)[1]
    c = win.counts()
    known_max = max(known_max, c["lm_known"])
    assert c["kf"] <= W
    if t >= 50:
        ticks.append(dt); worst = max(worst, dt)
    if t == 60:
        rss60 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
print(f"{N} ticks, window {W}: median tick {1e3 * np.median(ticks):.3f} ms (3 LM iterations), worst {1e3 * worst:.3f} ms, landmarks known <= {known_max}, "
      f"outliers removed {removed_total}, max RSS {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024:.0f} MB (at tick 60: {rss60 / 1024:.0f} MB), last census {win.counts()}")
win.close(); ctx.close()
