"""Batches of SMALL sliding windows (the reference's RL environments: 10-keyframe windows, 8 train / 100 test — environment.cpp:18-115,
td3.py:44-45) through lvf_problem_batch_solve: aggregate LM iterations/s for W = 1 .. 100 windows of n_kf keyframes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lvio_fusion_amd import api, synthetic as syn
import bench

n_kf = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n_lm = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
Ws = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 8, 32, 100]
iters = 20
ctx = api.Context(0)
wins = []
for i in range(max(Ws)):
    cfg = syn.config4_window(n_kf=n_kf, n_lm=n_lm, n_prewindow=max(100, n_lm // 5), seed=0x5A11 + i)
    pre = api.preintegrate_or_none(ctx, cfg)
    st = api.State(ctx, cfg["n_kf"], cfg["n_lm"])
    for field, key in ((api.POSES, "poses"), (api.VEL, "vel"), (api.BA, "ba"), (api.BG, "bg"), (api.INV_DEPTH, "inv_depth"), (api.W_VISUAL, "w_kf")):
        st.set(field, cfg[key])
    tc, tf, po = cfg["tc"], cfg["tf"], cfg["po"]
    hs = [api.two_camera_batch(ctx, cfg["cam0"], cfg["cam1"], tc["left_ob"], tc["right_ob"], tc["lm_idx"], tc["kf_idx"]),
          api.two_frame_batch(ctx, cfg["cam0"], cfg["cam1"], tf["first_ob"], tf["ob"], tf["lm_idx"], tf["kf1_idx"], tf["kf2_idx"]),
          api.pose_only_batch(ctx, cfg["cam0"], po["ob"], po["kf_idx"], po["pw_idx"], po["pw"]),
          api.imu_batch(ctx, pre, [f["kf_i"] for f in cfg["imu"]], [f["kf_j"] for f in cfg["imu"]])]
    wins.append((cfg, st, hs, api.Problem(ctx, st, *hs)))
print(f"{n_kf} KF / {n_lm} landmarks: {len(wins[0][0]['tf']['lm_idx'])} TwoFrame, {len(wins[0][0]['tc']['lm_idx'])} TwoCamera, {len(wins[0][0]['po']['kf_idx'])} PoseOnly blocks per window")
opt = bench.fixed_iterations(api, iters)
for W in Ws:
    b = api.ProblemBatch(ctx, [w[3] for w in wins[:W]])
    rates = []
    for rep in range(4):
        for cfg, st, _, _ in wins[:W]:
            bench.reset_state(api, st, cfg)
        ctx.synchronize(); t0 = time.perf_counter(); ss = b.solve(opt); dt = time.perf_counter() - t0
        if rep:
            rates.append(sum(s.num_iterations for s in ss) / dt)
    r = float(np.median(rates))
    print(f"W = {W:4d}: tables={b.uses_tables(opt)}  {1e3 * W / r:.3f} ms per batched iteration, {r:.0f} LM it/s aggregate")
    b.close()
