import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from lvio_fusion_amd import api, synthetic as syn
import bench
ctx = api.Context(0)
for n_kf, n_lm in ((5, 600), (10, 1500), (16, 3000), (20, 4000), (50, 10000)):
    cfg = syn.config4_window(n_kf=n_kf, n_lm=n_lm, n_prewindow=max(100, n_lm // 5))
    pre = api.preintegrate_or_none(ctx, cfg)
    st = api.State(ctx, cfg["n_kf"], cfg["n_lm"])
    for field, key in ((api.POSES, "poses"), (api.VEL, "vel"), (api.BA, "ba"), (api.BG, "bg"), (api.INV_DEPTH, "inv_depth"), (api.W_VISUAL, "w_kf")):
        st.set(field, cfg[key])
    tc, tf, po = cfg["tc"], cfg["tf"], cfg["po"]
    hs = [api.two_camera_batch(ctx, cfg["cam0"], cfg["cam1"], tc["left_ob"], tc["right_ob"], tc["lm_idx"], tc["kf_idx"]),
          api.two_frame_batch(ctx, cfg["cam0"], cfg["cam1"], tf["first_ob"], tf["ob"], tf["lm_idx"], tf["kf1_idx"], tf["kf2_idx"]),
          api.pose_only_batch(ctx, cfg["cam0"], po["ob"], po["kf_idx"], po["pw_idx"], po["pw"]),
          api.imu_batch(ctx, pre, [f["kf_i"] for f in cfg["imu"]], [f["kf_j"] for f in cfg["imu"]])]
    prob = api.Problem(ctx, st, *hs)
    opt = bench.fixed_iterations(api, 20)
    ts = []
    for rep in range(5):
        bench.reset_state(api, st, cfg)
        ctx.synchronize(); t0 = time.perf_counter(); s = prob.solve(opt); ts.append(time.perf_counter() - t0)
    bench.reset_state(api, st, cfg)
    stages = prob.stage_times(api.default_solver_options(), reps=10)
    print(f"FORCE={os.environ.get('LVF_FORCE_LEVELS','-')} {n_kf:3d} KF: {1e3 * min(ts) / s.num_iterations:.4f} ms/it cost {s.final_cost:.6g}; " + ", ".join(f"{n.split(' ')[0][2:8]} {us:.0f}x{l}" for n, us, l in stages if l))
