"""Three LM iterations through the per-call API — the entry the kernels' timing taps print from (LVF_BACK_TIMING, LVF_CHOL_TIMING,
LVF_COST_TIMING, LVF_LIN_TIMING, LVF_SP_TIMING = 1).  usage: python tools/one_iteration.py [n_kf n_lm]   (default: configs[3])"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lvio_fusion_amd import api, synthetic as syn

ctx = api.Context(0)
if len(sys.argv) > 2:
    n_kf, n_lm = int(sys.argv[1]), int(sys.argv[2])
    cfg = syn.config4_window(n_kf=n_kf, n_lm=n_lm, n_prewindow=max(n_lm // 5, 1), seed=0xC0FFEE)
else:
    cfg = syn.config4_window(seed=0xC0FFEE)
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
opt = api.default_solver_options()
r, d = 1e4, 2.0
for _ in range(3):
    o = prob.lm_iteration(opt, r, d); r, d = o["radius"], o["decrease_factor"]
print("cost", o["cost_after"])
