"""Like stress_concurrent.py but through the per-call API: finds WHICH iteration goes a different way under a second context's traffic."""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lvio_fusion_amd import api, synthetic as syn, relocalize as rl
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 40
ctx = api.Context(0)
cfg = syn.config4_window()
pre = api.preintegrate_or_none(ctx, cfg)
st = api.State(ctx, cfg["n_kf"], cfg["n_lm"])
F = ((api.POSES, "poses"), (api.VEL, "vel"), (api.BA, "ba"), (api.BG, "bg"), (api.INV_DEPTH, "inv_depth"))
for f, k in F + ((api.W_VISUAL, "w_kf"),):
    st.set(f, cfg[k])
tc, tf, po = cfg["tc"], cfg["tf"], cfg["po"]
hs = [api.two_camera_batch(ctx, cfg["cam0"], cfg["cam1"], tc["left_ob"], tc["right_ob"], tc["lm_idx"], tc["kf_idx"]),
      api.two_frame_batch(ctx, cfg["cam0"], cfg["cam1"], tf["first_ob"], tf["ob"], tf["lm_idx"], tf["kf1_idx"], tf["kf2_idx"]),
      api.pose_only_batch(ctx, cfg["cam0"], po["ob"], po["kf_idx"], po["pw_idx"], po["pw"]),
      api.imu_batch(ctx, pre, [f["kf_i"] for f in cfg["imu"]], [f["kf_j"] for f in cfg["imu"]])]
prob = api.Problem(ctx, st, *hs)
opt = api.default_solver_options()
def trajectory():
    for f, k in F:
        st.set(f, cfg[k])
    r, d, out = 1e4, 2.0, []
    for it in range(12):
        o = prob.lm_iteration(opt, r, d); r, d = o["radius"], o["decrease_factor"]
        out.append((o["cost_before"], o["cost_after"], o["accepted"], r))
    return out
ref = trajectory()
cands = syn.config5_candidates(2, seed=8, n_query=100000, n_az=2900, overlap="full")
stop, laps = threading.Event(), [0]
def traffic():
    c2 = api.Context(0)
    while not stop.is_set():
        rl.evaluate_candidate(api, c2, cands[laps[0] % 2]); laps[0] += 1
    c2.close()
t = threading.Thread(target=traffic); t.start()
bad = 0
for run in range(runs):
    tr = trajectory()
    for it, (a, b) in enumerate(zip(tr, ref)):
        if a[2] != b[2] or abs(a[1] - b[1]) > 1e-9 * abs(b[1]) or abs(a[0] - b[0]) > 1e-9 * abs(b[0]):
            bad += 1
            print("run", run, "iteration", it, "got", a, "want", b)
            break
stop.set(); t.join()
print(bad, "of", runs, "trajectories differ; traffic laps", laps[0])
