"""A batch of W configs[3] windows (lvf_problem_batch_solve) and a batch of small windows repeated while a second context keeps the GPU busy:
every window's result must equal the undisturbed one.  usage: stress_concurrent_batch.py [runs] [W]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lvio_fusion_amd import api, synthetic as syn
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 60
W = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ctx = api.Context(0)
F = ((api.POSES, "poses"), (api.VEL, "vel"), (api.BA, "ba"), (api.BG, "bg"), (api.INV_DEPTH, "inv_depth"))
def build(seed, **kw):
    cfg = syn.config4_window(seed=seed, **kw)
    pre = api.preintegrate_or_none(ctx, cfg)
    st = api.State(ctx, cfg["n_kf"], cfg["n_lm"])
    for f, k in F + ((api.W_VISUAL, "w_kf"),):
        st.set(f, cfg[k])
    tc, tf, po = cfg["tc"], cfg["tf"], cfg["po"]
    hs = [api.two_camera_batch(ctx, cfg["cam0"], cfg["cam1"], tc["left_ob"], tc["right_ob"], tc["lm_idx"], tc["kf_idx"]),
          api.two_frame_batch(ctx, cfg["cam0"], cfg["cam1"], tf["first_ob"], tf["ob"], tf["lm_idx"], tf["kf1_idx"], tf["kf2_idx"]),
          api.pose_only_batch(ctx, cfg["cam0"], po["ob"], po["kf_idx"], po["pw_idx"], po["pw"]),
          api.imu_batch(ctx, pre, [f["kf_i"] for f in cfg["imu"]], [f["kf_j"] for f in cfg["imu"]])]
    return cfg, st, hs, api.Problem(ctx, st, *hs)
opt = api.default_solver_options(); opt.max_num_iterations = 12; opt.function_tolerance = 0.0; opt.parameter_tolerance = 0.0; opt.gradient_tolerance = 0.0
sets = {"configs[3] x %d" % W: [build(0xC0FFEE + i) for i in range(W)],
        "10-keyframe x 24": [build(0xBEEF + i, n_kf=10, n_lm=1500, n_prewindow=200) for i in range(24)]}
c3 = syn.config3_icp()
stop, laps = threading.Event(), [0]
def traffic():
    c2 = api.Context(0)
    mp, sc = api.Map(c2, c3["map"], 4.0), api.Scan(c2, c3["query"])
    while not stop.is_set():
        for _ in range(40):
            api.knn3(mp, sc, c3["pose0"], 4.0)
        c2.synchronize(); laps[0] += 1
    c2.close()
for name, wins in sets.items():
    b = api.ProblemBatch(ctx, [w[3] for w in wins])
    def run():
        for w in wins:
            for f, k in F:
                w[1].set(f, w[0][k])
        return [(s.final_cost, s.num_iterations, s.num_successful_steps, s.hand_over_retries) for s in b.solve(opt)]
    ref = run()
    stop.clear(); laps[0] = 0
    t = threading.Thread(target=traffic); t.start()
    bad = 0
    for r in range(runs):
        got = run()
        for i, (g, w) in enumerate(zip(got, ref)):
            if abs(g[0] - w[0]) > 1e-9 * abs(w[0]) or g[1:3] != w[1:3]:
                bad += 1; print(name, "run", r, "window", i, "got", g, "want", w); break
    stop.set(); t.join()
    print(f"{name}: tables={b.uses_tables(opt)} {bad} of {runs} runs differ; traffic laps {laps[0]}")
    b.close()
