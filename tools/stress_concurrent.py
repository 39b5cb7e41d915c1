"""configs[3] prob.solve(20) repeated on one context while a second host thread keeps another context of the same GPU busy (loop-closure
candidates with full-size scans): every solve must equal the serial one.  Prints the runs that differ.  usage: stress_concurrent.py [runs]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lvio_fusion_amd import api, synthetic as syn, relocalize as rl
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 60
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
opt = api.default_solver_options(); opt.max_num_iterations = 20; opt.function_tolerance = 0.0; opt.parameter_tolerance = 0.0; opt.gradient_tolerance = 0.0
serial = prob.solve(opt)
print("serial", serial.final_cost, serial.num_iterations, serial.hand_over_retries)
cands = syn.config5_candidates(2, seed=8, n_query=100000, n_az=2900, overlap="full")
stop, laps = threading.Event(), [0]
MODE = os.environ.get("TRAFFIC", "candidates")      # candidates | tiny_knn | big_knn | state_copy | none
def traffic():
    c2 = api.Context(0)
    if MODE == "candidates":
        while not stop.is_set():
            rl.evaluate_candidate(api, c2, cands[laps[0] % 2]); laps[0] += 1
    elif MODE in ("tiny_knn", "big_knn"):
        c = cands[0]
        q = c["query"][:8] if MODE == "tiny_knn" else c["query"]
        mp = api.Map(c2, c["map"], 4.0); sc = api.Scan(c2, q)
        while not stop.is_set():
            for _ in range(50):
                api.knn3(mp, sc, c["init_pose"], 4.0)
            c2.synchronize(); laps[0] += 1
    elif MODE == "state_copy":
        s2 = api.State(c2, 50, 10000); x = np.zeros(10000)
        while not stop.is_set():
            s2.set(api.INV_DEPTH, x); laps[0] += 1
    else:
        while not stop.is_set():
            time.sleep(0.01)
    c2.close()
t = threading.Thread(target=traffic); t.start()
bad = 0
t0 = time.perf_counter()
for run in range(runs):
    for f, k in F:
        st.set(f, cfg[k])
    s = prob.solve(opt)
    d = abs(s.final_cost - serial.final_cost) / abs(serial.final_cost)
    if d > 1e-9 or s.num_iterations != serial.num_iterations:
        bad += 1
        print("run", run, "differs: rel", d, "iterations", s.num_iterations, "successes", s.num_successful_steps, "retries", s.hand_over_retries, "why", s.why)
        if os.environ.get("LVF_LM_HISTORY"):
            h = np.zeros(512); api._chk(ctx.L.lvf_problem_debug_history(prob.h, h.ctypes.data_as(api._lib.c_double_p)))
            for row in h.reshape(64, 8)[:20]:
                if row[4] == 0:
                    print("   it %2d cost %.9e cand %.9e model %.6e acc %d fail %d radius %.4e gmax %.4e" % (row[0], row[1], row[2], row[3], row[4], row[5], row[6], row[7]))
stop.set(); t.join()
print(f"{bad} of {runs} runs differ; retries {s.hand_over_retries}; traffic laps {laps[0]}; {time.perf_counter() - t0:.1f} s")
