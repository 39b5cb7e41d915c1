"""Timing harness: configs[3] LM iteration rate — single window (per-call API, device loop) and batches of W windows."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lvio_fusion_amd import api, synthetic as syn


def build(ctx, seed):
    cfg = syn.config4_window(seed=seed, ids_by_birth="birth" in sys.argv[3:])
    pre = api.preintegrate_or_none(ctx, cfg)
    st = api.State(ctx, cfg["n_kf"], cfg["n_lm"])
    for field, key in ((api.POSES, "poses"), (api.VEL, "vel"), (api.BA, "ba"), (api.BG, "bg"), (api.INV_DEPTH, "inv_depth"), (api.W_VISUAL, "w_kf")):
        st.set(field, cfg[key])
    tc, tf, po = cfg["tc"], cfg["tf"], cfg["po"]
    hs = [api.two_camera_batch(ctx, cfg["cam0"], cfg["cam1"], tc["left_ob"], tc["right_ob"], tc["lm_idx"], tc["kf_idx"]),
          api.two_frame_batch(ctx, cfg["cam0"], cfg["cam1"], tf["first_ob"], tf["ob"], tf["lm_idx"], tf["kf1_idx"], tf["kf2_idx"]),
          api.pose_only_batch(ctx, cfg["cam0"], po["ob"], po["kf_idx"], po["pw_idx"], po["pw"]),
          api.imu_batch(ctx, pre, [f["kf_i"] for f in cfg["imu"]], [f["kf_j"] for f in cfg["imu"]])]
    return cfg, st, hs, api.Problem(ctx, st, *hs)


def reset(st, cfg):
    for field, key in ((api.POSES, "poses"), (api.VEL, "vel"), (api.BA, "ba"), (api.BG, "bg"), (api.INV_DEPTH, "inv_depth")):
        st.set(field, cfg[key])


def single(ctx, opt, iters, cfg, st, prob):
    # per-call API
    r, d = 1e4, 2.0
    for _ in range(3):
        o = prob.lm_iteration(opt, r, d); r, d = o["radius"], o["decrease_factor"]
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(iters):
        o = prob.lm_iteration(opt, r, d); r, d = o["radius"], o["decrease_factor"]
    dt = time.perf_counter() - t0
    print(f"per-call lm_iteration: {1e3 * dt / iters:.3f} ms/it ({iters / dt:.0f} it/s)")
    for rep in range(2):
        reset(st, cfg)
        t0 = time.perf_counter(); s = prob.solve(opt); dt = time.perf_counter() - t0
    print(f"device loop, 1 window: {1e3 * dt / s.num_iterations:.3f} ms/it ({s.num_iterations / dt:.0f} it/s), {s.num_iterations} its, cost {s.initial_cost:.4g} -> {s.final_cost:.4g}")


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    Ws = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 4, 8, 16]
    ctx = api.Context(0)
    opt = api.default_solver_options()
    opt.max_num_iterations = iters; opt.function_tolerance = 0.0; opt.parameter_tolerance = 0.0; opt.gradient_tolerance = 0.0
    wins = [build(ctx, 0xC0FFEE + i) for i in range(max(Ws))]
    batch_only = "batchonly" in sys.argv[3:]       # (PMC passes of the batched chain alone: tools/prof_batch_pmc.sh)
    cfg, st, hs, prob = wins[0]
    if not batch_only:
        single(ctx, opt, iters, cfg, st, prob)
    for W in Ws:
        b = api.ProblemBatch(ctx, [w[3] for w in wins[:W]])
        for rep in range(1 if batch_only else 2):
            for w in wins[:W]:
                reset(w[1], w[0])
            t0 = time.perf_counter(); ss = b.solve(opt); dt = time.perf_counter() - t0
        n = sum(x.num_iterations for x in ss)
        print(f"batch of {W}: tables={b.uses_tables(opt)} {1e3 * dt / ss[0].num_iterations:.3f} ms per batched iteration, {n / dt:.0f} LM it/s aggregate")
        b.close()


main()
