"""Trajectory parity of the DEVICE-RESIDENT LM loop — the call bench.py times (lvf_problem_solve / lvf_problem_batch_solve) — against
the oracle's restatement of ceres::Solve's TrustRegionMinimizer loop (oracle/lm.h lm_solve = chained trial steps with ITS radius and
decrease factor, Ceres' termination order).  What adapt::Solve runs: backend.cpp:206-211, mapping.cpp:159-163.

Compared after prob.solve(max_num_iterations = K): the whole final state, final / initial cost, num_iterations, successful and
unsuccessful steps, termination type and WHICH test ended the loop.  Cases: the default start, starts whose steps are rejected
(a wrong radius update after a rejection would change every later iterate), each termination rule, the batched loop, call
sequences that re-enter a problem (solve -> cost -> solve), and configs[3] at full size with K = 20 (bench.py's exact call)."""
import numpy as np
import pytest

from lvio_fusion_amd import synthetic as syn
from tests.helpers import assert_parity

pytestmark = pytest.mark.gpu

FIELDS = ("poses", "vel", "ba", "bg", "inv_depth")


@pytest.fixture(scope="module")
def ctx():
    from lvio_fusion_amd import api
    c = api.Context(0)
    yield c
    c.close()


def make(api, ctx, oracle, n_kf, n_lm, seed, n_pre=40, perturb=1.0, imu_samples=5, cfg=None, pose_const=None):
    """One window on the device and the same window in the oracle.  perturb > 1 re-perturbs poses / inverse depths harder than the
    generator's sigma (0.5 deg, 5 cm, 5 %): far starts are where LM rejects steps."""
    if cfg is None:
        cfg = syn.config4_window(n_kf=n_kf, n_lm=n_lm, n_prewindow=n_pre, seed=seed, imu_samples=imu_samples)
    if perturb != 1.0:
        rng = np.random.default_rng(seed + 1000)
        cfg = dict(cfg)
        cfg["poses"] = syn.perturb_poses(cfg["poses"], rng, 0.5 * perturb, 0.05 * perturb)
        cfg["inv_depth"] = cfg["inv_depth"] * (1 + rng.normal(0, 0.05 * min(perturb, 8.0), cfg["inv_depth"].shape))
    pre = np.stack([oracle.imu_preintegrate(f["samples"], f["acc0"], f["gyr0"], f["ba"], f["bg"], syn.IMU_NOISE) for f in cfg["imu"]])
    st = api.State(ctx, cfg["n_kf"], cfg["n_lm"])
    for field, key in ((api.POSES, "poses"), (api.VEL, "vel"), (api.BA, "ba"), (api.BG, "bg"), (api.INV_DEPTH, "inv_depth"), (api.W_VISUAL, "w_kf")):
        st.set(field, cfg[key])
    tc, tf, po = cfg["tc"], cfg["tf"], cfg["po"]
    hs = [api.two_camera_batch(ctx, cfg["cam0"], cfg["cam1"], tc["left_ob"], tc["right_ob"], tc["lm_idx"], tc["kf_idx"]),
          api.two_frame_batch(ctx, cfg["cam0"], cfg["cam1"], tf["first_ob"], tf["ob"], tf["lm_idx"], tf["kf1_idx"], tf["kf2_idx"]),
          api.pose_only_batch(ctx, cfg["cam0"], po["ob"], po["kf_idx"], po["pw_idx"], po["pw"]),
          api.imu_batch(ctx, pre, [f["kf_i"] for f in cfg["imu"]], [f["kf_j"] for f in cfg["imu"]])]
    prob = api.Problem(ctx, st, *hs)
    if pose_const is not None:
        for k in np.flatnonzero(pose_const):
            prob.set_pose_constant(int(k), True)
    win = oracle.Window(cfg, pre, pose_const=pose_const)
    return dict(cfg=cfg, st=st, hs=hs, prob=prob, win=win)


def close(*ws):
    for w in ws:
        w["prob"].close()
        for h in w["hs"] + [w["st"]]:
            h.close()


def options(api, **kw):
    o = api.default_solver_options()
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def okw(o):
    return dict(max_num_iterations=o.max_num_iterations, huber_a=o.huber_a, initial_trust_region_radius=o.initial_trust_region_radius,
                function_tolerance=o.function_tolerance, gradient_tolerance=o.gradient_tolerance, parameter_tolerance=o.parameter_tolerance,
                min_relative_decrease=o.min_relative_decrease)


def check(api, w, s, ref, what, state_rtol_note=""):
    """device summary + state == oracle summary + state"""
    tag = f"{what}: device ({s.num_iterations} it, {s.num_successful_steps} ok, {s.num_unsuccessful_steps} rejected, {s.why}) vs oracle " \
          f"({ref['num_iterations']} it, {ref['num_successful_steps']} ok, {ref['num_unsuccessful_steps']} rejected, {ref['why']})"
    assert (s.num_iterations, s.num_successful_steps, s.num_unsuccessful_steps) == (ref["num_iterations"], ref["num_successful_steps"], ref["num_unsuccessful_steps"]), tag
    assert s.termination == ref["termination"] and s.why == ref["why"], tag
    assert abs(s.initial_cost - ref["initial_cost"]) <= 1e-9 * abs(ref["initial_cost"]), tag
    assert abs(s.final_cost - ref["final_cost"]) <= 1e-6 * abs(ref["final_cost"]), tag
    for k, f in zip(FIELDS, (api.POSES, api.VEL, api.BA, api.BG, api.INV_DEPTH)):
        got = np.asarray(w["st"].get(f)); want = np.asarray(getattr(w["win"], k))
        assert_parity(got.reshape(want.shape), want, f"{what}: {k} after the solve")


@pytest.mark.parametrize("n_kf,n_lm,seed,K", [(8, 300, 5, 25), (12, 500, 21, 20), (5, 120, 33, 12), (20, 900, 41, 10)])
def test_device_loop_equals_oracle_chain(ctx, oracle, n_kf, n_lm, seed, K):
    from lvio_fusion_amd import api
    w = make(api, ctx, oracle, n_kf, n_lm, seed)
    o = options(api, max_num_iterations=K)
    ref = w["win"].solve(**okw(o))
    s = w["prob"].solve(o)
    check(api, w, s, ref, f"{n_kf} KF / {n_lm} landmarks, K = {K}")
    assert ref["num_successful_steps"] >= 3
    close(w)


def weak_window(n_kf=8, n_lm=300, seed=5, w_weak=4e-7):
    """A window with NEAR-ZERO columns: the last keyframe's visual weight is ~0, so the landmarks born there (seen by their TwoCamera
    block only, weight 5 w) have C_l = (5 w dpx/drho)^2 < 1e-6, and their inverse depths start 50 % off so that the columns carry a
    gradient.  There Ceres' clamp acts on the Jacobi-SCALED diagonal: damping 1e-6 (1 + sqrt(H0_jj))^2 / radius instead of
    max(H_jj, 1e-6) / radius (oracle/lm.h header).  With a small trust region the two give different iterates."""
    cfg = dict(syn.config4_window(n_kf=n_kf, n_lm=n_lm, n_prewindow=40, seed=seed, imu_samples=5))
    cfg["w_kf"] = np.array(cfg["w_kf"], dtype=np.float64).copy(); cfg["w_kf"][-1] = w_weak
    born_last = cfg["tc"]["lm_idx"][cfg["tc"]["kf_idx"] == n_kf - 1]
    cfg["inv_depth"] = np.array(cfg["inv_depth"], dtype=np.float64).copy(); cfg["inv_depth"][born_last] *= 1.5
    return cfg, born_last


@pytest.mark.parametrize("radius,K", [(1e-5, 12), (1e-6, 12)])
def test_near_zero_columns_take_ceres_jacobi_scaled_damping(ctx, oracle, radius, K):
    """the device loop follows Ceres' Jacobi column scaling (s_j = 1 / (1 + sqrt(H0_jj)) frozen at iteration 0, clamp on
    the scaled diagonal) — on a window where that differs from a clamp on the unscaled diagonal by 100x the parity tolerance."""
    from lvio_fusion_amd import api
    cfg, born_last = weak_window()
    w = make(api, ctx, oracle, cfg["n_kf"], cfg["n_lm"], 5, cfg=cfg)
    o = options(api, max_num_iterations=K, initial_trust_region_radius=radius)
    ref = w["win"].solve(**okw(o))
    # the case discriminates: the pre-round-5 damping lands somewhere else
    pre = np.stack([oracle.imu_preintegrate(f["samples"], f["acc0"], f["gyr0"], f["ba"], f["bg"], syn.IMU_NOISE) for f in cfg["imu"]])
    old = oracle.Window(cfg, pre)
    old.solve(unscaled_clamp=True, **okw(o))
    gap = np.abs(old.inv_depth - w["win"].inv_depth)[born_last] / np.abs(w["win"].inv_depth[born_last])
    assert gap.max() > 1e-5, f"the window no longer tells the scaled clamp from the unscaled one ({gap.max():.2e})"
    s = w["prob"].solve(o)
    check(api, w, s, ref, f"weak last keyframe, radius {radius:g}")
    got = np.asarray(w["st"].get(api.INV_DEPTH))[born_last]
    assert np.abs(got - w["win"].inv_depth[born_last]).max() <= 1e-6 * np.abs(got).max()
    assert np.abs(got - old.inv_depth[born_last]).max() > 10e-6 * np.abs(got).max(), "the device follows the UNSCALED clamp"
    close(w)


def test_near_zero_columns_batched_and_per_iteration(ctx, oracle):
    """the same corner through the batched loop (tables) and through the per-iteration entry point (one-iteration solves: the scaling is
    retaken at every call, as chained one-iteration ceres::Solve calls would)"""
    from lvio_fusion_amd import api
    cfg, born_last = weak_window()
    ws = [make(api, ctx, oracle, cfg["n_kf"], cfg["n_lm"], 5, cfg=cfg) for _ in range(2)] + [make(api, ctx, oracle, 8, 300, 6)]
    o = options(api, max_num_iterations=8, initial_trust_region_radius=1e-5)
    b = api.ProblemBatch(ctx, [w["prob"] for w in ws])
    ss = b.solve(o)
    for i, w in enumerate(ws):
        check(api, w, ss[i], w["win"].solve(**okw(o)), f"batched weak window {i}")
    b.close()
    close(*ws)
    w = make(api, ctx, oracle, cfg["n_kf"], cfg["n_lm"], 5, cfg=cfg)
    r, d = 1e-5, 2.0
    for it in range(4):
        g = w["prob"].lm_iteration(options(api), r, d)
        ref = w["win"].lm_iteration(r, d)
        assert abs(g["cost_after"] - ref["cost_after"]) <= 1e-6 * abs(ref["cost_after"]) and bool(g["accepted"]) == ref["accepted"], (it, g, ref["cost_after"])
        r, d = ref["radius"], ref["decrease_factor"]
        got = np.asarray(w["st"].get(api.INV_DEPTH))[born_last]
        assert np.abs(got - w["win"].inv_depth[born_last]).max() <= 1e-6 * np.abs(got).max(), it
    close(w)


@pytest.mark.parametrize("perturb,radius,seed", [(20.0, 1e16, 5), (20.0, 1e4, 5), (30.0, 1e10, 9)])
def test_rejected_steps_follow_the_oracle(ctx, oracle, perturb, radius, seed):
    """Far starts: several steps are rejected (radius / decrease_factor, decrease_factor doubling) before the loop recovers; every later
    iterate depends on those radius updates."""
    from lvio_fusion_amd import api
    w = make(api, ctx, oracle, 8, 300, seed, n_pre=60, perturb=perturb)
    o = options(api, max_num_iterations=25, initial_trust_region_radius=radius)
    ref = w["win"].solve(**okw(o))
    assert ref["num_unsuccessful_steps"] >= 2, f"the case no longer rejects steps: {ref}"
    s = w["prob"].solve(o)
    check(api, w, s, ref, f"perturbation x{perturb}, radius {radius:g}")
    close(w)


def test_termination_rules_in_ceres_order(ctx, oracle):
    from lvio_fusion_amd import api
    # (options, expected reason) — each on a fresh window, device vs oracle
    cases = [(dict(max_num_iterations=3), "max_num_iterations"),
             (dict(max_num_iterations=50, function_tolerance=1e-3), "function_tolerance"),      # stops BEFORE taking the last candidate
             (dict(max_num_iterations=50, gradient_tolerance=1e30), "gradient_tolerance"),      # at the first linearisation: zero iterations
             (dict(max_num_iterations=50, parameter_tolerance=1e3), "parameter_tolerance"),     # first trial step already "small": not taken
             (dict(max_num_iterations=50, initial_trust_region_radius=1e-33), "min_trust_region_radius"),
             (dict(max_num_iterations=1), "max_num_iterations")]
    for kw, why in cases:
        w = make(api, ctx, oracle, 8, 200, 77)
        o = options(api, **kw)
        ref = w["win"].solve(**okw(o))
        assert ref["why"] == why, (kw, ref["why"])
        s = w["prob"].solve(o)
        check(api, w, s, ref, f"{kw}")
        if why in ("gradient_tolerance", "parameter_tolerance", "min_trust_region_radius"):
            assert s.num_iterations == 0 and s.final_cost == s.initial_cost
            assert np.array_equal(np.asarray(w["st"].get(api.POSES)).reshape(-1, 7), w["cfg"]["poses"]), "the state must not move"
        close(w)


def test_constant_pose_and_reentry(ctx, oracle):
    """A window with its first pose constant (x_norm leaves it out), solved twice with a cost query in between: the second solve starts
    from clean accumulators (solve -> cost -> solve) and from the initial radius again, like two ceres::Solve calls."""
    from lvio_fusion_amd import api
    pc = np.zeros(10, np.uint8); pc[0] = 1
    w = make(api, ctx, oracle, 10, 400, 63, pose_const=pc)
    o = options(api, max_num_iterations=4)
    ref1 = w["win"].solve(**okw(o))
    s1 = w["prob"].solve(o)
    check(api, w, s1, ref1, "first solve")
    c = w["prob"].cost(o)
    assert abs(c - ref1["final_cost"]) <= 1e-6 * c
    ref2 = w["win"].solve(**okw(o))
    s2 = w["prob"].solve(o)
    assert abs(s2.initial_cost - ref1["final_cost"]) <= 1e-6 * ref1["final_cost"], "second solve must not see the cost query's sum"
    check(api, w, s2, ref2, "second solve")
    # ... and through the per-call API after a cost query
    c = w["prob"].cost(o)
    g = w["prob"].lm_iteration(o, 1e4, 2.0)
    r = w["win"].lm_iteration(1e4, 2.0)
    assert abs(g["cost_before"] - r["cost_before"]) <= 1e-8 * r["cost_before"] and bool(g["accepted"]) == bool(r["accepted"])
    close(w)


def test_batched_loop_equals_oracle_chains(ctx, oracle):
    """lvf_problem_batch_solve: windows of different shapes and starts, one launch chain, per-window accept / reject / termination."""
    from lvio_fusion_amd import api
    specs = [(8, 300, 5, 1.0), (8, 300, 5, 20.0), (12, 500, 21, 1.0), (6, 150, 2, 10.0), (10, 400, 63, 1.0)]
    ws = [make(api, ctx, oracle, n_kf, n_lm, seed, n_pre=60, perturb=p) for n_kf, n_lm, seed, p in specs]
    o = options(api, max_num_iterations=25, initial_trust_region_radius=1e16)
    refs = [w["win"].solve(**okw(o)) for w in ws]
    assert any(r["num_unsuccessful_steps"] >= 1 for r in refs)
    batch = api.ProblemBatch(ctx, [w["prob"] for w in ws])
    assert batch.uses_tables(o) == 1
    out = batch.solve(o)
    for i, (w, s, ref) in enumerate(zip(ws, out, refs)):
        check(api, w, s, ref, f"batch member {i} {specs[i]}")
    batch.close(); close(*ws)


def test_config3_k20_the_call_bench_times(ctx, oracle):
    """configs[3] at BASELINE size: 50 keyframes, 10 000 landmarks (+ 2 000 pre-window), 49 ImuError factors; prob.solve(K = 20) — the
    call bench.py's headline loop makes — against the oracle chain (~3 s of CPU)."""
    from lvio_fusion_amd import api
    cfg = syn.config4_window()
    w = make(api, ctx, oracle, 0, 0, syn.SEED_CFG4, cfg=cfg)
    o = options(api, max_num_iterations=20)
    ref = w["win"].solve(**okw(o))
    s = w["prob"].solve(o)
    check(api, w, s, ref, "configs[3], K = 20")
    assert s.num_iterations == 20 or s.termination == 0
    close(w)


def test_many_small_windows_in_one_batch(ctx, oracle):
    """The reference's RL-environment shape (10-keyframe windows, up to 100 at once: environment.cpp:18-115, td3.py:44-45): 40 small windows
    with different seeds / starts in one lvf_problem_batch_solve, every window against its own oracle chain."""
    from lvio_fusion_amd import api
    ws = [make(api, ctx, oracle, 10, 300, 900 + i, n_pre=60, perturb=(1.0 if i % 4 else 12.0)) for i in range(40)]
    o = options(api, max_num_iterations=10)
    refs = [w["win"].solve(**okw(o)) for w in ws]
    batch = api.ProblemBatch(ctx, [w["prob"] for w in ws])
    assert batch.uses_tables(o) == 1
    out = batch.solve(o)
    for i, (w, s, ref) in enumerate(zip(ws, out, refs)):
        check(api, w, s, ref, f"small window {i} of 40")
    batch.close(); close(*ws)
