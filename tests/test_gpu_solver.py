"""GPU parity of the sliding-window BA solver (lvf_problem_*) against the oracle's LM iteration
(oracle/lm.h: dense normal equations + exact Schur on the inverse-depth blocks) from identical state."""
import numpy as np
import pytest

from lvio_fusion_amd import synthetic as syn
from tests.helpers import assert_parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from lvio_fusion_amd import api
    c = api.Context(0)
    yield c
    c.close()


def build(api, ctx, oracle, n_kf, n_lm, seed, n_pre=40, use=("tc", "tf", "po", "imu"), imu_drop=(), ids_by_birth=False):
    cfg = syn.config4_window(n_kf=n_kf, n_lm=n_lm, n_prewindow=n_pre, seed=seed, imu_samples=5, ids_by_birth=ids_by_birth)
    if imu_drop:       # IMU drop-outs: the (v, ba, bg) coupling graph falls apart into several chains and isolated blocks
        cfg["imu"] = [f for k, f in enumerate(cfg["imu"]) if k not in imu_drop]
    pre = np.stack([oracle.imu_preintegrate(f["samples"], f["acc0"], f["gyr0"], f["ba"], f["bg"], syn.IMU_NOISE) for f in cfg["imu"]])
    st = api.State(ctx, n_kf, n_lm)
    for field, key in ((api.POSES, "poses"), (api.VEL, "vel"), (api.BA, "ba"), (api.BG, "bg"), (api.INV_DEPTH, "inv_depth"), (api.W_VISUAL, "w_kf")):
        st.set(field, cfg[key])
    tc, tf, po = cfg["tc"], cfg["tf"], cfg["po"]
    b = dict(
        tc=api.two_camera_batch(ctx, cfg["cam0"], cfg["cam1"], tc["left_ob"], tc["right_ob"], tc["lm_idx"], tc["kf_idx"]) if "tc" in use else None,
        tf=api.two_frame_batch(ctx, cfg["cam0"], cfg["cam1"], tf["first_ob"], tf["ob"], tf["lm_idx"], tf["kf1_idx"], tf["kf2_idx"]) if "tf" in use else None,
        po=api.pose_only_batch(ctx, cfg["cam0"], po["ob"], po["kf_idx"], po["pw_idx"], po["pw"]) if "po" in use else None,
        imu=api.imu_batch(ctx, pre, [f["kf_i"] for f in cfg["imu"]], [f["kf_j"] for f in cfg["imu"]]) if "imu" in use else None)
    prob = api.Problem(ctx, st, b["tc"], b["tf"], b["po"], b["imu"])
    win = oracle.Window(cfg, pre, use=use)
    return cfg, st, b, prob, win


def state_of(api, st):
    return {k: st.get(f) for k, f in (("poses", api.POSES), ("vel", api.VEL), ("ba", api.BA), ("bg", api.BG), ("inv_depth", api.INV_DEPTH))}


@pytest.mark.parametrize("n_kf,n_lm,seed,use", [(8, 120, 3, ("tc", "tf", "po", "imu")), (12, 300, 5, ("tc", "tf", "po", "imu")),
                                                (6, 80, 7, ("tc", "tf", "po")), (5, 0, 9, ("po", "imu")),
                                                (7, 150, 29, ("tc", "po", "imu")),      # landmarks without any pose-dependent block
                                                (2, 40, 31, ("tc", "tf", "po", "imu"))])   # the smallest window with an IMU factor
def test_lm_iteration_parity(ctx, oracle, n_kf, n_lm, seed, use):
    from lvio_fusion_amd import api
    if n_lm == 0:
        use = tuple(u for u in use if u in ("po", "imu"))
    cfg, st, b, prob, win = build(api, ctx, oracle, n_kf, max(n_lm, 1), seed, use=use)
    opt = api.default_solver_options()
    c_gpu = prob.cost(opt)
    assert abs(c_gpu - win.cost()) <= 1e-9 * abs(c_gpu)
    radius, dec = 1e4, 2.0
    for it in range(4):
        ref = win.lm_iteration(radius, dec)
        got = prob.lm_iteration(opt, radius, dec)
        assert abs(got["cost_before"] - ref["cost_before"]) <= 1e-8 * abs(ref["cost_before"])
        S, rhs = prob.reduced_system()
        scale = np.abs(ref["S"]).max()
        assert np.abs(S - ref["S"]).max() <= 1e-7 * scale, f"iteration {it}: reduced system mismatch"
        assert_parity(rhs, ref["rhs"], f"rhs it{it}")
        assert got["accepted"] == ref["accepted"]
        assert abs(got["cost_after"] - ref["cost_after"]) <= 1e-6 * abs(ref["cost_after"])
        assert abs(got["radius"] - ref["radius"]) <= 1e-5 * ref["radius"]
        s = state_of(api, st)
        assert_parity(s["poses"].reshape(-1, 7), win.poses, f"poses it{it}")
        assert_parity(s["inv_depth"], win.inv_depth, f"inv_depth it{it}")
        assert_parity(s["vel"].reshape(-1, 3), win.vel, f"vel it{it}")
        assert_parity(s["ba"].reshape(-1, 3), win.ba, f"ba it{it}")
        assert_parity(s["bg"].reshape(-1, 3), win.bg, f"bg it{it}")
        radius, dec = ref["radius"], ref["decrease_factor"]
    for h in list(b.values()) + [st]:
        if h is not None:
            h.close()
    prob.close()


@pytest.mark.parametrize("n_kf,n_lm,seed,drop", [(12, 200, 11, (2, 3, 7)), (9, 100, 13, (0, 7)), (60, 300, 17, (10, 11, 30)),
                                                 (140, 600, 41, ())])      # top-level blocks too wide for the sparse path stay dense
def test_lm_iteration_parity_with_imu_gaps_and_large_windows(ctx, oracle, n_kf, n_lm, seed, drop):
    """The elimination plan is built from the actual IMU coupling graph (nested-dissection levels per chain, isolated blocks
    first); 60 keyframes also takes the Schur complement off the LDS-band path (ldE > 320) and deepens the level tree."""
    from lvio_fusion_amd import api
    cfg, st, b, prob, win = build(api, ctx, oracle, n_kf, n_lm, seed, imu_drop=drop)
    opt = api.default_solver_options()
    radius, dec = 1e4, 2.0
    for it in range(3):
        ref = win.lm_iteration(radius, dec)
        got = prob.lm_iteration(opt, radius, dec)
        assert abs(got["cost_before"] - ref["cost_before"]) <= 1e-8 * abs(ref["cost_before"])
        S, rhs = prob.reduced_system()
        assert np.abs(S - ref["S"]).max() <= 1e-7 * np.abs(ref["S"]).max(), f"iteration {it}: reduced system mismatch"
        assert got["accepted"] == ref["accepted"]
        assert abs(got["cost_after"] - ref["cost_after"]) <= 1e-6 * abs(ref["cost_after"])
        s = state_of(api, st)
        assert_parity(s["poses"].reshape(-1, 7), win.poses, f"poses it{it}")
        assert_parity(s["inv_depth"], win.inv_depth, f"inv_depth it{it}")
        assert_parity(s["vel"].reshape(-1, 3), win.vel, f"vel it{it}")
        assert_parity(s["ba"].reshape(-1, 3), win.ba, f"ba it{it}")
        assert_parity(s["bg"].reshape(-1, 3), win.bg, f"bg it{it}")
        radius, dec = ref["radius"], ref["decrease_factor"]
    for h in list(b.values()) + [st]:
        if h is not None:
            h.close()
    prob.close()


def test_lm_iteration_parity_with_landmark_ids_in_creation_order(ctx, oracle):
    """Ids handed out in birth order make whole waves of TwoFrame blocks share their first keyframe (the wave-reduction path of
    the linearisation) and give the band-limited Schur complement its natural ordering."""
    from lvio_fusion_amd import api
    cfg, st, b, prob, win = build(api, ctx, oracle, 8, 2400, 23, ids_by_birth=True)
    opt = api.default_solver_options()
    radius, dec = 1e4, 2.0
    for it in range(3):
        ref = win.lm_iteration(radius, dec)
        got = prob.lm_iteration(opt, radius, dec)
        assert abs(got["cost_before"] - ref["cost_before"]) <= 1e-8 * abs(ref["cost_before"])
        S, rhs = prob.reduced_system()
        assert np.abs(S - ref["S"]).max() <= 1e-7 * np.abs(ref["S"]).max(), f"iteration {it}: reduced system mismatch"
        assert_parity(rhs, ref["rhs"], f"rhs it{it}")
        assert got["accepted"] == ref["accepted"]
        s = state_of(api, st)
        assert_parity(s["poses"].reshape(-1, 7), win.poses, f"poses it{it}")
        assert_parity(s["inv_depth"], win.inv_depth, f"inv_depth it{it}")
        radius, dec = ref["radius"], ref["decrease_factor"]
    for h in list(b.values()) + [st]:
        if h is not None:
            h.close()
    prob.close()


def test_solve_reduces_cost_and_pose_error(ctx, oracle):
    from lvio_fusion_amd import api
    cfg, st, b, prob, win = build(api, ctx, oracle, 10, 200, 21)
    opt = api.default_solver_options()
    opt.max_num_iterations = 15
    e0 = np.abs(cfg["poses"] - cfg["poses_true"]).max()
    summ = prob.solve(opt)
    assert summ.final_cost < 0.2 * summ.initial_cost
    assert summ.num_successful_steps >= 3
    e1 = np.abs(st.get(api.POSES).reshape(-1, 7) - cfg["poses_true"]).max()
    assert e1 < 0.5 * e0
    # quaternions stay unit length under the EigenQuaternion plus operation
    q = st.get(api.POSES).reshape(-1, 7)[:, :4]
    assert np.allclose(np.linalg.norm(q, axis=1), 1.0, atol=1e-9)
    prob.close()


def test_constant_pose_is_not_moved(ctx, oracle):
    from lvio_fusion_amd import api
    cfg, st, b, prob, win = build(api, ctx, oracle, 6, 60, 33, use=("tc", "tf", "po"))
    prob.set_pose_constant(0, True)
    opt = api.default_solver_options()
    p0 = st.get(api.POSES).reshape(-1, 7).copy()
    r = prob.lm_iteration(opt, 1e4)
    assert r["accepted"]
    p1 = st.get(api.POSES).reshape(-1, 7)
    assert np.abs(p1[0] - p0[0]).max() < 1e-12 and np.abs(p1[1:] - p0[1:]).max() > 1e-6
    prob.close()


def test_unsorted_two_frame_uses_generic_path(ctx, oracle):
    """Blocks in arbitrary order (not sorted by current keyframe) go through the generic atomic linearisation and
    must give the same reduced system as the sorted fast path."""
    from lvio_fusion_amd import api
    cfg = syn.config4_window(n_kf=7, n_lm=90, n_prewindow=20, seed=41, imu_samples=4)
    perm = np.random.default_rng(0).permutation(len(cfg["tf"]["lm_idx"]))
    cfg2 = dict(cfg); cfg2["tf"] = {k: v[perm] for k, v in cfg["tf"].items()}
    pre = np.stack([oracle.imu_preintegrate(f["samples"], f["acc0"], f["gyr0"], f["ba"], f["bg"], syn.IMU_NOISE) for f in cfg["imu"]])
    out = []
    for c in (cfg, cfg2):
        st = api.State(ctx, c["n_kf"], c["n_lm"])
        for field, key in ((api.POSES, "poses"), (api.VEL, "vel"), (api.BA, "ba"), (api.BG, "bg"), (api.INV_DEPTH, "inv_depth"), (api.W_VISUAL, "w_kf")):
            st.set(field, c[key])
        tf = c["tf"]
        btf = api.two_frame_batch(ctx, c["cam0"], c["cam1"], tf["first_ob"], tf["ob"], tf["lm_idx"], tf["kf1_idx"], tf["kf2_idx"])
        prob = api.Problem(ctx, st, None, btf, None, None)
        r = prob.lm_iteration(api.default_solver_options(), 1e4)
        S, rhs = prob.reduced_system()
        out.append((r, S, rhs, st.get(api.POSES)))
        prob.close(); btf.close(); st.close()
    assert abs(out[0][0]["cost_before"] - out[1][0]["cost_before"]) <= 1e-10 * out[0][0]["cost_before"]
    assert np.abs(out[0][1] - out[1][1]).max() <= 1e-9 * np.abs(out[0][1]).max()
    assert_parity(out[1][3], out[0][3], "poses after one iteration")


def test_mixed_call_sequences_leave_no_stale_accumulators(ctx, oracle):
    """The device loop clears the normal-equation accumulators at the END of an iteration for the next one and remembers that they are
    clean; the stand-alone taps and the per-call API dirty them again.  Whatever the order of calls, an LM iteration must see exactly
    the accumulators of its own linearisation: every sequence below is checked against the oracle from the state it starts at."""
    from lvio_fusion_amd import api
    cfg, st, b, prob, win = build(api, ctx, oracle, 9, 200, 41)
    opt = api.default_solver_options()

    def check_iteration(tag):
        s = state_of(api, st)
        win.poses[:] = s["poses"].reshape(-1, 7); win.vel[:] = s["vel"].reshape(-1, 3); win.ba[:] = s["ba"].reshape(-1, 3)
        win.bg[:] = s["bg"].reshape(-1, 3); win.inv_depth[:] = s["inv_depth"]
        ref = win.lm_iteration(1e4, 2.0)
        got = prob.lm_iteration(opt, 1e4, 2.0)
        assert abs(got["cost_before"] - ref["cost_before"]) <= 1e-8 * abs(ref["cost_before"]), tag
        assert abs(got["cost_after"] - ref["cost_after"]) <= 1e-6 * abs(ref["cost_after"]), tag
        assert got["accepted"] == ref["accepted"], tag
        S, rhs = prob.reduced_system()
        assert np.abs(S - ref["S"]).max() <= 1e-7 * np.abs(ref["S"]).max(), tag

    o2 = api.default_solver_options(); o2.max_num_iterations = 2; o2.function_tolerance = 0.0; o2.parameter_tolerance = 0.0; o2.gradient_tolerance = 0.0
    check_iteration("fresh problem")
    prob.solve(o2)                                    # device loop: leaves the accumulators clean
    with pytest.raises(api.LvfError):
        prob.reduced_system()                         # ... and no linearisation to tap
    check_iteration("after a device-loop solve")
    prob.gradient(opt)                                # a stand-alone tap dirties them
    prob.solve(o2)
    check_iteration("after gradient + solve")
    prob.cost(opt)
    check_iteration("after cost")
    batch = api.ProblemBatch(ctx, [prob])
    batch.solve(o2)
    check_iteration("after a batch solve")
    batch.lm_iteration(opt, [1e4], [2.0])
    S, rhs = prob.reduced_system()                    # the per-call batch API keeps its normal equations too
    assert np.isfinite(S).all()
    check_iteration("after a batch iteration")
    batch.close()
    for h in [prob, st] + [x for x in b.values() if x is not None]:
        h.close()


def test_upper_triangle_of_the_reduced_system_is_never_read():
    """k_prepare assembles only the lower triangle of S.  With LVF_POISON_S=1 every byte of S is 0xff (NaN) before each assembly, so the
    upper triangle stays NaN through the Schur complement, the sparse levels, the dense factorisation and the back substitution: the
    per-iteration and trajectory parity cases must still pass (the switch is read once per process, hence the child process)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ); env["LVF_POISON_S"] = "1"
    for args in ([os.path.join(root, "tests", "test_gpu_solver.py"), "-k", "lm_iteration_parity"],
                 [os.path.join(root, "tests", "test_gpu_solve_trajectory.py"), "-k", "oracle_chain or rejected or batched_loop"]):
        p = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-x", "-p", "no:cacheprovider"] + args, cwd=root, env=env,
                           capture_output=True, text=True, timeout=900)
        assert p.returncode == 0 and " passed" in p.stdout, p.stdout[-3000:] + p.stderr[-2000:]


def test_constant_velocity_and_bias_blocks(ctx, oracle):
    """SetParameterBlockConstant on (v, ba, bg) blocks (Environment::Optimize, environment.cpp:62-68): per keyframe any subset of the three
    can be held; the ImuError factors keep their residuals, the blocks get no Jacobian columns, never move, and stay out of x_norm.
    Per-iteration parity and a device-loop trajectory against the oracle with the same masks."""
    from lvio_fusion_amd import api
    n_kf = 9
    cfg, st, b, prob, _ = build(api, ctx, oracle, n_kf, 200, 57)
    pre = np.stack([oracle.imu_preintegrate(f["samples"], f["acc0"], f["gyr0"], f["ba"], f["bg"], syn.IMU_NOISE) for f in cfg["imu"]])
    masks = np.array([7, 0, 1, 2, 4, 3, 0, 6, 5], np.uint8)        # bit 0 v, bit 1 ba, bit 2 bg
    pc = np.zeros(n_kf, np.uint8); pc[0] = 1
    win = oracle.Window(cfg, pre, pose_const=pc, vbb_const=masks)
    prob.set_pose_constant(0, True)
    for k, m in enumerate(masks):
        prob.set_vbb_constant(k, v=bool(m & 1), ba=bool(m & 2), bg=bool(m & 4))
    opt = api.default_solver_options()
    radius, dec = 1e4, 2.0
    for it in range(3):
        ref = win.lm_iteration(radius, dec)
        got = prob.lm_iteration(opt, radius, dec)
        S, rhs = prob.reduced_system()
        assert np.abs(S - ref["S"]).max() <= 1e-7 * np.abs(ref["S"]).max() and got["accepted"] == ref["accepted"]
        assert abs(got["cost_after"] - ref["cost_after"]) <= 1e-6 * abs(ref["cost_after"])
        s = state_of(api, st)
        for name in ("poses", "vel", "ba", "bg", "inv_depth"):
            assert_parity(np.asarray(s[name]).reshape(np.asarray(getattr(win, name)).shape), getattr(win, name), f"{name} it{it}")
        radius, dec = ref["radius"], ref["decrease_factor"]
    s = state_of(api, st)
    for k, m in enumerate(masks):          # held blocks are bit-for-bit where they started
        for bit, name in ((1, "vel"), (2, "ba"), (4, "bg")):
            if m & bit:
                assert np.array_equal(np.asarray(s[name]).reshape(-1, 3)[k], cfg[name][k]), (k, name)
    assert np.array_equal(np.asarray(s["poses"]).reshape(-1, 7)[0], cfg["poses"][0])
    o = api.default_solver_options(); o.max_num_iterations = 8
    ref = win.solve(max_num_iterations=8)
    summ = prob.solve(o)
    assert (summ.num_iterations, summ.num_successful_steps, summ.why) == (ref["num_iterations"], ref["num_successful_steps"], ref["why"])
    assert abs(summ.final_cost - ref["final_cost"]) <= 1e-6 * ref["final_cost"]
    prob.close()
    for h in list(b.values()) + [st]:
        if h is not None:
            h.close()


@pytest.mark.parametrize("n_kf,n_lm,seed,drop", [(16, 400, 51, ()), (24, 500, 53, ()), (24, 500, 55, (5, 6, 17)), (33, 600, 57, ()), (48, 800, 59, ())])
def test_sparse_level_placements(ctx, oracle, n_kf, n_lm, seed, drop):
    """The (v, ba, bg) levels ride in different launches depending on how many there are (three: all chained inside the Schur launch;
    four / five: the first one / two ride in the k_tf_reduce / k_prepare launches): three LM iterations at window sizes on either side
    of every switch, with the damped-system tap (classic assembly) between them."""
    from lvio_fusion_amd import api
    cfg, st, b, prob, win = build(api, ctx, oracle, n_kf, n_lm, seed, imu_drop=drop)
    opt = api.default_solver_options()
    radius, dec = 1e4, 2.0
    for it in range(3):
        ref = win.lm_iteration(radius, dec)
        got = prob.lm_iteration(opt, radius, dec)
        assert abs(got["cost_before"] - ref["cost_before"]) <= 1e-8 * abs(ref["cost_before"])
        if it == 1:
            S, rhs = prob.reduced_system()
            assert np.abs(S - ref["S"]).max() <= 1e-7 * np.abs(ref["S"]).max()
        assert got["accepted"] == ref["accepted"]
        assert abs(got["cost_after"] - ref["cost_after"]) <= 1e-6 * abs(ref["cost_after"])
        s = state_of(api, st)
        assert_parity(s["poses"].reshape(-1, 7), win.poses, f"poses it{it}")
        assert_parity(s["vel"].reshape(-1, 3), win.vel, f"vel it{it}")
        assert_parity(s["bg"].reshape(-1, 3), win.bg, f"bg it{it}")
        radius, dec = ref["radius"], ref["decrease_factor"]
    for h in list(b.values()) + [st]:
        if h is not None:
            h.close()
    prob.close()


def test_chained_levels_give_the_same_answer_every_time(ctx, oracle):
    """Levels chained inside one launch hand their updates over through memory while the launch runs.  The case that exposed a hand-over
    race (8 keyframes, 20 000 landmarks: a long Schur launch around a three-level chain; 5 of 12 runs were wrong before the updates
    the next level reads became returning atomics) is solved 25 times from the same start: every run must land on the same cost."""
    from lvio_fusion_amd import api
    cfg, st, b, prob, win = build(api, ctx, oracle, 8, 20000, 2020, n_pre=0)
    opt = api.default_solver_options()
    opt.max_num_iterations = 2
    opt.function_tolerance = 0.0; opt.parameter_tolerance = 0.0; opt.gradient_tolerance = 0.0
    finals = []
    for run in range(25):
        for field, key in ((api.POSES, "poses"), (api.VEL, "vel"), (api.BA, "ba"), (api.BG, "bg"), (api.INV_DEPTH, "inv_depth")):
            st.set(field, cfg[key])
        finals.append(prob.solve(opt).final_cost)
    finals = np.array(finals)
    assert np.all(np.isfinite(finals))
    assert np.ptp(finals) <= 1e-9 * abs(finals[0]), f"runs disagree: spread {np.ptp(finals):.3e} on {finals[0]:.6e}"
    for h in list(b.values()) + [st]:
        if h is not None:
            h.close()
    prob.close()
