"""The in-launch hand-over between chained sparse levels (csrc/solver_kernels.hip, SpSrc) must never turn a scheduling delay into a
numerical outcome: a consumer that does not see its producers in time raises a flag, the device loop stops WITHOUT judging the step, and
the host re-runs the iteration with un-chained launches (lvf_solver_summary::hand_over_retries).  Exercised three ways: a forced
time-out on an idle GPU (single window, per-call API, batch), the configs[3] solve under a second context's traffic — the reference runs
Backend::Optimize and Relocator -> Mapping::Relocate concurrently (src/lvio_fusion/src/relocator.cpp:188,210; backend.cpp:32) — and
the batch's promise to leave its members alone (solved alone -> in a batch -> alone: first and third results identical)."""
import threading

import numpy as np
import pytest

from lvio_fusion_amd import synthetic as syn
from tests.test_gpu_solver import build, state_of

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from lvio_fusion_amd import api
    c = api.Context(0)
    yield c
    c.close()


def fixed(api, n):
    o = api.default_solver_options()
    o.max_num_iterations = n; o.function_tolerance = 0.0; o.parameter_tolerance = 0.0; o.gradient_tolerance = 0.0
    return o


def reset(api, st, cfg):
    for field, key in ((api.POSES, "poses"), (api.VEL, "vel"), (api.BA, "ba"), (api.BG, "bg"), (api.INV_DEPTH, "inv_depth")):
        st.set(field, cfg[key])


def close_all(prob, b, st):
    prob.close()
    for h in list(b.values()) + [st]:
        if h is not None:
            h.close()


def same_state(a, b, rtol=1e-9):
    for k in a:
        scale = np.abs(a[k]).max() + 1e-300
        assert np.abs(a[k] - b[k]).max() <= rtol * scale, f"{k}: {np.abs(a[k] - b[k]).max():.3e} on {scale:.3e}"


@pytest.mark.parametrize("n_kf,n_lm,seed", [(20, 4000, 77), (8, 20000, 2020)])
def test_forced_timeout_is_retried_unchained(ctx, oracle, n_kf, n_lm, seed):
    """A producer that never arrives: the loop must report the retry, take the same number of iterations and land where the undisturbed
    solve lands (to summation order: the un-chained launches add into S in a different order)."""
    from lvio_fusion_amd import api
    cfg, st, b, prob, win = build(api, ctx, oracle, n_kf, n_lm, seed, n_pre=0)
    opt = fixed(api, 6)
    ref = prob.solve(opt)
    x_ref = state_of(api, st)
    assert ref.hand_over_retries == 0 and ref.num_iterations == 6
    reset(api, st, cfg)
    prob.debug_force_handover_timeout(1)
    got = prob.solve(opt)
    assert got.hand_over_retries == 1, "the forced time-out was not seen (are the levels chained at this size?)"
    assert got.num_iterations == ref.num_iterations and got.num_successful_steps == ref.num_successful_steps and got.termination_reason == ref.termination_reason
    assert abs(got.final_cost - ref.final_cost) <= 1e-9 * abs(ref.final_cost)
    same_state(x_ref, state_of(api, st))
    # the problem stays un-chained afterwards and keeps giving the same answer; the per-call API takes the same path
    reset(api, st, cfg)
    again = prob.solve(opt)
    assert again.hand_over_retries == 1 and abs(again.final_cost - ref.final_cost) <= 1e-9 * abs(ref.final_cost)
    # chaining comes back after 32 un-chained solves (one scheduling blip must not cost a persistent window its chained launches for
    # good); the answers do not move across the switch
    for k in range(36):
        reset(api, st, cfg)
        s = prob.solve(opt)
        assert s.hand_over_retries == 1 and s.num_iterations == ref.num_iterations and abs(s.final_cost - ref.final_cost) <= 1e-9 * abs(ref.final_cost), k
    reset(api, st, cfg)
    o1 = prob.lm_iteration(opt, 1e4, 2.0)
    reset(api, st, cfg)
    prob.debug_force_handover_timeout(1)
    o2 = prob.lm_iteration(opt, 1e4, 2.0)
    assert o1["accepted"] == o2["accepted"] and abs(o1["cost_after"] - o2["cost_after"]) <= 1e-9 * abs(o1["cost_after"]) and o1["radius"] == pytest.approx(o2["radius"], rel=1e-6)
    close_all(prob, b, st)


def test_forced_timeout_of_the_pose_hand_over(ctx, oracle):
    """The merged back substitution + step tail (k_backsolve_tail): the landmark workgroups wait INSIDE the launch for the pose increments.
    n = 2 makes that wait (and the chained level's) give up: the pass must be re-run with separate launches and land where the undisturbed
    solve lands."""
    from lvio_fusion_amd import api
    cfg, st, b, prob, win = build(api, ctx, oracle, 20, 4000, 78, n_pre=0)
    opt = fixed(api, 5)
    ref = prob.solve(opt)
    x_ref = state_of(api, st)
    reset(api, st, cfg)
    prob.debug_force_handover_timeout(2)
    got = prob.solve(opt)
    assert got.hand_over_retries >= 1 and got.num_iterations == ref.num_iterations and got.num_successful_steps == ref.num_successful_steps
    assert abs(got.final_cost - ref.final_cost) <= 1e-9 * abs(ref.final_cost)
    same_state(x_ref, state_of(api, st))
    close_all(prob, b, st)


def test_forced_timeout_in_a_batch(ctx, oracle):
    """One window of three times out: it alone is re-run, the other two are untouched; every window equals its single solve."""
    from lvio_fusion_amd import api
    wins = [build(api, ctx, oracle, 20, 3000, 500 + i, n_pre=0) for i in range(3)]
    opt = fixed(api, 5)
    singles = []
    for cfg, st, b, prob, _ in wins:
        singles.append((prob.solve(opt).final_cost, state_of(api, st)))
        reset(api, st, cfg)
    batch = api.ProblemBatch(ctx, [w[3] for w in wins])
    assert batch.uses_tables(opt) == 1
    wins[1][3].debug_force_handover_timeout(1)
    ss = batch.solve(opt)
    assert [s.hand_over_retries for s in ss] == [0, 1, 0]
    for (cost, x), s, (cfg, st, b, prob, _) in zip(singles, ss, wins):
        assert s.num_iterations == 5 and abs(s.final_cost - cost) <= 1e-9 * abs(cost)
        same_state(x, state_of(api, st))
    batch.close()
    for cfg, st, b, prob, _ in wins:
        close_all(prob, b, st)


def test_configs3_solve_under_a_second_contexts_traffic(oracle):
    """60 x prob.solve(20) of the BASELINE window on one context while a second host thread keeps another context of the same GPU busy
    (loop-closure candidates and chip-filling associations): every solve must equal the serial one to 1e-9 — same iteration and step
    counts, same cost, same state — whether or not a hand-over had to be retried."""
    from lvio_fusion_amd import api, relocalize as rl
    ctx_a = api.Context(0)
    cfg = syn.config4_window()
    pre = api.preintegrate_or_none(ctx_a, cfg)
    st = api.State(ctx_a, cfg["n_kf"], cfg["n_lm"])
    for field, key in ((api.POSES, "poses"), (api.VEL, "vel"), (api.BA, "ba"), (api.BG, "bg"), (api.INV_DEPTH, "inv_depth"), (api.W_VISUAL, "w_kf")):
        st.set(field, cfg[key])
    tc, tf, po = cfg["tc"], cfg["tf"], cfg["po"]
    hs = [api.two_camera_batch(ctx_a, cfg["cam0"], cfg["cam1"], tc["left_ob"], tc["right_ob"], tc["lm_idx"], tc["kf_idx"]),
          api.two_frame_batch(ctx_a, cfg["cam0"], cfg["cam1"], tf["first_ob"], tf["ob"], tf["lm_idx"], tf["kf1_idx"], tf["kf2_idx"]),
          api.pose_only_batch(ctx_a, cfg["cam0"], po["ob"], po["kf_idx"], po["pw_idx"], po["pw"]),
          api.imu_batch(ctx_a, pre, [f["kf_i"] for f in cfg["imu"]], [f["kf_j"] for f in cfg["imu"]])]
    prob = api.Problem(ctx_a, st, *hs)
    opt = fixed(api, 20)
    serial = prob.solve(opt)
    x_serial = state_of(api, st)
    assert serial.hand_over_retries == 0
    cands = syn.config5_candidates(2, seed=8, n_query=100000, n_az=2900, overlap="full")       # full-size scans: launches that fill the chip
    stop, err, laps = threading.Event(), [], [0]

    def traffic():
        # two kinds of load, alternating: whole candidate evaluations (uploads, index builds, many small launches) and back-to-back
        # chip-filling associations (3 125 workgroups each) — with the latter a workgroup of the solver's launches can be dispatched
        # many microseconds after its siblings, which is what exposed the in-place store of the factored diagonal block (round 4:
        # 197 of 400 solves took a step as invalid before the block went to a side buffer)
        c2 = api.Context(0)
        try:
            c = cands[0]
            mp, sc = api.Map(c2, c["map"], 4.0), api.Scan(c2, c["query"])
            while not stop.is_set():
                if laps[0] % 2 == 0:
                    rl.evaluate_candidate(api, c2, cands[(laps[0] // 2) % 2])
                else:
                    for _ in range(40):
                        api.knn3(mp, sc, c["init_pose"], 4.0)
                    c2.synchronize()
                laps[0] += 1
            mp.close(); sc.close()
        except Exception as e:
            err.append(e)
        finally:
            c2.close()
    t = threading.Thread(target=traffic)
    t.start()
    try:
        retries = 0
        for run in range(60):
            reset(api, st, cfg)
            s = prob.solve(opt)
            assert s.termination_reason == serial.termination_reason and s.num_iterations == serial.num_iterations and s.num_successful_steps == serial.num_successful_steps, (run, s.why, s.num_iterations, s.num_successful_steps)
            assert abs(s.final_cost - serial.final_cost) <= 1e-9 * abs(serial.final_cost), (run, s.final_cost, serial.final_cost)
            same_state(x_serial, state_of(api, st))
            retries = s.hand_over_retries
    finally:
        stop.set()
        t.join(timeout=120)
    assert not t.is_alive() and not err, err
    assert laps[0] > 0, "the second context did no work while the solves ran"
    print(f"hand-over retries over 60 solves under traffic: {retries}; traffic laps meanwhile: {laps[0]}")
    prob.close()
    for h in hs + [st]:
        h.close()
    ctx_a.close()


def test_a_batch_leaves_its_members_alone(ctx, oracle):
    """alone -> in a batch -> alone: the first and third results agree to the run-to-run spread of one chain (the batch's wider Schur
    slices are its own: the member's chain is never rebuilt), the batched one agrees to summation order; destroying a member before its batch orphans the batch instead of leaving it a dangling pointer."""
    from lvio_fusion_amd import api
    wins = [build(api, ctx, oracle, 12, 1500, 900 + i, n_pre=20) for i in range(2)]
    opt = fixed(api, 4)
    cfg, st, b, prob, _ = wins[0]
    first = prob.solve(opt); x1 = state_of(api, st)
    for w in wins:
        reset(api, w[1], w[0])
    batch = api.ProblemBatch(ctx, [w[3] for w in wins])
    ss = batch.solve(opt); x2 = state_of(api, st)
    reset(api, st, cfg)
    third = prob.solve(opt); x3 = state_of(api, st)           # the batch still exists: it must not have changed the member's chain
    # (not bit-identical: the Schur complement's output adds are floating-point atomics whose order varies from run to run of the SAME chain;
    # 1e-12 is that run-to-run spread, two orders below what a change of slice width moves)
    assert abs(first.final_cost - third.final_cost) <= 1e-12 * abs(first.final_cost)
    same_state(x1, x3, rtol=1e-11)
    assert abs(ss[0].final_cost - first.final_cost) <= 1e-9 * abs(first.final_cost)
    same_state(x1, x2)
    wins[1][3].close()                                         # a member dies first
    with pytest.raises(api.LvfError):
        batch.solve(opt)
    batch.close()
    reset(api, st, cfg)
    assert abs(prob.solve(opt).final_cost - first.final_cost) <= 1e-12 * abs(first.final_cost)      # the survivor is unaffected
    close_all(prob, b, st)
    for h in list(wins[1][2].values()) + [wins[1][1]]:
        if h is not None:
            h.close()
