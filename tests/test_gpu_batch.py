"""GPU parity of the batched multi-window solver (lvf_problem_batch_*): W independent windows advanced by one chain of launches per
LM iteration (blockIdx.y = window, per-window argument tables, device-resident accept / reject and termination) must give, per window,
exactly what the single-window path gives on that window alone — and the single-window path is pinned to oracle/lm.h in
tests/test_gpu_solver.py and tests/test_gpu_baseline_sizes.py."""
import numpy as np
import pytest

from lvio_fusion_amd import synthetic as syn
from tests.helpers import assert_parity
from tests.test_gpu_solver import build, state_of

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from lvio_fusion_amd import api
    c = api.Context(0)
    yield c
    c.close()


SHAPES = [(8, 120, 3), (12, 300, 5), (6, 80, 7), (12, 200, 11), (9, 100, 13), (8, 2400, 23)]     # different sizes, level trees and block counts


def _windows(api, ctx, oracle, shapes, **kw):
    out = []
    for (n_kf, n_lm, seed) in shapes:
        cfg, st, b, prob, win = build(api, ctx, oracle, n_kf, n_lm, seed, **kw)
        out.append(dict(cfg=cfg, st=st, b=b, prob=prob, win=win))
    return out


def _close(ws):
    for w in ws:
        w["prob"].close()
        for h in list(w["b"].values()) + [w["st"]]:
            if h is not None:
                h.close()


def test_batch_lm_iterations_match_the_oracle_per_window(ctx, oracle):
    from lvio_fusion_amd import api
    ws = _windows(api, ctx, oracle, SHAPES)
    batch = api.ProblemBatch(ctx, [w["prob"] for w in ws])
    opt = api.default_solver_options()
    assert batch.uses_tables(opt) == 1
    radius, dec = np.full(len(ws), 1e4), np.full(len(ws), 2.0)
    for it in range(4):
        got = batch.lm_iteration(opt, radius, dec)
        for i, w in enumerate(ws):
            ref = w["win"].lm_iteration(radius[i], dec[i])
            g = got[i]
            assert abs(g["cost_before"] - ref["cost_before"]) <= 1e-8 * abs(ref["cost_before"]), (it, i)
            assert g["accepted"] == ref["accepted"], (it, i)
            assert abs(g["cost_after"] - ref["cost_after"]) <= 1e-6 * abs(ref["cost_after"]), (it, i)
            assert abs(g["radius"] - ref["radius"]) <= 1e-5 * ref["radius"], (it, i)
            s = state_of(api, w["st"])
            assert_parity(s["poses"].reshape(-1, 7), w["win"].poses, f"poses it{it} window {i}")
            assert_parity(s["inv_depth"], w["win"].inv_depth, f"inv_depth it{it} window {i}")
            assert_parity(s["vel"].reshape(-1, 3), w["win"].vel, f"vel it{it} window {i}")
            radius[i], dec[i] = ref["radius"], ref["decrease_factor"]
            if it == 1 and i == 2:      # the taps of one member still work inside a batch
                S, rhs = w["prob"].reduced_system()
                assert np.abs(S - ref["S"]).max() <= 1e-7 * np.abs(ref["S"]).max()
    batch.close(); _close(ws)


def test_batch_solve_equals_single_window_solves(ctx, oracle):
    """Full device LM loops: per-window termination (windows finish after different numbers of iterations), identical summaries and states."""
    from lvio_fusion_amd import api
    shapes = SHAPES[:5]
    a = _windows(api, ctx, oracle, shapes)
    b = _windows(api, ctx, oracle, shapes)
    opt = api.default_solver_options(); opt.max_num_iterations = 12
    single = [w["prob"].solve(opt) for w in a]
    batch = api.ProblemBatch(ctx, [w["prob"] for w in b])
    multi = batch.solve(opt)
    its = set()
    for i, (s1, s2) in enumerate(zip(single, multi)):
        assert (s1.num_iterations, s1.num_successful_steps, s1.termination, s1.num_residual_blocks) == (s2.num_iterations, s2.num_successful_steps, s2.termination, s2.num_residual_blocks), i
        assert abs(s1.initial_cost - s2.initial_cost) <= 1e-12 * s1.initial_cost and abs(s1.final_cost - s2.final_cost) <= 1e-9 * s1.final_cost
        for k, f in (("poses", api.POSES), ("inv_depth", api.INV_DEPTH), ("vel", api.VEL), ("ba", api.BA), ("bg", api.BG)):
            assert_parity(b[i]["st"].get(f), a[i]["st"].get(f), f"{k} window {i}")
        assert s1.final_cost < s1.initial_cost
        its.add(s1.num_iterations)
    batch.close(); _close(a); _close(b)


def test_batch_falls_back_when_a_window_has_no_table_form(ctx, oracle):
    """A window without IMU blocks (or with pose priors) has no table form: the batch still solves every window correctly."""
    from lvio_fusion_amd import api
    a = _windows(api, ctx, oracle, [(8, 120, 3)]) + _windows(api, ctx, oracle, [(6, 80, 7)], use=("tc", "tf", "po"))
    b = _windows(api, ctx, oracle, [(8, 120, 3)]) + _windows(api, ctx, oracle, [(6, 80, 7)], use=("tc", "tf", "po"))
    opt = api.default_solver_options(); opt.max_num_iterations = 5
    batch = api.ProblemBatch(ctx, [w["prob"] for w in b])
    assert batch.uses_tables(opt) == 0
    multi = batch.solve(opt)
    for i, w in enumerate(a):
        s1 = w["prob"].solve(opt)
        assert (s1.num_iterations, s1.num_successful_steps) == (multi[i].num_iterations, multi[i].num_successful_steps)
        assert_parity(b[i]["st"].get(api.POSES), w["st"].get(api.POSES), f"poses window {i}")
    batch.close(); _close(a); _close(b)
    with pytest.raises(api.LvfError):
        api.ProblemBatch(ctx, [])


def test_solve_termination_rules(ctx, oracle):
    """Device-side termination: max_num_iterations caps the loop (NO_CONVERGENCE), a converged problem stops on the function tolerance,
    max_num_iterations = 0 only reports the cost, and a second solve from the converged state stops immediately."""
    from lvio_fusion_amd import api
    cfg, st, b, prob, win = build(api, ctx, oracle, 8, 150, 77)
    opt = api.default_solver_options()
    opt.max_num_iterations = 0
    s = prob.solve(opt)
    assert s.num_iterations == 0 and abs(s.initial_cost - win.cost()) <= 1e-9 * s.initial_cost and s.final_cost == s.initial_cost
    opt.max_num_iterations = 2
    s = prob.solve(opt)
    assert s.num_iterations == 2 and s.termination == 1
    opt.max_num_iterations = 50
    s = prob.solve(opt)
    assert s.termination == 0 and s.num_iterations < 50 and s.final_cost <= s.initial_cost
    s2 = prob.solve(opt)
    assert s2.termination == 0 and s2.num_iterations <= 3 and abs(s2.final_cost - s.final_cost) <= 1e-5 * s.final_cost
    prob.close()
    for h in list(b.values()) + [st]:
        if h is not None:
            h.close()
