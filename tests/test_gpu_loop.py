"""GPU parity of the loop-correction tail: RelocateRError<7,4>, Relocator::UpdateNewSubmap's rotation solve and
PoseGraph::ForwardUpdate, each through the C-ABI against the oracle (oracle/factors.h RelocateRResidual — itself pinned bit-for-bit to
the reference's pose_error.hpp:192-222 in tests/test_oracle_ref.py — and oracle/loop.h)."""
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


def submap(n, seed, noise=1e-3, angle=(0.05, -0.02, 0.03)):
    """n keyframes of a new submap: unrelocated_i = base^-1 * pose_i, relocated_i = [R0, 0] * unrelocated_i + noise."""
    rng = np.random.default_rng(seed)
    un = np.zeros((n, 7)); un[:, :4] = syn.quat_from_ypr(*rng.normal(0, 0.3, (3, n))); un[:, 4:] = rng.normal(0, 5, (n, 3))
    R = np.concatenate([syn.quat_from_ypr(*angle), [0, 0, 0]])
    rel = syn.se3_mul(np.tile(R, (n, 1)), un) + rng.normal(0, noise, (n, 7))
    return rel, un, R[:4]


@pytest.mark.parametrize("n,seed", [(1, 1), (12, 2), (300, 3)])
def test_relocate_r_error_parity(ctx, oracle, n, seed):
    from lvio_fusion_amd import api
    rel, un, _ = submap(n, seed)
    for q in (np.array([0, 0, 0, 1.0]), np.array([0.07, -0.11, 0.2, 1.3])):          # identity (the reference's start) and a non-unit quaternion
        r, J = api.relocate_r_evaluate(ctx, rel, un, q)
        for i in range(n):
            r0, J0 = oracle.relocate_r(rel[i], un[i], q)
            assert_parity(r[i], r0, f"RelocateRError r[{i}]"); assert_parity(J[i], J0, f"RelocateRError J[{i}]")
        r2, none = api.relocate_r_evaluate(ctx, rel, un, q, jacobians=False)
        assert none is None and np.array_equal(r2, r)


@pytest.mark.parametrize("n,seed,noise,angle", [(12, 5, 1e-3, (0.05, -0.02, 0.03)), (40, 6, 0.05, (0.6, -0.3, 0.2)), (3, 7, 0.0, (0.0, 0.0, 0.0)),
                                                 (1, 8, 1e-2, (-0.2, 0.1, 0.4))])
def test_relocate_rotation_solve_matches_oracle(ctx, oracle, n, seed, noise, angle):
    from lvio_fusion_amd import api
    rel, un, q_true = submap(n, seed, noise, angle)
    q0 = np.array([0, 0, 0, 1.0])
    q_ref, s_ref = oracle.relocate_rotation_solve(rel, un, q0)
    q, s = api.relocate_rotation_solve(ctx, rel, un, q0)
    assert (s.num_iterations, s.num_successful_steps, s.termination) == (s_ref["num_iterations"], s_ref["num_successful_steps"], s_ref["termination"])
    assert abs(s.initial_cost - s_ref["initial_cost"]) <= 1e-9 * max(s_ref["initial_cost"], 1e-30)
    assert abs(s.final_cost - s_ref["final_cost"]) <= 1e-6 * max(s_ref["final_cost"], 1e-12) + 1e-15
    assert np.abs(q - q_ref).max() <= 1e-9
    if noise <= 1e-3:
        assert np.abs(q * np.sign(q[3]) - q_true).max() < 5e-3
    assert np.array_equal(q0, [0, 0, 0, 1.0])      # the caller's array is only touched through the returned copy


def test_relocate_rotation_solve_one_iteration_and_empty(ctx, oracle):
    from lvio_fusion_amd import api
    rel, un, _ = submap(9, 11, 0.02, (0.3, 0.1, -0.2))
    opt = api.default_solver_options(); opt.max_num_iterations = 1
    q_ref, s_ref = oracle.relocate_rotation_solve(rel, un, [0, 0, 0, 1.0], max_iters=1)
    q, s = api.relocate_rotation_solve(ctx, rel, un, [0, 0, 0, 1.0], opt)
    assert s.num_iterations == 1 == s_ref["num_iterations"] and np.abs(q - q_ref).max() <= 1e-10
    q, s = api.relocate_rotation_solve(ctx, np.zeros((0, 7)), np.zeros((0, 7)), [0, 0, 0, 1.0])
    assert s.num_iterations == 0 and np.array_equal(q, [0, 0, 0, 1.0])
    with pytest.raises(api.LvfError):
        api.relocate_rotation_solve(ctx, rel, un, [0, 0, 0, 0.0])


@pytest.mark.parametrize("n", [1, 50, 1000])
def test_forward_update_parity(ctx, oracle, n):
    from lvio_fusion_amd import api
    rng = np.random.default_rng(n)
    poses = np.zeros((n, 7)); poses[:, :4] = syn.quat_from_ypr(*rng.normal(0, 1.0, (3, n))); poses[:, 4:] = rng.normal(0, 30, (n, 3))
    vw = rng.normal(0, 3, (n, 3))
    T = np.concatenate([syn.quat_from_ypr(0.4, -0.1, 0.05) * 1.0000001, [3.0, -2.0, 0.5]])      # Sophus keeps it unit up to rounding
    P0, V0 = oracle.forward_update(T, poses, vw)
    P, V = api.forward_update(ctx, T, poses, vw)
    assert_parity(P, P0, "ForwardUpdate poses"); assert_parity(V, V0, "ForwardUpdate Vw")
    assert np.allclose(np.linalg.norm(P[:, :4], axis=1), 1.0, atol=1e-14)
    P2, none = api.forward_update(ctx, T, poses)
    assert none is None and np.array_equal(P2, P)
    # device-resident state: keyframes [first, n) only
    st = api.State(ctx, n, 0)
    st.set(api.POSES, poses); st.set(api.VEL, vw)
    first = n // 3
    ctx.L.lvf_state_forward_update(st.h, T.ctypes.data_as(api._lib.c_double_p), first)
    Ps, Vs = st.get(api.POSES).reshape(-1, 7), st.get(api.VEL).reshape(-1, 3)
    assert np.array_equal(Ps[:first], poses[:first]) and np.array_equal(Vs[:first], vw[:first])
    assert_parity(Ps[first:], P0[first:], "state ForwardUpdate poses"); assert_parity(Vs[first:], V0[first:], "state ForwardUpdate Vw")
    st.close()
