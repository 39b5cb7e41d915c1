"""GPU parity of the batched Problem::Evaluate surface: robustified residuals, LOCAL-coordinate Jacobians per batch and
the gradient J^T r, against the same quantities assembled in numpy from the oracle's per-factor ambient outputs (Huber corrector and
EigenQuaternionParameterization restated in tests/test_oracle_lm_numpy.py)."""
import numpy as np
import pytest

from lvio_fusion_amd import synthetic as syn
from tests.helpers import assert_parity
from tests.test_gpu_solver import build
from tests.test_oracle_lm_numpy import dense_system, huber_scale, to_local

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from lvio_fusion_amd import api
    c = api.Context(0)
    yield c
    c.close()


def test_local_jacobians_and_gradient(ctx, oracle):
    from lvio_fusion_amd import api
    n_kf, n_lm = 7, 120
    cfg, st, b, prob, win = build(api, ctx, oracle, n_kf, n_lm, 77)
    pre = np.stack([oracle.imu_preintegrate(f["samples"], f["acc0"], f["gyr0"], f["ba"], f["bg"], syn.IMU_NOISE) for f in cfg["imu"]])
    state = [np.asarray(cfg[k], dtype=np.float64) for k in ("poses", "vel", "ba", "bg", "inv_depth")]
    J, r, cost = dense_system(oracle, cfg, pre, state)        # rows: tc, tf, po, imu blocks in batch order
    g = J.T @ r
    opt = api.default_solver_options()
    gc, gl = prob.gradient(opt)
    assert_parity(gc, g[:15 * n_kf], "gradient (keyframe part)"); assert_parity(gl, g[15 * n_kf:], "gradient (inverse depths)")
    assert abs(prob.cost(opt) - cost) <= 1e-10 * cost
    # per-batch local Jacobians against the rows of the dense matrix
    tc, tf, po = cfg["tc"], cfg["tf"], cfg["po"]
    row = 0
    rr, Jl = b["tc"].evaluate_local(st, 1.0)
    for i in range(b["tc"].n):
        assert_parity(rr[i], r[row:row + 2], "tc r"); assert_parity(Jl[i][:, 0], J[row:row + 2, 15 * n_kf + tc["lm_idx"][i]], "tc J"); row += 2
    rr, Jl = b["tf"].evaluate_local(st, 1.0)
    assert Jl.shape[2] == 13
    for i in range(b["tf"].n):
        k1, k2, l = tf["kf1_idx"][i], tf["kf2_idx"][i], tf["lm_idx"][i]
        ref = np.concatenate([J[row:row + 2, [15 * n_kf + l]], J[row:row + 2, 6 * k1:6 * k1 + 6], J[row:row + 2, 6 * k2:6 * k2 + 6]], axis=1)
        assert_parity(rr[i], r[row:row + 2], "tf r"); assert_parity(Jl[i], ref, "tf J"); row += 2
    rr, Jl = b["po"].evaluate_local(st, 1.0)
    for i in range(b["po"].n):
        k = po["kf_idx"][i]
        assert_parity(rr[i], r[row:row + 2], "po r"); assert_parity(Jl[i], J[row:row + 2, 6 * k:6 * k + 6], "po J"); row += 2
    rr, Jl = b["imu"].evaluate_local(st, 1.0)
    assert Jl.shape[1:] == (15, 30)
    for f, fac in enumerate(cfg["imu"]):
        ki, kj = fac["kf_i"], fac["kf_j"]
        oi, oj = 6 * n_kf + 9 * ki, 6 * n_kf + 9 * kj
        ref = np.concatenate([J[row:row + 15, 6 * ki:6 * ki + 6], J[row:row + 15, oi:oi + 9], J[row:row + 15, 6 * kj:6 * kj + 6], J[row:row + 15, oj:oj + 9]], axis=1)
        assert_parity(rr[f], r[row:row + 15], "imu r"); assert_parity(Jl[f], ref, "imu J"); row += 15
    assert row == len(r)
    # apply_loss_function = false: plain residuals, unscaled local Jacobians
    r_raw, J_raw = b["po"].evaluate_local(st, 0.0)
    b["po"].evaluate(st)
    assert_parity(r_raw, b["po"].residuals(), "po raw r")
    Ja = b["po"].jacobian(0)
    for i in range(0, b["po"].n, 7):
        assert_parity(J_raw[i], to_local(Ja[i], state[0][po["kf_idx"][i]]), "po raw J")
    # residuals only
    r_only, none = b["tf"].evaluate_local(st, 1.0, jacobians=False)
    assert none is None
    for h in list(b.values()) + [st]:
        if h is not None:
            h.close()
    prob.close()


def test_prior_batch_local_and_two_camera_block_weights(ctx, oracle):
    from lvio_fusion_amd import api
    cfg = syn.config4_window(n_kf=5, n_lm=60, n_prewindow=0, seed=91, imu_samples=3)
    st = api.State(ctx, 5, 60)
    for field, key in ((api.POSES, "poses"), (api.INV_DEPTH, "inv_depth"), (api.W_VISUAL, "w_kf")):
        st.set(field, cfg[key])
    tgt = np.zeros((3, 7)); tgt[0] = cfg["poses_true"][0]; tgt[1, :6] = oracle.pose_graph_target(cfg["poses_true"][1], cfg["poses_true"][2]); tgt[2] = cfg["poses_true"][3]
    bp = api.pose_prior_batch(ctx, [-1, 1, -2], [0, 2, 3], tgt, [100.0, 100.0, 3.0], [0.0, 0.5, 0.0])
    r, J = bp.evaluate_local(st, 1.0)          # priors carry no loss: huber is ignored
    assert J.shape == (3, 6, 12)
    r0, J0 = oracle.pose_prior(tgt[0], 100.0, 0.0, cfg["poses"][0])
    assert_parity(r[0], r0, "PoseError r"); assert np.all(J[0][:, :6] == 0); assert_parity(J[0][:, 6:], to_local(J0, cfg["poses"][0]), "PoseError J")
    r1, Ja, Jb = oracle.pose_graph(tgt[1, :6], 100.0, 0.5, cfg["poses"][1], cfg["poses"][2])
    assert_parity(r[1], r1, "PoseGraphError r"); assert_parity(J[1][:, :6], to_local(Ja, cfg["poses"][1]), "PoseGraphError Ja"); assert_parity(J[1][:, 6:], to_local(Jb, cfg["poses"][2]), "PoseGraphError Jb")
    r2, J2 = oracle.r_error(tgt[2], 3.0, cfg["poses"][3])
    assert_parity(r[2][:4], r2, "RError r"); assert np.all(r[2][4:] == 0); assert_parity(J[2][:4, 6:], to_local(J2, cfg["poses"][3]), "RError J")
    bp.close()
    # TwoCamera: per-block weights (the functor's own ctor argument) instead of 5 * w_visual[kf]
    tc = cfg["tc"]
    b = api.two_camera_batch(ctx, cfg["cam0"], cfg["cam1"], tc["left_ob"], tc["right_ob"], tc["lm_idx"], np.zeros_like(tc["kf_idx"]))
    w = np.random.default_rng(0).uniform(100, 600, len(tc["lm_idx"]))
    b.set_block_weights(w)
    b.evaluate(st)
    wk = np.concatenate([[1.0], w / 5.0])      # oracle: weight = 5 * w_kf[kf] -> give every block its own pseudo keyframe
    from tests.helpers import ocam
    r_ref, J_ref = oracle.two_camera(tc["left_ob"], tc["right_ob"], tc["lm_idx"], 1 + np.arange(len(w), dtype=np.int32), cfg["inv_depth"], wk, ocam(oracle, cfg["cam0"]), ocam(oracle, cfg["cam1"]))
    assert_parity(b.residuals(), r_ref, "TwoCamera r (block weights)"); assert_parity(b.jacobian(0)[:, :, 0], J_ref, "TwoCamera J (block weights)")
    # ... and through the solver: cost with block weights == cost of the per-keyframe rule when the weights agree
    prob = api.Problem(ctx, st, b, None, None, None)
    opt = api.default_solver_options()
    c_blk = prob.cost(opt)
    ref_cost = 0.0
    for i in range(len(w)):
        ci, _ = huber_scale(1.0, r_ref[i] @ r_ref[i]); ref_cost += 0.5 * ci
    assert abs(c_blk - ref_cost) <= 1e-10 * ref_cost
    it = prob.lm_iteration(opt, 1e4)
    assert abs(it["cost_before"] - ref_cost) <= 1e-10 * ref_cost and it["accepted"]
    b.set_block_weights(None)
    b2w = 5.0 * cfg["w_kf"][0]
    b.evaluate(st)
    r_kf, _ = oracle.two_camera(tc["left_ob"], tc["right_ob"], tc["lm_idx"], np.zeros_like(tc["kf_idx"]), st.get(api.INV_DEPTH), cfg["w_kf"], ocam(oracle, cfg["cam0"]), ocam(oracle, cfg["cam1"]))
    assert_parity(b.residuals(), r_kf, "TwoCamera r (per-keyframe rule restored)"); assert b2w > 0
    prob.close(); b.close(); st.close()
