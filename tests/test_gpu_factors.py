"""GPU parity: every factor batch evaluated through the C-ABI vs the CPU oracle (Jet autodiff) on the same seeded
inputs.  Tolerance: 1e-6 relative fp64 (north_star); measured agreement is ~1e-12."""
import numpy as np
import pytest

from lvio_fusion_amd import synthetic as syn
from tests.helpers import assert_parity, ocam

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from lvio_fusion_amd import api
    c = api.Context(0)
    yield c
    c.close()


def _state(api, ctx, cfg, poses=None):
    st = api.State(ctx, cfg["n_kf"], cfg.get("n_lm", 0) if "inv_depth" in cfg else 0)
    st.set(api.POSES, cfg["poses"] if poses is None else poses)
    st.set(api.W_VISUAL, cfg["w_kf"])
    if "inv_depth" in cfg:
        st.set(api.INV_DEPTH, cfg["inv_depth"]); st.set(api.VEL, cfg["vel"]); st.set(api.BA, cfg["ba"]); st.set(api.BG, cfg["bg"])
    return st


@pytest.mark.parametrize("n_lm,n_kf,nonunit", [(300, 7, False), (1000, 50, True), (37, 70, True), (1, 1, False)])
def test_pose_only_parity(ctx, oracle, n_lm, n_kf, nonunit):
    from lvio_fusion_amd import api
    cfg = syn.config2_pose_only(n_lm=n_lm, n_kf=n_kf, seed=100 + n_lm)
    poses = cfg["poses"].copy()
    if nonunit:   # exercise the d(q/|q|)/dq projector
        poses[:, :4] *= np.random.default_rng(1).uniform(0.5, 1.8, (n_kf, 1))
    st = _state(api, ctx, cfg, poses)
    b = api.pose_only_batch(ctx, cfg["cam0"], cfg["ob"], cfg["kf_idx"], cfg["pw_idx"], cfg["pw"])
    b.evaluate(st)
    r_ref, J_ref = oracle.pose_only(cfg["ob"], cfg["kf_idx"], cfg["pw_idx"], cfg["pw"], poses, cfg["w_kf"], ocam(oracle, cfg["cam0"]))
    assert_parity(b.residuals(), r_ref, "pose_only r")
    assert_parity(b.jacobian(0), J_ref, "pose_only J")
    # residual-only evaluation (jacobians == NULL in CostFunction::Evaluate)
    b.evaluate(st, jacobians=False)
    assert_parity(b.residuals(), r_ref, "pose_only r (no J)")
    with pytest.raises(api.LvfError):
        b.jacobian(0)
    b.close(); st.close()


def test_pose_only_unsorted_and_ragged(ctx, oracle):
    """Unsorted keyframe order and a block count that is not a multiple of the workgroup size."""
    from lvio_fusion_amd import api
    cfg = syn.config2_pose_only(n_lm=53, n_kf=11, seed=5)
    perm = np.random.default_rng(0).permutation(cfg["ob"].shape[0])[:517]
    ob, kf, pw_idx = cfg["ob"][perm], cfg["kf_idx"][perm], cfg["pw_idx"][perm]
    st = _state(api, ctx, cfg)
    b = api.pose_only_batch(ctx, cfg["cam0"], ob, kf, pw_idx, cfg["pw"])
    b.evaluate(st)
    r_ref, J_ref = oracle.pose_only(ob, kf, pw_idx, cfg["pw"], cfg["poses"], cfg["w_kf"], ocam(oracle, cfg["cam0"]))
    assert_parity(b.residuals(), r_ref, "r"); assert_parity(b.jacobian(0), J_ref, "J")
    b.close(); st.close()


def test_empty_batches(ctx):
    from lvio_fusion_amd import api
    cams = syn.kitti_cameras()
    st = api.State(ctx, 3, 2)
    b = api.pose_only_batch(ctx, cams[0], np.zeros((0, 2)), np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros((0, 3)))
    b.evaluate(st)
    assert b.residuals().shape == (0, 2) and b.jacobian(0).shape == (0, 2, 7)
    b.close()
    b = api.two_frame_batch(ctx, cams[0], cams[1], np.zeros((0, 2)), np.zeros((0, 2)), [], [], [])
    b.evaluate(st); assert b.residuals().shape == (0, 2); b.close()
    st.close()


def test_error_paths(ctx):
    from lvio_fusion_amd import api
    cams = syn.kitti_cameras()
    with pytest.raises(api.LvfError):      # negative index
        api.pose_only_batch(ctx, cams[0], np.zeros((1, 2)), [-1], [0], np.zeros((1, 3)))
    with pytest.raises(api.LvfError):      # landmark index out of range
        api.pose_only_batch(ctx, cams[0], np.zeros((1, 2)), [0], [3], np.zeros((1, 3)))
    b = api.pose_only_batch(ctx, cams[0], np.zeros((1, 2)), [5], [0], np.ones((1, 3)))
    st = api.State(ctx, 2, 0)
    with pytest.raises(api.LvfError):      # state smaller than the batch's keyframe indices
        b.evaluate(st)
    with pytest.raises(api.LvfError):      # download before evaluate
        b.residuals()
    b.close(); st.close()


@pytest.mark.parametrize("n_kf,n_lm,seed", [(6, 60, 11), (50, 1500, 12), (80, 300, 13)])
def test_two_frame_and_two_camera_parity(ctx, oracle, n_kf, n_lm, seed):
    from lvio_fusion_amd import api
    cfg = syn.config4_window(n_kf=n_kf, n_lm=n_lm, n_prewindow=20, seed=seed, imu_samples=3)
    poses = cfg["poses"].copy()
    poses[:, :4] *= np.random.default_rng(2).uniform(0.7, 1.4, (n_kf, 1))
    st = _state(api, ctx, cfg, poses)
    left, right = ocam(oracle, cfg["cam0"]), ocam(oracle, cfg["cam1"])
    tf = cfg["tf"]
    b = api.two_frame_batch(ctx, cfg["cam0"], cfg["cam1"], tf["first_ob"], tf["ob"], tf["lm_idx"], tf["kf1_idx"], tf["kf2_idx"])
    b.evaluate(st)
    r, Jd, J1, J2 = oracle.two_frame(tf["first_ob"], tf["ob"], tf["lm_idx"], tf["kf1_idx"], tf["kf2_idx"], cfg["inv_depth"], poses, cfg["w_kf"], left, right)
    assert_parity(b.residuals(), r, "two_frame r")
    assert_parity(b.jacobian(0)[:, :, 0], Jd, "two_frame J_invdepth")
    assert_parity(b.jacobian(1), J1, "two_frame J_pose1")
    assert_parity(b.jacobian(2), J2, "two_frame J_pose2")
    b.close()
    tc = cfg["tc"]
    b = api.two_camera_batch(ctx, cfg["cam0"], cfg["cam1"], tc["left_ob"], tc["right_ob"], tc["lm_idx"], tc["kf_idx"])
    b.evaluate(st)
    r, J = oracle.two_camera(tc["left_ob"], tc["right_ob"], tc["lm_idx"], tc["kf_idx"], cfg["inv_depth"], cfg["w_kf"], left, right)
    assert_parity(b.residuals(), r, "two_camera r")
    assert_parity(b.jacobian(0)[:, :, 0], J, "two_camera J")
    b.close(); st.close()


@pytest.mark.parametrize("mode", [0, 1])
def test_lidar_plane_parity(ctx, oracle, mode):
    from lvio_fusion_amd import api
    rng = np.random.default_rng(40 + mode)
    n = 5000 + 13
    p = rng.uniform(-30, 30, (n, 3)); pa = p + rng.normal(0, 0.5, (n, 3))
    pb = pa + rng.normal(0, 0.4, (n, 3)); pc = pa + rng.normal(0, 0.4, (n, 3))
    Twc1 = syn.drive_poses(4, rng)[3]
    Twc1[:4] *= 1.3
    rpyxyz = np.array([0.21, -0.04, 0.03, 0.9, -0.3, 0.05])
    w = syn.W_LIDAR_GROUND if mode == 0 else syn.W_LIDAR_SURF
    b = api.lidar_plane_batch(ctx, mode, p, pa, pb, pc, Twc1, w)
    nrm = oracle.plane_normals(pa, pb, pc)
    assert_parity(b.normals(), nrm, "normals")
    b.evaluate(None, rpyxyz=rpyxyz)
    r, J = oracle.lidar_plane(mode, p, pa, nrm, Twc1, rpyxyz, w)
    assert_parity(b.residuals()[:, 0], r, "lidar r")
    for k in range(3):
        assert_parity(b.jacobian(k)[:, 0, 0], J[:, k], f"lidar J{k}")
    # the functor holds a LIVE pointer to rpyxyz (lidar_error.hpp:52,87): a second evaluate sees the new values
    rpyxyz2 = rpyxyz + np.array([0.01, 0.002, -0.003, 0.05, 0.02, -0.01])
    b.evaluate(None, rpyxyz=rpyxyz2)
    r2, J2 = oracle.lidar_plane(mode, p, pa, nrm, Twc1, rpyxyz2, w)
    assert_parity(b.residuals()[:, 0], r2, "lidar r (moved)")
    b.close()


def test_imu_parity(ctx, oracle):
    from lvio_fusion_amd import api
    cfg = syn.config4_window(n_kf=12, n_lm=30, n_prewindow=5, seed=77, imu_samples=10)
    pre = np.stack([oracle.imu_preintegrate(f["samples"], f["acc0"], f["gyr0"], f["ba"], f["bg"], syn.IMU_NOISE) for f in cfg["imu"]])
    kf_i = [f["kf_i"] for f in cfg["imu"]]; kf_j = [f["kf_j"] for f in cfg["imu"]]
    st = _state(api, ctx, cfg)
    b = api.imu_batch(ctx, pre, kf_i, kf_j)
    b.evaluate(st)
    r, J = oracle.imu_eval(pre, kf_i, kf_j, cfg["poses"], cfg["vel"], cfg["ba"], cfg["bg"])
    assert_parity(b.residuals(), r, "imu r")
    Jb = oracle.imu_split_jac(J)
    for k in range(8):
        assert_parity(b.jacobian(k), Jb[k], f"imu J{k}")
    b.close(); st.close()


def test_preintegration_parity(ctx, oracle):
    """lvf_preintegrate (device mid-point integration + F/V covariance propagation) vs the oracle, ragged sample counts."""
    from lvio_fusion_amd import api
    cfg = syn.config4_window(n_kf=9, n_lm=10, n_prewindow=2, seed=88, imu_samples=10)
    rng = np.random.default_rng(0)
    samples = [f["samples"][: int(rng.integers(1, 11))] for f in cfg["imu"]]
    samples[2] = np.zeros((0, 7))                     # a pair with no samples: identity jacobian, zero covariance
    samples[3] = np.concatenate([cfg["imu"][3]["samples"]] * 10)   # 100 samples (100 Hz IMU)
    a0 = np.stack([f["acc0"] for f in cfg["imu"]]); g0 = np.stack([f["gyr0"] for f in cfg["imu"]])
    ba = np.stack([f["ba"] for f in cfg["imu"]]); bg = np.stack([f["bg"] for f in cfg["imu"]])
    got = api.preintegrate(ctx, samples, a0, g0, ba, bg, syn.IMU_NOISE)
    for k, s in enumerate(samples):
        ref = oracle.imu_preintegrate(s, a0[k], g0[k], ba[k], bg[k], syn.IMU_NOISE)
        assert_parity(got[k][:17], ref[:17], f"pair {k} state")
        assert_parity(got[k][17:242], ref[17:242], f"pair {k} jacobian")
        assert_parity(got[k][242:], ref[242:], f"pair {k} covariance")
    assert api.preintegrate(ctx, [], np.zeros((0, 3)), np.zeros((0, 3)), np.zeros((0, 3)), np.zeros((0, 3)), syn.IMU_NOISE).shape == (0, 467)
