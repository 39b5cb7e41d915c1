"""GPU parity of the on-device scan-to-map solve (lvf_icp_solve) against the oracle's restatement of
ScanToMapWithGround / ScanToMapWithSegmented + the 4-iteration 3-DoF solve (mapping.cpp:154-178)."""
import numpy as np
import pytest

from lvio_fusion_amd import synthetic as syn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from lvio_fusion_amd import api
    c = api.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def scene():
    c = syn.config3_icp()
    sel = np.sort(np.random.default_rng(1).choice(c["query"].shape[0], 6000, replace=False))
    return c, c["query"][sel], c["query_ground"][sel]


@pytest.mark.parametrize("mode,prior", [(0, 0.0), (1, 0.0), (0, 500 * syn.W_VISUAL), (1, 500 * syn.W_VISUAL)])
def test_icp_solve_parity(ctx, oracle, scene, mode, prior):
    from lvio_fusion_amd import api
    c, q, qg = scene
    qq = q[qg] if mode == 0 else q[~qg]
    mm = c["map"][c["map_ground"]] if mode == 0 else c["map"][~c["map_ground"]]
    thr = c["thr_ground"] if mode == 0 else c["thr_surf"]
    w = syn.W_LIDAR_GROUND if mode == 0 else syn.W_LIDAR_SURF
    huber = 0.0 if mode == 0 else 0.1
    rel = oracle.se3_mul(oracle.se3_inv(c["map_pose"]), c["pose0"])
    rpyxyz0 = oracle.se3_to_rpyxyz(rel)
    ref_x, ref = oracle.icp_solve(mm, qq, c["map_pose"], c["pose0"], rpyxyz0, mode, thr, w, huber, prior_w=prior)
    mp, sc = api.Map(ctx, mm, thr), api.Scan(ctx, qq)
    x = rpyxyz0.copy()
    s = api.icp_solve(mp, sc, c["map_pose"], c["pose0"], x, mode, thr, w, huber, prior_weight=prior)
    assert s.num_residual_blocks == ref["num_residual_blocks"] and s.num_residual_blocks > 500
    assert s.num_iterations == ref["num_iterations"] and s.num_successful_steps == ref["num_successful_steps"]
    assert abs(s.initial_cost - ref["initial_cost"]) <= 1e-9 * abs(ref["initial_cost"])
    assert abs(s.final_cost - ref["final_cost"]) <= 1e-6 * abs(ref["final_cost"])
    assert np.allclose(x, ref_x, rtol=1e-6, atol=1e-9)
    assert s.final_cost <= s.initial_cost and (prior > 0 or s.final_cost < s.initial_cost)
    untouched = [0, 3, 4] if mode == 0 else [1, 2, 5]
    assert np.array_equal(x[untouched], rpyxyz0[untouched])      # only the three parameter blocks of the sub-problem move
    mp.close(); sc.close()


def test_icp_recovers_pose_like_mapping_optimize(ctx, oracle, scene):
    """Mapping::Optimize order: ground (pitch,roll,z) then surf (yaw,x,y); the composed pose moves toward the truth."""
    from lvio_fusion_amd import api
    c, q, qg = scene
    pose = c["pose0"].copy()
    rel = oracle.se3_mul(oracle.se3_inv(c["map_pose"]), pose)
    x = oracle.se3_to_rpyxyz(rel)
    e0 = np.abs(pose[4:] - c["pose_true"][4:]).max()
    for mode in (0, 1):
        qq = q[qg] if mode == 0 else q[~qg]
        mm = c["map"][c["map_ground"]] if mode == 0 else c["map"][~c["map_ground"]]
        thr = c["thr_ground"] if mode == 0 else c["thr_surf"]
        mp, sc = api.Map(ctx, mm, thr), api.Scan(ctx, qq)
        api.icp_solve(mp, sc, c["map_pose"], pose, x, mode, thr, syn.W_LIDAR_GROUND if mode == 0 else syn.W_LIDAR_SURF,
                      0.0 if mode == 0 else 0.1)
        pose = oracle.se3_mul(c["map_pose"], oracle.rpyxyz_to_se3(x))      # mapping.cpp:164
        mp.close(); sc.close()
    e1 = np.abs(pose[4:] - c["pose_true"][4:]).max()
    assert e1 < 0.5 * e0


def test_icp_no_correspondences(ctx):
    from lvio_fusion_amd import api
    m = np.zeros((10, 4), np.float32); m[:, 0] = 1000.0
    q = np.zeros((50, 4), np.float32)
    mp, sc = api.Map(ctx, m, 1.0), api.Scan(ctx, q)
    x = np.array([0.1, 0.0, 0.0, 1.0, 2.0, 3.0])
    s = api.icp_solve(mp, sc, [0, 0, 0, 1, 0, 0, 0], [0, 0, 0, 1, 1, 2, 3], x, 1, 1.0, 0.01, 0.1)
    assert s.num_residual_blocks == 0 and s.final_cost == 0.0
    assert np.array_equal(x, [0.1, 0.0, 0.0, 1.0, 2.0, 3.0])
    mp.close(); sc.close()


@pytest.mark.parametrize("mode,prior", [(0, 0.0), (1, 500 * syn.W_VISUAL)])
def test_lidar_solve_on_caller_built_batch(ctx, oracle, scene, mode, prior):
    """adapt::Solve-level drop-in: the caller keeps association.cpp's host loop (here: the device 3-NN result downloaded
    and turned into LidarPlaneError constructor arguments), hands the blocks over as one lidar batch, and
    lvf_lidar_solve must land exactly where the oracle's ScanToMap + solve lands."""
    from lvio_fusion_amd import api
    c, q, qg = scene
    qq = q[qg] if mode == 0 else q[~qg]
    mm = c["map"][c["map_ground"]] if mode == 0 else c["map"][~c["map_ground"]]
    thr = c["thr_ground"] if mode == 0 else c["thr_surf"]
    w = syn.W_LIDAR_GROUND if mode == 0 else syn.W_LIDAR_SURF
    huber = 0.0 if mode == 0 else 0.1
    rpyxyz0 = oracle.se3_to_rpyxyz(oracle.se3_mul(oracle.se3_inv(c["map_pose"]), c["pose0"]))
    ref_x, ref = oracle.icp_solve(mm, qq, c["map_pose"], c["pose0"], rpyxyz0, mode, thr, w, huber, prior_w=prior)
    mp, sc = api.Map(ctx, mm, thr), api.Scan(ctx, qq)
    api.knn3(mp, sc, c["pose0"], thr)
    idx, d2, valid = sc.download()
    v = valid > 0
    p = qq[v, :3].astype(np.float64)
    pa, pb, pc = (mm[idx[v, k], :3].astype(np.float64) for k in range(3))
    b = api.lidar_plane_batch(ctx, mode, p, pa, pb, pc, c["map_pose"], w)
    x = rpyxyz0.copy()
    s = api.lidar_solve(b, x, huber, prior_weight=prior)
    assert s.num_residual_blocks == ref["num_residual_blocks"]
    assert s.num_iterations == ref["num_iterations"] and s.num_successful_steps == ref["num_successful_steps"]
    assert abs(s.initial_cost - ref["initial_cost"]) <= 1e-9 * abs(ref["initial_cost"])
    assert abs(s.final_cost - ref["final_cost"]) <= 1e-6 * abs(ref["final_cost"])
    assert np.allclose(x, ref_x, rtol=1e-6, atol=1e-9)
    b.close(); mp.close(); sc.close()


@pytest.mark.parametrize("max_iters", [0, 1, 2, 3])
@pytest.mark.parametrize("far", [False, True])
def test_icp_iteration_caps_and_a_far_start(ctx, oracle, scene, max_iters, far):
    """The device loop runs ONE pass per LM iteration (the candidate pass carries the Jacobian; the last allowed iteration's pass only the
    cost): every cap, and a start far enough for the trust region to work (rejected / short steps), must end where ceres::Solve's
    restatement ends — same counts, same state."""
    from lvio_fusion_amd import api
    c, q, qg = scene
    mode = 1
    qq, mm = q[~qg], c["map"][~c["map_ground"]]
    thr, w, huber = c["thr_surf"], syn.W_LIDAR_SURF, 0.1
    pose0 = np.array(c["pose0"], float)
    if far:
        d = np.array([0.03, -0.02, 0.05, 1.0]); d /= np.linalg.norm(d)
        pose0 = oracle.se3_mul(np.concatenate([d, [0.6, -0.4, 0.1]]), pose0)
    rel = oracle.se3_mul(oracle.se3_inv(c["map_pose"]), pose0)
    rpyxyz0 = oracle.se3_to_rpyxyz(rel)
    ref_x, ref = oracle.icp_solve(mm, qq, c["map_pose"], pose0, rpyxyz0, mode, thr, w, huber, prior_w=0.0, max_iters=max_iters)
    mp, sc = api.Map(ctx, mm, thr), api.Scan(ctx, qq)
    x = rpyxyz0.copy()
    s = api.icp_solve(mp, sc, c["map_pose"], pose0, x, mode, thr, w, huber, prior_weight=0.0, max_num_iterations=max_iters)
    assert s.num_residual_blocks == ref["num_residual_blocks"]
    assert (s.num_iterations, s.num_successful_steps) == (ref["num_iterations"], ref["num_successful_steps"])
    assert abs(s.initial_cost - ref["initial_cost"]) <= 1e-9 * abs(ref["initial_cost"])
    assert abs(s.final_cost - ref["final_cost"]) <= 1e-6 * abs(ref["final_cost"]) + 1e-12
    assert np.allclose(x, ref_x, rtol=1e-6, atol=1e-9)
    if max_iters == 0:
        assert np.array_equal(x, rpyxyz0) and s.num_iterations == 0
    mp.close(); sc.close()
