"""GPU parity AT THE BASELINE SIZES — the configurations bench.py quotes its numbers on:

  configs[1]  500 000 PoseOnly blocks (10 k landmarks x 50 keyframes): every residual and every 2x7 Jacobian vs the oracle;
  configs[2]  all 100 000 scan points against the ~340 k-point map, both gates: indices and float32 d2 bit-exact;
  configs[3]  the 50-keyframe / 10 k-landmark window (both landmark-id orders): three LM iterations, reduced system S / rhs,
              accept-reject, radius and the whole state vs oracle/lm.h from identical state.

These exercise the geometry the small cases never reach: 40 band-Schur slices, the 10 k-row landmark sort, the merged
Schur + sparse-level-0 launch at full width, the persistent PoseOnly grid at 1954 tiles."""
import numpy as np
import pytest

from lvio_fusion_amd import synthetic as syn
from tests.helpers import assert_parity, ocam
from tests.test_gpu_solver import build, state_of

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from lvio_fusion_amd import api
    c = api.Context(0)
    yield c
    c.close()


def test_config2_all_500k_pose_only_blocks(ctx, oracle):
    from lvio_fusion_amd import api
    cfg = syn.config2_pose_only()
    assert cfg["ob"].shape[0] == 500000
    st = api.State(ctx, cfg["n_kf"], 0)
    st.set(api.POSES, cfg["poses"]); st.set(api.W_VISUAL, cfg["w_kf"])
    b = api.pose_only_batch(ctx, cfg["cam0"], cfg["ob"], cfg["kf_idx"], cfg["pw_idx"], cfg["pw"])
    b.evaluate(st)
    r_ref, J_ref = oracle.pose_only(cfg["ob"], cfg["kf_idx"], cfg["pw_idx"], cfg["pw"], cfg["poses"], cfg["w_kf"], ocam(oracle, cfg["cam0"]),
                                    threads=oracle.max_threads())
    assert_parity(b.residuals(), r_ref, "config2 r")
    assert_parity(b.jacobian(0), J_ref, "config2 J")
    b.close(); st.close()


def test_config3_all_100k_queries_bit_exact(ctx, oracle):
    from lvio_fusion_amd import api
    c = syn.config3_icp()
    assert c["query"].shape[0] == 100000
    mp = api.Map(ctx, c["map"], c["thr_ground"]); sc = api.Scan(ctx, c["query"])
    for thr in (c["thr_ground"], c["thr_surf"]):
        api.knn3(mp, sc, c["pose0"], thr)
        idx, d2, valid = sc.download()
        i0, d0, v0 = oracle.knn3(c["map"], c["query"], c["pose0"], thr, method=0, threads=oracle.max_threads())
        assert np.array_equal(valid, v0)
        sel = v0.astype(bool)
        assert sel.mean() > 0.5
        assert np.array_equal(idx[sel], i0[sel])
        assert np.array_equal(d2[sel].view(np.uint32), d0[sel].view(np.uint32))
    mp.close(); sc.close()


@pytest.mark.parametrize("ids_by_birth", [False, True])
def test_config4_full_window_lm_iterations(ctx, oracle, ids_by_birth):
    from lvio_fusion_amd import api
    cfg = syn.config4_window(ids_by_birth=ids_by_birth)
    assert cfg["n_kf"] == 50 and cfg["n_lm"] == 10000
    pre = np.stack([oracle.imu_preintegrate(f["samples"], f["acc0"], f["gyr0"], f["ba"], f["bg"], syn.IMU_NOISE) for f in cfg["imu"]])
    st = api.State(ctx, cfg["n_kf"], cfg["n_lm"])
    for field, key in ((api.POSES, "poses"), (api.VEL, "vel"), (api.BA, "ba"), (api.BG, "bg"), (api.INV_DEPTH, "inv_depth"), (api.W_VISUAL, "w_kf")):
        st.set(field, cfg[key])
    tc, tf, po = cfg["tc"], cfg["tf"], cfg["po"]
    btc = api.two_camera_batch(ctx, cfg["cam0"], cfg["cam1"], tc["left_ob"], tc["right_ob"], tc["lm_idx"], tc["kf_idx"])
    btf = api.two_frame_batch(ctx, cfg["cam0"], cfg["cam1"], tf["first_ob"], tf["ob"], tf["lm_idx"], tf["kf1_idx"], tf["kf2_idx"])
    bpo = api.pose_only_batch(ctx, cfg["cam0"], po["ob"], po["kf_idx"], po["pw_idx"], po["pw"])
    bimu = api.imu_batch(ctx, pre, [f["kf_i"] for f in cfg["imu"]], [f["kf_j"] for f in cfg["imu"]])
    prob = api.Problem(ctx, st, btc, btf, bpo, bimu)
    win = oracle.Window(cfg, pre)
    opt = api.default_solver_options()
    c_gpu = prob.cost(opt)
    assert abs(c_gpu - win.cost()) <= 1e-9 * abs(c_gpu)
    radius, dec = 1e4, 2.0
    for it in range(3):
        ref = win.lm_iteration(radius, dec)
        got = prob.lm_iteration(opt, radius, dec)
        assert abs(got["cost_before"] - ref["cost_before"]) <= 1e-8 * abs(ref["cost_before"])
        S, rhs = prob.reduced_system()
        assert np.abs(S - ref["S"]).max() <= 1e-7 * np.abs(ref["S"]).max(), f"iteration {it}: reduced system mismatch"
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
    for h in (prob, btc, btf, bpo, bimu, st):
        h.close()


def test_more_than_16k_landmarks(ctx, oracle):
    """20 000 landmarks in a short window: the one-workgroup scans of problem_configure (k_lm_offsets) and of the device-side window
    assembly (k_da_scan) work in passes of 16 k landmarks — both are taken through their second pass here.  The flat problem is
    compared with the oracle; the persistent window must then reproduce the flat problem's iteration."""
    from lvio_fusion_amd import api
    cfg = syn.config4_window(n_kf=8, n_lm=20000, n_prewindow=0, imu_samples=4, seed=2020)
    pre = np.stack([oracle.imu_preintegrate(f["samples"], f["acc0"], f["gyr0"], f["ba"], f["bg"], syn.IMU_NOISE) for f in cfg["imu"]])
    st = api.State(ctx, cfg["n_kf"], cfg["n_lm"])
    for field, key in ((api.POSES, "poses"), (api.VEL, "vel"), (api.BA, "ba"), (api.BG, "bg"), (api.INV_DEPTH, "inv_depth"), (api.W_VISUAL, "w_kf")):
        st.set(field, cfg[key])
    tc, tf, po = cfg["tc"], cfg["tf"], cfg["po"]
    assert len(po["kf_idx"]) == 0
    btc = api.two_camera_batch(ctx, cfg["cam0"], cfg["cam1"], tc["left_ob"], tc["right_ob"], tc["lm_idx"], tc["kf_idx"])
    btf = api.two_frame_batch(ctx, cfg["cam0"], cfg["cam1"], tf["first_ob"], tf["ob"], tf["lm_idx"], tf["kf1_idx"], tf["kf2_idx"])
    bimu = api.imu_batch(ctx, pre, [f["kf_i"] for f in cfg["imu"]], [f["kf_j"] for f in cfg["imu"]])
    prob = api.Problem(ctx, st, btc, btf, None, bimu)
    ref_win = oracle.Window(cfg, pre)
    opt = api.default_solver_options()
    ref = ref_win.lm_iteration(1e4, 2.0)
    got = prob.lm_iteration(opt, 1e4, 2.0)
    assert abs(got["cost_before"] - ref["cost_before"]) <= 1e-8 * abs(ref["cost_before"])
    assert got["accepted"] == ref["accepted"]
    assert abs(got["cost_after"] - ref["cost_after"]) <= 1e-6 * abs(ref["cost_after"])
    s = state_of(api, st)
    assert_parity(s["poses"].reshape(-1, 7), ref_win.poses, "poses")
    assert_parity(s["inv_depth"], ref_win.inv_depth, "inv_depth")
    # ---- the same window through lvf_window_* (device-side assembly), one iteration from the same start
    win = api.Window(ctx, cfg["cam0"], cfg["cam1"], baseline=syn.baseline(), weak_visual_threshold=0)
    order_tc = np.argsort(tc["kf_idx"], kind="stable")
    j = 0
    for k in range(cfg["n_kf"]):
        win.add_keyframe(k, cfg["poses"][k], cfg["w_kf"][k])
        win.set_imu(k, cfg["vel"][k], cfg["ba"][k], cfg["bg"][k], pre[k - 1] if k > 0 else None)
        while j < len(order_tc) and tc["kf_idx"][order_tc[j]] == k:
            i = order_tc[j]; j += 1
            win.add_landmark(int(tc["lm_idx"][i]), k, tc["left_ob"][i], tc["right_ob"][i], cfg["inv_depth"][tc["lm_idx"][i]])
        for i in np.nonzero(tf["kf2_idx"] == k)[0]:
            win.add_observation(int(tf["lm_idx"][i]), k, tf["ob"][i])
    opt1 = api.default_solver_options(); opt1.max_num_iterations = 1
    sw = win.solve(opt1)
    assert abs(sw.initial_cost - ref["cost_before"]) <= 1e-8 * abs(ref["cost_before"])
    for k in range(cfg["n_kf"]):
        assert_parity(win.pose(k), ref_win.poses[k], f"window pose {k}")
    for l in (0, 1, 9999, 16383, 16384, 19999):
        assert abs(win.inv_depth(l) - ref_win.inv_depth[l]) <= 1e-6 * abs(ref_win.inv_depth[l])
    win.solve(opt1)          # a second tick runs on the resident landmark table
    assert win.counts()["lm"] == 20000
    for h in (win, prob, btc, btf, bimu, st):
        h.close()
