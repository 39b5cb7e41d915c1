"""The C++ host interface (include/lvf_ceres_adapter.hpp: reference-shaped CostFunction factories + adapt::Solve routed to
the GPU) driven by lvio_fusion_amd/host/adapter_selftest in the exact shape of Backend::BuildProblem / ScanToMapWith* —
checked against the oracle (per-block Evaluate, batched residual vector) and against the same window solved through the
flat C-ABI from Python."""
import json
import os
import subprocess

import numpy as np
import pytest

from lvio_fusion_amd import synthetic as syn
from tests.helpers import assert_parity, ocam

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "lvio_fusion_amd", "host", "adapter_selftest")

pytestmark = pytest.mark.gpu


def _dump(d, name, arr, dtype):
    np.ascontiguousarray(arr, dtype=dtype).tofile(os.path.join(d, name))


def _cam_vec(c):
    return np.concatenate([[c["fx"], c["fy"], c["cx"], c["cy"]], c["extrinsic"]])


def _run(*args):
    assert os.path.exists(EXE), f"{EXE} is missing: run __graft_entry__.build()"
    p = subprocess.run([EXE, *args], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, f"adapter_selftest failed: {p.stdout}\n{p.stderr}"
    return json.loads(p.stdout.strip().splitlines()[-1])


def _sorted_by_kf(cfg):
    """TwoCamera blocks in per-frame order (BuildProblem walks frames, then each frame's features)."""
    tc = cfg["tc"]
    o = np.argsort(tc["kf_idx"], kind="stable")
    cfg = dict(cfg)
    cfg["tc"] = {k: v[o] for k, v in tc.items()}
    return cfg


@pytest.mark.parametrize("with_imu,weak_thr,const_kf", [(True, 0, -1), (False, 10 ** 6, -1), (False, 0, 0)])
def test_window_through_adapter(tmp_path, oracle, with_imu, weak_thr, const_kf):
    from lvio_fusion_amd import api
    n_kf, n_lm, max_it = 7, 90, 6
    cfg = _sorted_by_kf(syn.config4_window(n_kf=n_kf, n_lm=n_lm, n_prewindow=30, seed=77, imu_samples=4))
    pre = np.stack([oracle.imu_preintegrate(f["samples"], f["acc0"], f["gyr0"], f["ba"], f["bg"], syn.IMU_NOISE) for f in cfg["imu"]])
    d = str(tmp_path)
    tc, tf, po = cfg["tc"], cfg["tf"], cfg["po"]
    _dump(d, "meta.i32", [n_kf, n_lm, max_it, weak_thr, const_kf], np.int32)
    for name, key in (("poses", "poses"), ("vel", "vel"), ("ba", "ba"), ("bg", "bg"), ("inv_depth", "inv_depth"), ("w_kf", "w_kf")):
        _dump(d, name + ".f64", cfg[key], np.float64)
    _dump(d, "cam0.f64", _cam_vec(cfg["cam0"]), np.float64); _dump(d, "cam1.f64", _cam_vec(cfg["cam1"]), np.float64)
    _dump(d, "tc_left_ob.f64", tc["left_ob"], np.float64); _dump(d, "tc_right_ob.f64", tc["right_ob"], np.float64)
    _dump(d, "tc_lm.i32", tc["lm_idx"], np.int32); _dump(d, "tc_kf.i32", tc["kf_idx"], np.int32)
    _dump(d, "tf_first_ob.f64", tf["first_ob"], np.float64); _dump(d, "tf_ob.f64", tf["ob"], np.float64)
    _dump(d, "tf_lm.i32", tf["lm_idx"], np.int32); _dump(d, "tf_kf1.i32", tf["kf1_idx"], np.int32); _dump(d, "tf_kf2.i32", tf["kf2_idx"], np.int32)
    _dump(d, "po_ob.f64", po["ob"], np.float64); _dump(d, "po_pw.f64", po["pw"], np.float64)
    _dump(d, "po_kf.i32", po["kf_idx"], np.int32); _dump(d, "po_pw_idx.i32", po["pw_idx"], np.int32)
    imu = cfg["imu"] if with_imu else []
    _dump(d, "preint.f64", pre if with_imu else np.zeros(0), np.float64)
    _dump(d, "imu_i.i32", [f["kf_i"] for f in imu], np.int32); _dump(d, "imu_j.i32", [f["kf_j"] for f in imu], np.int32)
    out = _run("window", d)
    assert out["ok"] == 1 and out["num_frames"] == n_kf

    # ---- expected priors (backend.cpp:164-178 with the test's threshold): frames without IMU and few visual blocks
    kf_a, kf_b, tgt = [], [], []
    if not with_imu:
        for k in range(n_kf):
            n_vis = int((po["kf_idx"] == k).sum() + (tf["kf2_idx"] == k).sum())
            if n_vis < weak_thr:
                if k == 0:
                    kf_a.append(-1); kf_b.append(0); tgt.append(cfg["poses"][0])
                else:
                    kf_a.append(k - 1); kf_b.append(k)
                    tgt.append(np.concatenate([oracle.pose_graph_target(cfg["poses"][k - 1], cfg["poses"][k]), [0.0]]))
    assert out["n_prior"] == len(kf_b)
    n_blocks = len(tc["lm_idx"]) + len(tf["lm_idx"]) + len(po["kf_idx"]) + len(imu) + len(kf_b)
    assert out["num_residual_blocks"] == n_blocks

    # ---- the same window through the flat C-ABI from python (identical kernels; only block/landmark numbering differs)
    ctx = api.Context(0)
    st = api.State(ctx, n_kf, n_lm)
    for field, key in ((api.POSES, "poses"), (api.VEL, "vel"), (api.BA, "ba"), (api.BG, "bg"), (api.INV_DEPTH, "inv_depth"), (api.W_VISUAL, "w_kf")):
        st.set(field, cfg[key])
    btc = api.two_camera_batch(ctx, cfg["cam0"], cfg["cam1"], tc["left_ob"], tc["right_ob"], tc["lm_idx"], tc["kf_idx"])
    btf = api.two_frame_batch(ctx, cfg["cam0"], cfg["cam1"], tf["first_ob"], tf["ob"], tf["lm_idx"], tf["kf1_idx"], tf["kf2_idx"])
    bpo = api.pose_only_batch(ctx, cfg["cam0"], po["ob"], po["kf_idx"], po["pw_idx"], po["pw"])
    bimu = api.imu_batch(ctx, pre, [f["kf_i"] for f in imu], [f["kf_j"] for f in imu]) if with_imu else None
    prob = api.Problem(ctx, st, btc, btf, bpo, bimu)
    bpr = None
    if kf_b:
        bpr = api.pose_prior_batch(ctx, kf_a, kf_b, np.array(tgt), np.full(len(kf_b), 100.0), np.zeros(len(kf_b)))
        prob.set_pose_priors(bpr)
    if const_kf >= 0:
        prob.set_pose_constant(const_kf, True)
    opt = api.default_solver_options()
    opt.max_num_iterations = max_it
    cost0 = prob.cost(opt)
    assert abs(out["cost0"] - cost0) <= 1e-9 * cost0
    summ = prob.solve(opt)
    assert abs(out["initial_cost"] - summ.initial_cost) <= 1e-9 * summ.initial_cost
    assert abs(out["final_cost"] - summ.final_cost) <= 1e-6 * summ.final_cost
    assert out["successful"] == summ.num_successful_steps and out["successful"] >= 2
    assert out["final_cost"] < 0.5 * out["initial_cost"]
    rd = lambda name: np.fromfile(os.path.join(d, name))
    assert_parity(rd("out_poses.f64"), st.get(api.POSES), "poses written back in place")
    assert_parity(rd("out_inv_depth.f64"), st.get(api.INV_DEPTH), "inverse depths written back in place")
    if with_imu:
        assert_parity(rd("out_vel.f64"), st.get(api.VEL), "velocities"); assert_parity(rd("out_ba.f64"), st.get(api.BA), "ba")
        assert_parity(rd("out_bg.f64"), st.get(api.BG), "bg")
    else:   # blocks that were never registered are never written
        assert np.array_equal(rd("out_vel.f64"), cfg["vel"].ravel())
    if const_kf >= 0:
        assert np.array_equal(rd("out_poses.f64")[7 * const_kf:7 * const_kf + 7], cfg["poses"][const_kf])

    # ---- batched Evaluate: residual vector in block insertion order vs the oracle
    c0, c1 = ocam(oracle, cfg["cam0"]), ocam(oracle, cfg["cam1"])
    r_tc, _ = oracle.two_camera(tc["left_ob"], tc["right_ob"], tc["lm_idx"], tc["kf_idx"], cfg["inv_depth"], cfg["w_kf"], c0, c1, jac=False)
    r_tf = oracle.two_frame(tf["first_ob"], tf["ob"], tf["lm_idx"], tf["kf1_idx"], tf["kf2_idx"], cfg["inv_depth"], cfg["poses"], cfg["w_kf"], c0, c1, jac=False)[0]
    r_po, _ = oracle.pose_only(po["ob"], po["kf_idx"], po["pw_idx"], po["pw"], cfg["poses"], cfg["w_kf"], c0, jac=False)
    r_imu = oracle.imu_eval(pre, [f["kf_i"] for f in imu], [f["kf_j"] for f in imu], cfg["poses"], cfg["vel"], cfg["ba"], cfg["bg"], jac=False)[0] if with_imu else None
    exp = []
    ip = 0
    for k in range(n_kf):
        exp += [r_tc[i] for i in np.nonzero(tc["kf_idx"] == k)[0]]
        exp += [r_po[i] for i in np.nonzero(po["kf_idx"] == k)[0]]
        exp += [r_tf[i] for i in np.nonzero(tf["kf2_idx"] == k)[0]]
        if with_imu and k > 0:
            exp.append(r_imu[k - 1])
        if ip < len(kf_b) and kf_b[ip] == k:
            exp.append(np.zeros(6)); ip += 1      # priors are anchored at the initial poses: zero residual
    exp = np.concatenate(exp)
    got = rd("out_residuals.f64")
    assert got.shape == exp.shape
    scale = np.abs(exp).max()
    assert np.abs(got - exp).max() <= 1e-6 * scale

    # ---- upstream-shaped Problem::Evaluate: robustified residuals, gradient, CRS Jacobian in local coordinates vs a numpy assembly
    from tests.test_oracle_lm_numpy import dense_system, to_local
    cfg_d = dict(cfg); cfg_d["imu"] = list(imu)
    state0 = [np.asarray(cfg[k], dtype=np.float64) for k in ("poses", "vel", "ba", "bg", "inv_depth")]
    Jd, rd_, cost_d = dense_system(oracle, cfg_d, pre if with_imu else np.zeros((0, 467)), state0)
    n_tc, n_tf, n_po = len(tc["lm_idx"]), len(tf["lm_idx"]), len(po["kf_idx"])
    ncol_d = 15 * n_kf + n_lm
    rows_e, res_e = [], []
    ip = 0
    for k in range(n_kf):
        for i in np.nonzero(tc["kf_idx"] == k)[0]:
            rows_e.append(Jd[2 * i:2 * i + 2]); res_e.append(rd_[2 * i:2 * i + 2])
        for i in np.nonzero(po["kf_idx"] == k)[0]:
            o = 2 * (n_tc + n_tf + i); rows_e.append(Jd[o:o + 2]); res_e.append(rd_[o:o + 2])
        for i in np.nonzero(tf["kf2_idx"] == k)[0]:
            o = 2 * (n_tc + i); rows_e.append(Jd[o:o + 2]); res_e.append(rd_[o:o + 2])
        if with_imu and k > 0:
            o = 2 * (n_tc + n_tf + n_po) + 15 * (k - 1); rows_e.append(Jd[o:o + 15]); res_e.append(rd_[o:o + 15])
        if ip < len(kf_b) and kf_b[ip] == k:
            blk = np.zeros((6, ncol_d))
            if kf_a[ip] < 0:
                r6, J6 = oracle.pose_prior(tgt[ip], 100.0, 0.0, cfg["poses"][k]); blk[:, 6 * k:6 * k + 6] = to_local(J6, cfg["poses"][k])
            else:
                r6, Ja, Jb = oracle.pose_graph(tgt[ip][:6], 100.0, 0.0, cfg["poses"][k - 1], cfg["poses"][k])
                blk[:, 6 * (k - 1):6 * k] = to_local(Ja, cfg["poses"][k - 1]); blk[:, 6 * k:6 * k + 6] = to_local(Jb, cfg["poses"][k])
            rows_e.append(blk); res_e.append(r6); ip += 1
    J_exp, r_exp = np.concatenate(rows_e), np.concatenate(res_e)
    if const_kf >= 0:
        J_exp[:, 6 * const_kf:6 * const_kf + 6] = 0.0
    colmap = np.fromfile(os.path.join(d, "out_crs_colmap.i32"), dtype=np.int32).reshape(-1, 3)
    crs_rows, crs_cols = np.fromfile(os.path.join(d, "out_crs_rows.i32"), dtype=np.int32), np.fromfile(os.path.join(d, "out_crs_cols.i32"), dtype=np.int32)
    crs_vals = rd("out_crs_values.f64")
    assert out["crs_rows"] == J_exp.shape[0] == len(crs_rows) - 1
    J_got = np.zeros((out["crs_rows"], out["crs_cols"]))
    for i in range(out["crs_rows"]):
        cs = crs_cols[crs_rows[i]:crs_rows[i + 1]]
        assert np.all(np.diff(cs) > 0), "CRS columns of a row must ascend"
        J_got[i, cs] = crs_vals[crs_rows[i]:crs_rows[i + 1]]
    local = {0: 6, 1: 3, 2: 3, 3: 3, 4: 1}
    dense_col = lambda kind, idx: (6 * idx if kind == 0 else (6 * n_kf + 9 * idx + 3 * (kind - 1) if kind < 4 else 15 * n_kf + idx))
    perm = np.concatenate([np.arange(dense_col(k_, i_), dense_col(k_, i_) + local[k_]) for k_, i_, c_ in colmap])
    assert [c_ for _, _, c_ in colmap] == list(np.cumsum([0] + [local[k_] for k_, _, _ in colmap])[:-1])
    assert_parity(rd("out_eval_residuals.f64"), r_exp, "Problem::Evaluate residuals (loss applied)")
    scale = np.abs(J_exp).max()
    assert np.abs(J_got - J_exp[:, perm]).max() <= 1e-6 * scale, "Problem::Evaluate CRS Jacobian"
    g_exp = (J_exp.T @ r_exp)[perm]
    assert np.abs(rd("out_eval_gradient.f64") - g_exp).max() <= 1e-6 * np.abs(g_exp).max(), "Problem::Evaluate gradient"
    if const_kf >= 0:       # a constant block keeps its columns, without entries
        c0_ = int(colmap[(colmap[:, 0] == 0) & (colmap[:, 1] == const_kf)][0, 2])
        assert not np.isin(crs_cols, np.arange(c0_, c0_ + 6)).any()
    # subset call: columns = (pose n_kf-1, pose 1), raw residual Jacobian (apply_loss_function = false)
    Jr, rr_, _ = dense_system(oracle, cfg_d, pre if with_imu else np.zeros((0, 467)), state0, huber_a=0.0)
    sub_rows, sub_cols, sub_vals = np.fromfile(os.path.join(d, "out_sub_rows.i32"), dtype=np.int32), np.fromfile(os.path.join(d, "out_sub_cols.i32"), dtype=np.int32), rd("out_sub_values.f64")
    J_sub = np.zeros((len(sub_rows) - 1, 12))
    for i in range(len(sub_rows) - 1):
        J_sub[i, sub_cols[sub_rows[i]:sub_rows[i + 1]]] = sub_vals[sub_rows[i]:sub_rows[i + 1]]
    # rows of the raw system in insertion order (priors carry no loss: same rows as above)
    rows_r, ip = [], 0
    pr_rows = [b_ for b_ in rows_e if b_.shape[0] == 6]
    for k in range(n_kf):
        rows_r += [Jr[2 * i:2 * i + 2] for i in np.nonzero(tc["kf_idx"] == k)[0]]
        rows_r += [Jr[2 * (n_tc + n_tf + i):2 * (n_tc + n_tf + i) + 2] for i in np.nonzero(po["kf_idx"] == k)[0]]
        rows_r += [Jr[2 * (n_tc + i):2 * (n_tc + i) + 2] for i in np.nonzero(tf["kf2_idx"] == k)[0]]
        if with_imu and k > 0:
            o = 2 * (n_tc + n_tf + n_po) + 15 * (k - 1); rows_r.append(Jr[o:o + 15])
        if ip < len(kf_b) and kf_b[ip] == k:
            rows_r.append(pr_rows[ip]); ip += 1
    Jr_exp = np.concatenate(rows_r)
    sel = np.concatenate([np.arange(6 * (n_kf - 1), 6 * n_kf), np.arange(6, 12)])
    if const_kf == 1 or const_kf == n_kf - 1:
        Jr_exp[:, 6 * const_kf:6 * const_kf + 6] = 0.0
    assert np.abs(J_sub - Jr_exp[:, sel]).max() <= 1e-6 * np.abs(Jr_exp).max(), "Problem::Evaluate with options.parameter_blocks"

    # ---- per-block CostFunction::Evaluate probes: [TwoCamera | PoseOnly | TwoFrame | ImuError], jacobians[1] = NULL honoured
    probe = rd("out_probe.f64")
    assert out["n_probe"] == (4 if with_imu else 3)
    i_tc = int(np.nonzero(tc["kf_idx"] == 0)[0][0]); i_po = int(np.nonzero(po["kf_idx"] == 0)[0][0]); i_tf = 0
    r, J = oracle.two_camera(tc["left_ob"][i_tc:i_tc + 1], tc["right_ob"][i_tc:i_tc + 1], tc["lm_idx"][i_tc:i_tc + 1], tc["kf_idx"][i_tc:i_tc + 1], cfg["inv_depth"], cfg["w_kf"], c0, c1)
    assert_parity(probe[:4], np.concatenate([r[0], J[0]]), "TwoCamera Evaluate"); probe = probe[4:]
    r, J = oracle.pose_only(po["ob"][i_po:i_po + 1], po["kf_idx"][i_po:i_po + 1], po["pw_idx"][i_po:i_po + 1], po["pw"], cfg["poses"], cfg["w_kf"], c0)
    assert_parity(probe[:16], np.concatenate([r[0], J[0].ravel()]), "PoseOnly Evaluate"); probe = probe[16:]
    r, Jd, J1, J2 = oracle.two_frame(tf["first_ob"][:1], tf["ob"][:1], tf["lm_idx"][:1], tf["kf1_idx"][:1], tf["kf2_idx"][:1], cfg["inv_depth"], cfg["poses"], cfg["w_kf"], c0, c1)
    assert_parity(probe[:32], np.concatenate([r[0], Jd[0], np.zeros(14), J2[0].ravel()]), "TwoFrame Evaluate (jacobians[1] NULL)"); probe = probe[32:]
    if with_imu:
        r, J = oracle.imu_eval(pre[:1], [0], [1], cfg["poses"], cfg["vel"], cfg["ba"], cfg["bg"])
        Js = oracle.imu_split_jac(J)
        exp_imu = np.concatenate([r[0]] + [Js[k][0].ravel() if k != 1 else np.zeros(45) for k in range(8)])
        assert_parity(probe, exp_imu, "ImuError Evaluate (jacobians[1] NULL)")
    for h in (prob, btc, btf, bpo, bimu, bpr, st):
        if h is not None:
            h.close()
    ctx.close()


@pytest.mark.parametrize("mode,use_prior", [(0, 1), (1, 0)])
def test_scan_to_map_through_adapter(tmp_path, oracle, mode, use_prior):
    from lvio_fusion_amd import api
    c = syn.config3_icp()
    sel = np.sort(np.random.default_rng(5).choice(c["query"].shape[0], 4000, replace=False))
    q, qg = c["query"][sel], c["query_ground"][sel]
    qq = q[qg] if mode == 0 else q[~qg]
    mm = c["map"][c["map_ground"]] if mode == 0 else c["map"][~c["map_ground"]]
    thr = c["thr_ground"] if mode == 0 else c["thr_surf"]
    w = syn.W_LIDAR_GROUND if mode == 0 else syn.W_LIDAR_SURF
    huber = 0.0 if mode == 0 else 0.1
    prior = 300 * syn.W_VISUAL if use_prior else 0.0
    rpyxyz0 = oracle.se3_to_rpyxyz(oracle.se3_mul(oracle.se3_inv(c["map_pose"]), c["pose0"]))
    ref_x, ref = oracle.icp_solve(mm, qq, c["map_pose"], c["pose0"], rpyxyz0, mode, thr, w, huber, prior_w=prior)
    # association stays where the reference has it (host loop over the 3-NN result); here the device 3-NN provides it
    ctx = api.Context(0)
    mp, sc = api.Map(ctx, mm, thr), api.Scan(ctx, qq)
    api.knn3(mp, sc, c["pose0"], thr)
    idx, d2, valid = sc.download()
    v = valid > 0
    p = qq[v, :3].astype(np.float64)
    pa, pb, pc = (mm[idx[v, k], :3].astype(np.float64) for k in range(3))
    mp.close(); sc.close(); ctx.close()
    d = str(tmp_path)
    _dump(d, "meta.i32", [mode, use_prior], np.int32)
    _dump(d, "scalars.f64", [w, huber, prior], np.float64)
    for name, arr in (("p", p), ("pa", pa), ("pb", pb), ("pc", pc), ("map_pose", c["map_pose"]), ("rpyxyz", rpyxyz0)):
        _dump(d, name + ".f64", arr, np.float64)
    out = _run("lidar", d)
    assert out["ok"] == 1
    assert out["num_residual_blocks"] == ref["num_residual_blocks"]
    assert out["successful"] == ref["num_successful_steps"]
    assert abs(out["initial_cost"] - ref["initial_cost"]) <= 1e-9 * abs(ref["initial_cost"])
    assert abs(out["final_cost"] - ref["final_cost"]) <= 1e-6 * abs(ref["final_cost"])
    x = np.fromfile(os.path.join(d, "out_rpyxyz.f64"))
    assert np.allclose(x, ref_x, rtol=1e-6, atol=1e-9)
    probe = np.fromfile(os.path.join(d, "out_probe.f64"))
    nrm = oracle.plane_normals(pa[:1], pb[:1], pc[:1])
    r, J = oracle.lidar_plane(mode, p[:1], pa[:1], nrm, c["map_pose"], rpyxyz0, w)
    assert_parity(probe[:4], np.concatenate([r.ravel(), J.ravel()]), "LidarPlaneError Evaluate")
    if use_prior:
        s = [1, 2, 5] if mode == 0 else [0, 3, 4]
        xx = rpyxyz0.copy(); xx[s[0]] += 0.01; xx[s[1]] -= 0.02; xx[s[2]] += 0.03
        r3, J3 = oracle.prior3(mode, rpyxyz0, prior, xx)
        assert_parity(probe[4:7], r3, "PoseErrorRPZ/YXY residuals")
        assert_parity(probe[7:16].reshape(3, 3).T, J3, "PoseErrorRPZ/YXY jacobians")


def test_foreign_cost_function_fails_soft():
    out = _run("foreign")
    assert out["ok"] == 0 and out["x"] == 3.0 and "not an lvio_fusion::gpu cost function" in out["message"]


def test_recorded_window_equals_the_accessor_walk(tmp_path, oracle):
    """adapt::Problem records the SoA payload while blocks are added (gpu::Recorder); gpu::Solve then skips the walk through the Ceres
    accessors.  Both paths must hand the device the same window: identical summaries and written-back parameters, with IMU blocks, weak
    priors and a constant pose in play (LVF_ADAPTER_WALK=1 forces the walk)."""
    n_kf, n_lm, max_it = 9, 200, 5
    for with_imu, weak_thr, const_kf in ((True, 0, 0), (False, 10 ** 6, -1)):
        cfg = _sorted_by_kf(syn.config4_window(n_kf=n_kf, n_lm=n_lm, n_prewindow=40, seed=123, imu_samples=4))
        pre = np.stack([oracle.imu_preintegrate(f["samples"], f["acc0"], f["gyr0"], f["ba"], f["bg"], syn.IMU_NOISE) for f in cfg["imu"]])
        outs = []
        for sub, env in (("rec", {}), ("walk", {"LVF_ADAPTER_WALK": "1"})):
            d = str(tmp_path / f"{sub}{int(with_imu)}"); os.makedirs(d)
            tc, tf, po = cfg["tc"], cfg["tf"], cfg["po"]
            _dump(d, "meta.i32", [n_kf, n_lm, max_it, weak_thr, const_kf], np.int32)
            for name, key in (("poses", "poses"), ("vel", "vel"), ("ba", "ba"), ("bg", "bg"), ("inv_depth", "inv_depth"), ("w_kf", "w_kf")):
                _dump(d, name + ".f64", cfg[key], np.float64)
            _dump(d, "cam0.f64", _cam_vec(cfg["cam0"]), np.float64); _dump(d, "cam1.f64", _cam_vec(cfg["cam1"]), np.float64)
            _dump(d, "tc_left_ob.f64", tc["left_ob"], np.float64); _dump(d, "tc_right_ob.f64", tc["right_ob"], np.float64)
            _dump(d, "tc_lm.i32", tc["lm_idx"], np.int32); _dump(d, "tc_kf.i32", tc["kf_idx"], np.int32)
            _dump(d, "tf_first_ob.f64", tf["first_ob"], np.float64); _dump(d, "tf_ob.f64", tf["ob"], np.float64)
            _dump(d, "tf_lm.i32", tf["lm_idx"], np.int32); _dump(d, "tf_kf1.i32", tf["kf1_idx"], np.int32); _dump(d, "tf_kf2.i32", tf["kf2_idx"], np.int32)
            _dump(d, "po_ob.f64", po["ob"], np.float64); _dump(d, "po_pw.f64", po["pw"], np.float64)
            _dump(d, "po_kf.i32", po["kf_idx"], np.int32); _dump(d, "po_pw_idx.i32", po["pw_idx"], np.int32)
            imu = cfg["imu"] if with_imu else []
            _dump(d, "preint.f64", pre if with_imu else np.zeros(0), np.float64)
            _dump(d, "imu_i.i32", [f["kf_i"] for f in imu], np.int32); _dump(d, "imu_j.i32", [f["kf_j"] for f in imu], np.int32)
            e = dict(os.environ); e.update(env); e["LVF_ADAPTER_TIMING"] = "1"
            p = subprocess.run([EXE, "window", d], capture_output=True, text=True, timeout=300, env=e)
            assert p.returncode == 0, p.stdout + p.stderr
            assert ("recorded window" in p.stderr) == (sub == "rec"), p.stderr            # the path under test really ran
            o = json.loads(p.stdout.strip().splitlines()[-1])
            outs.append((o, {f: np.fromfile(os.path.join(d, f)) for f in ("out_poses.f64", "out_inv_depth.f64", "out_vel.f64", "out_ba.f64", "out_bg.f64")}))
        (a, fa), (b, fb) = outs
        for k in ("ok", "successful", "num_residual_blocks", "n_prior", "num_frames"):
            assert a[k] == b[k], (k, a[k], b[k])
        for k in ("initial_cost", "final_cost", "cost0"):       # (device sums through atomics: equal up to summation order)
            assert abs(a[k] - b[k]) <= 1e-12 * abs(b[k]), (k, a[k], b[k])
        for f in fa:
            assert np.allclose(fa[f], fb[f], rtol=1e-10, atol=1e-13), f


@pytest.mark.parametrize("with_imu", [True, False])
def test_environment_optimize_through_adapter(tmp_path, oracle, with_imu):
    """Environment::Optimize's visual + IMU solve (src/environment.cpp:18-75) through adapt::Solve: one free pose, PoseOnly blocks under
    HuberLoss(1.0), one ImuError whose other seven parameter blocks (the previous keyframe's pose / v / ba / bg and the frame's own
    v / ba / bg) are held constant.  Against the oracle's ceres::Solve restatement with the same constant masks; the constant blocks must
    come back untouched."""
    from lvio_fusion_amd import api
    cfg = syn.config4_window(n_kf=6, n_lm=40, n_prewindow=300, seed=90, imu_samples=6)
    po = cfg["po"]
    kf = 4                                                # the environment's frame; its previous keyframe is kf - 1
    sel = np.flatnonzero(po["kf_idx"] == kf)
    assert len(sel) >= 20, "generator changed: no PoseOnly blocks on the chosen keyframe"
    ob, pw = po["ob"][sel], po["pw"][po["pw_idx"][sel]]
    f = [x for x in cfg["imu"] if x["kf_j"] == kf][0]
    pre = oracle.imu_preintegrate(f["samples"], f["acc0"], f["gyr0"], f["ba"], f["bg"], syn.IMU_NOISE)
    d = str(tmp_path)
    _dump(d, "meta.i32", [len(sel), 50, int(with_imu)], np.int32)
    _dump(d, "pose.f64", cfg["poses"][kf], np.float64); _dump(d, "last_pose.f64", cfg["poses"][kf - 1], np.float64)
    vbb = np.concatenate([cfg["vel"][kf], cfg["ba"][kf], cfg["bg"][kf], cfg["vel"][kf - 1], cfg["ba"][kf - 1], cfg["bg"][kf - 1]])
    _dump(d, "vbb.f64", vbb, np.float64)
    _dump(d, "ob.f64", ob, np.float64); _dump(d, "pw.f64", pw, np.float64); _dump(d, "weight.f64", [cfg["w_kf"][kf]], np.float64)
    _dump(d, "cam0.f64", _cam_vec(cfg["cam0"]), np.float64); _dump(d, "preint.f64", pre, np.float64)
    out = _run("environment", d)
    assert out["ok"] == 1 and out["constant_blocks_untouched"] == 1 and out["recorder_used"] == 1
    assert out["num_residual_blocks"] == len(sel) + int(with_imu)
    # the oracle: a two-keyframe window (0 = the frame, 1 = the previous keyframe: registration order), everything but pose 0 constant
    w2 = dict(n_kf=2, n_lm=1, poses=np.stack([cfg["poses"][kf], cfg["poses"][kf - 1]]), vel=np.stack([cfg["vel"][kf], cfg["vel"][kf - 1]]),
              ba=np.stack([cfg["ba"][kf], cfg["ba"][kf - 1]]), bg=np.stack([cfg["bg"][kf], cfg["bg"][kf - 1]]), inv_depth=np.ones(1), w_kf=np.full(2, cfg["w_kf"][kf]),
              cam0=cfg["cam0"], cam1=cfg["cam1"],
              tc=dict(left_ob=np.zeros((0, 2)), right_ob=np.zeros((0, 2)), lm_idx=np.zeros(0, np.int32), kf_idx=np.zeros(0, np.int32)),
              tf=dict(first_ob=np.zeros((0, 2)), ob=np.zeros((0, 2)), lm_idx=np.zeros(0, np.int32), kf1_idx=np.zeros(0, np.int32), kf2_idx=np.zeros(0, np.int32)),
              po=dict(ob=ob, kf_idx=np.zeros(len(sel), np.int32), pw_idx=np.arange(len(sel), dtype=np.int32), pw=pw),
              imu=[dict(kf_i=1, kf_j=0)] if with_imu else [])
    win = oracle.Window(w2, pre.reshape(1, -1) if with_imu else np.zeros((0, 467)), pose_const=np.array([0, 1], np.uint8), vbb_const=np.array([7, 7], np.uint8),
                        use=("po", "imu") if with_imu else ("po",))
    ref = win.solve(max_num_iterations=50)
    assert abs(out["initial_cost"] - ref["initial_cost"]) <= 1e-9 * ref["initial_cost"]
    assert abs(out["final_cost"] - ref["final_cost"]) <= 1e-6 * ref["final_cost"]
    assert (out["successful"], out["unsuccessful"], out["termination"] == 0) == (ref["num_successful_steps"], ref["num_unsuccessful_steps"], ref["termination"] == 0), (out, ref["why"])
    assert_parity(np.fromfile(os.path.join(d, "out_pose.f64")), win.poses[0], "the environment frame's pose")
    assert ref["final_cost"] < ref["initial_cost"] and ref["num_successful_steps"] >= 2


def test_relocate_rotation_through_adapter(tmp_path, oracle):
    """Relocator::UpdateNewSubmap's problem (relocator.cpp:251-268) built with gpu::RelocateRError::Create and solved by adapt::Solve -> gpu::Solve ->
    lvf_relocate_rotation_solve; CostFunction::Evaluate against the oracle's autodiff; a mixed problem is refused softly"""
    rng = np.random.default_rng(21)
    m = 7
    un = np.zeros((m, 7)); un[:, :4] = syn.quat_from_ypr(*rng.normal(0, 0.3, (3, m))); un[:, 4:] = rng.normal(0, 5, (m, 3))
    R = np.concatenate([syn.quat_from_ypr(0.05, -0.02, 0.03), [0, 0, 0]])
    rel = syn.se3_mul(np.tile(R, (m, 1)), un) + rng.normal(0, 1e-3, (m, 7))
    d = str(tmp_path)
    _dump(d, "relocated.f64", rel, np.float64); _dump(d, "unrelocated.f64", un, np.float64)
    out = _run("relocate", d)
    assert out["ok"] == 1 and out["n"] == m and out["mixed_problem_refused_softly"] == 1
    q_ref, s_ref = oracle.relocate_rotation_solve(rel, un, [0, 0, 0, 1.0])
    q = np.fromfile(os.path.join(d, "out_q.f64"))
    assert np.abs(q - q_ref).max() <= 1e-9
    probe = np.fromfile(os.path.join(d, "out_probe.f64"))
    r0, J0 = oracle.relocate_r(rel[0], un[0], [0.02, -0.01, 0.03, 0.98])
    assert_parity(probe[:7], r0, "RelocateRError::Evaluate r"); assert_parity(probe[7:35].reshape(7, 4), J0, "RelocateRError::Evaluate J")
    assert abs(probe[35] - s_ref["initial_cost"]) <= 1e-9 * s_ref["initial_cost"] and abs(probe[36] - s_ref["final_cost"]) <= 1e-6 * max(s_ref["final_cost"], 1e-12)
    assert int(probe[37]) == s_ref["num_successful_steps"] and int(probe[38]) == m
