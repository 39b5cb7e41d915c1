"""The COMPILED drop-in (north_star: "backend.cpp's solver loop calls it as a drop-in"):
oracle/_ref/liblvf_dropin.so is the reference's backend.cpp / association.cpp / landmark.cpp / preintegration.cpp compiled UNMODIFIED with
include/reference_patch ahead of the reference's include directory (oracle/Makefile `dropin`, built in the build container where
/root/reference exists; the library travels with the snapshot).  Here `Backend::BuildProblem -> adapt::Solve` and
`FeatureAssociation::ScanToMapWith{Ground,Segmented} -> adapt::Solve` run on the MI355X from the reference's own text, and what they
leave in the caller-owned parameter arrays is compared with
  * lvf_window_solve on the same ticks of tests/window_replay.py's 12-keyframe drive (same kernels below, different host path), and
  * the oracle's LM loop (oracle/lm.h) on the block lists of the same tick / the oracle's ICP (oracle/icp.h)
to 1e-6 relative (tests/helpers.assert_parity)."""
import numpy as np
import pytest

from lvio_fusion_amd import synthetic as syn
from tests import mapping_replay as mr
from tests import window_replay as wr
from tests.helpers import assert_parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dropin():
    from oracle import pydropin
    if not pydropin.available():
        pytest.skip("oracle/_ref/liblvf_dropin.so was not built (needs /root/reference at build time)")
    pydropin.lib()
    return pydropin


def _window_at_tick(api, ctx, drive, t_stop, max_it, pre):
    """lvf_window_* replayed to tick t_stop (0 iterations on the way: the drive never depends on a solver result), then solved with max_it"""
    cfg = drive.cfg
    win = api.Window(ctx, cfg["cam0"], cfg["cam1"], baseline=syn.baseline(), device_assembly=True)
    for t in range(t_stop + 1):
        ev, first = drive.tick(t)
        for e in ev:
            if e[0] == "kf":
                win.add_keyframe(wr.KF_ID0 + t, cfg["poses"][t], drive.w_kf[t])
                if drive.with_imu:
                    win.set_imu(wr.KF_ID0 + t, cfg["vel"][t], cfg["ba"][t], cfg["bg"][t], pre[t - 1] if t > 0 else None)
            elif e[0] == "lm":
                l = e[1]
                win.add_landmark(int(drive.lm_id[l]), wr.KF_ID0 + t, drive.left_ob[l], drive.right_ob[l], drive.inv_depth[l])
            elif e[0] == "ob":
                win.add_observation(int(drive.lm_id[e[1]]), wr.KF_ID0 + t, e[3])
            else:
                win.remove_observation(int(drive.lm_id[e[1]]), wr.KF_ID0 + e[2])
        win.slide(wr.KF_ID0 + first)
        opt = api.default_solver_options(); opt.max_num_iterations = max_it if t == t_stop else 0
        s = win.solve(opt)
    return win, first, s


def _oracle_window(oracle, drive, lists, first, t, pre):
    """the tick's block lists (lvf_window_debug_blocks: pinned bit for bit to the reference's BuildProblem by test_gpu_window.py) as an oracle Window"""
    cfg = drive.cfg
    kf_of = {wr.KF_ID0 + k: k - first for k in range(first, t + 1)}
    n_kf = t - first + 1
    lm_ids = sorted({int(i) for kind in ("TwoCamera", "TwoFrame") for i in lists[kind]["ids"][:, 0]})
    lm_of = {i: n for n, i in enumerate(lm_ids)}
    id2l = {int(i): l for l, i in enumerate(drive.lm_id)}
    tc, tf, po = lists["TwoCamera"], lists["TwoFrame"], lists["PoseOnly"]
    w = dict(n_kf=n_kf, n_lm=len(lm_ids), cam0=cfg["cam0"], cam1=cfg["cam1"], poses=cfg["poses"][first:t + 1], vel=cfg["vel"][first:t + 1], ba=cfg["ba"][first:t + 1],
             bg=cfg["bg"][first:t + 1], inv_depth=np.array([drive.inv_depth[id2l[i]] for i in lm_ids]), w_kf=drive.w_kf[first:t + 1])
    w["tc"] = dict(left_ob=tc["vals"][:, 1:3], right_ob=tc["vals"][:, 3:5], lm_idx=[lm_of[int(i)] for i in tc["ids"][:, 0]], kf_idx=[kf_of[int(k)] for k in tc["ids"][:, 2]])
    w["tf"] = dict(first_ob=tf["vals"][:, 3:5], ob=tf["vals"][:, 1:3], lm_idx=[lm_of[int(i)] for i in tf["ids"][:, 0]], kf1_idx=[kf_of[int(k)] for k in tf["ids"][:, 1]],
                   kf2_idx=[kf_of[int(k)] for k in tf["ids"][:, 2]])
    w["po"] = dict(ob=po["vals"][:, 1:3], kf_idx=[kf_of[int(k)] for k in po["ids"][:, 2]], pw_idx=np.arange(len(po["ids"])), pw=po["vals"][:, 5:8].reshape(-1, 3))
    imu_ids = lists["ImuError"]["ids"]
    w["imu"] = [dict(kf_i=kf_of[int(a)], kf_j=kf_of[int(b)]) for a, b in imu_ids[:, 1:3]]
    opre = np.stack([pre[int(b) - wr.KF_ID0 - 1] for b in imu_ids[:, 2]]) if len(imu_ids) else np.zeros((0, oracle.PREINT_DOUBLES))
    priors = None
    pg, pe = lists["PoseGraphError"], lists["PoseError"]
    if len(pg["ids"]) + len(pe["ids"]):
        ka, kb, tgt, wt, vv = [], [], [], [], []
        for ids, vals in zip(pe["ids"], pe["vals"]):
            k = kf_of[int(ids[2])]
            ka.append(-1); kb.append(k); tgt.append(w["poses"][k]); wt.append(vals[0]); vv.append(vals[1])
        for ids, vals in zip(pg["ids"], pg["vals"]):
            a, b = kf_of[int(ids[1])], kf_of[int(ids[2])]
            ka.append(a); kb.append(b); tgt.append(np.concatenate([oracle.pose_graph_target(w["poses"][a], w["poses"][b]), [0.0]])); wt.append(vals[0]); vv.append(vals[1])
        priors = dict(kf_a=ka, kf_b=kb, target=np.array(tgt), weight=np.array(wt), v=np.array(vv))
    use = tuple(k for k, n in (("tc", len(tc["ids"])), ("tf", len(tf["ids"])), ("po", len(po["ids"])), ("imu", len(imu_ids))) if n)
    return oracle.Window(w, opre, use=use, priors=priors), lm_ids


@pytest.mark.parametrize("with_imu", [True, False])
@pytest.mark.parametrize("t", [0, 1, 3, 7, 11])      # t = 0: a one-keyframe window (TwoCamera blocks + the weak-constraint PoseError only)
def test_build_problem_and_solve_from_the_reference_text(dropin, oracle, with_imu, t):
    from lvio_fusion_amd import api
    K = 4
    drive = wr.Drive(with_imu)
    cfg = drive.cfg
    pre = [oracle.imu_preintegrate(f["samples"], f["acc0"], f["gyr0"], f["ba"], f["bg"], syn.IMU_NOISE) for f in cfg["imu"]] if with_imu else None
    ctx = api.Context(0)
    win, first, s_win = _window_at_tick(api, ctx, drive, t, K, pre)
    lists = wr.window_lists(win)
    # ---- the reference's text on the same tick: Backend::BuildProblem -> adapt::Solve -> gpu::Solve
    inp, live = drive.reference_input(t, first)
    out = dropin.backend_solve(cfg["cam0"], cfg["cam1"], syn.baseline(), max_num_iterations=K, vel=cfg["vel"][:t + 1], ba=cfg["ba"][:t + 1], bg=cfg["bg"][:t + 1],
                               imu=([None] + [cfg["imu"][k - 1] for k in range(1, t + 1)]) if with_imu else None, imu_noise=syn.IMU_NOISE, **inp)
    assert out["rc"] == 0, out["message"]
    assert out["recorded"], "the recorder hooks of reference_patch/lvio_fusion/adapt/problem.h did not follow BuildProblem"
    assert out["num_frames"] == t - first + 1
    n_blocks = sum(len(lists[k]["ids"]) for k in wr.KINDS)
    assert out["num_residual_blocks"] == n_blocks
    assert out["num_successful_steps"] == s_win.num_successful_steps and out["num_successful_steps"] >= 1
    assert abs(out["initial_cost"] - s_win.initial_cost) <= 1e-9 * s_win.initial_cost and abs(out["final_cost"] - s_win.final_cost) <= 1e-6 * s_win.final_cost
    assert out["final_cost"] < out["initial_cost"]
    # ---- (1) against lvf_window_solve
    w_pose = np.stack([win.pose(wr.KF_ID0 + k) for k in range(first, t + 1)])
    assert_parity(out["pose"][first:], w_pose, "poses in frame->pose vs lvf_window")
    assert np.array_equal(out["pose"][:first], cfg["poses"][:first]), "departed frames are not in the problem: untouched"
    w_invd = np.array([win.inv_depth(int(drive.lm_id[l])) for l in live])
    assert_parity(out["inv_depth"], w_invd, "inverse depths in landmark->inv_depth vs lvf_window")
    if with_imu:
        w_imu = [win.imu(wr.KF_ID0 + k) for k in range(first, t + 1)]
        for j, name in enumerate(("vel", "ba", "bg")):
            assert_parity(out[name][first:], np.stack([x[j] for x in w_imu]), f"{name} vs lvf_window")
    # ---- (2) against the oracle's LM loop on the same block lists
    ow, lm_ids = _oracle_window(oracle, drive, lists, first, t, pre)
    o = api.default_solver_options()
    ref = ow.solve(max_num_iterations=K, huber_a=o.huber_a, initial_trust_region_radius=o.initial_trust_region_radius, function_tolerance=o.function_tolerance,
                   gradient_tolerance=o.gradient_tolerance, parameter_tolerance=o.parameter_tolerance, min_relative_decrease=o.min_relative_decrease)
    assert ref["num_successful_steps"] == out["num_successful_steps"]
    assert abs(out["final_cost"] - ref["final_cost"]) <= 1e-6 * ref["final_cost"]
    assert_parity(out["pose"][first:], ow.poses, "poses vs the oracle")
    id2row = {int(drive.lm_id[l]): i for i, l in enumerate(live)}
    assert_parity(np.array([out["inv_depth"][id2row[i]] for i in lm_ids]), ow.inv_depth, "inverse depths vs the oracle")
    if with_imu:
        assert_parity(out["vel"][first:], ow.vel, "velocities vs the oracle")
    win.close(); ctx.close()


@pytest.mark.parametrize("mode,relocate", [(0, False), (1, False), (0, True), (1, True)])
def test_scan_to_map_from_the_reference_text(dropin, oracle, mode, relocate):
    """association.cpp:270-384 (the reference's kd-tree loop, gates and Create calls) + mapping.cpp:158-164's solve through adapt::Solve on the GPU,
    against lvf_icp_solve (device association + the same LM) and the oracle's ICP (oracle/icp.h)"""
    from lvio_fusion_amd import api
    c = syn.config3_icp(n_query=8000, n_az=700)
    qm, mm = (c["query_ground"], c["map_ground"]) if mode == 0 else (~c["query_ground"], ~c["map_ground"])
    scan = np.ascontiguousarray(c["query"][qm][:2500], np.float32); mp = np.ascontiguousarray(c["map"][mm][:12000], np.float32)
    thr = c["thr_ground"] if mode == 0 else c["thr_surf"]
    res = 0.2
    assert abs(thr - res * res * (100 if mode == 0 else 25)) < 1e-12
    n_feat = 40
    prior = 0.0 if relocate else n_feat * syn.W_VISUAL        # association.cpp:323 / :381
    w = syn.W_LIDAR_GROUND if mode == 0 else syn.W_LIDAR_SURF
    huber = 0.0 if mode == 0 else 0.1
    rp0 = oracle.se3_to_rpyxyz(oracle.se3_mul(oracle.se3_inv(c["map_pose"]), c["pose0"]))
    out = dropin.scan_to_map_solve(mode, scan, mp, c["pose0"], c["map_pose"], rp0, syn.W_LIDAR_GROUND, syn.W_LIDAR_SURF, syn.W_VISUAL, n_feat, relocate, res, 4)
    assert out["rc"] == 0, out["message"]
    assert out["n_lidar"] > 300
    # ---- lvf_icp_solve on the same clouds
    ctx = api.Context(0)
    m, sc = api.Map(ctx, mp, thr), api.Scan(ctx, scan)
    x = rp0.copy()
    s = api.icp_solve(m, sc, c["map_pose"], c["pose0"], x, mode, thr, w, huber, prior_weight=prior)
    assert s.num_residual_blocks == out["n_lidar"] + (0 if relocate else 1), "the reference's kd-tree loop and the device association accepted different points"
    assert_parity(out["para"], x, "rpyxyz written in place vs lvf_icp_solve")
    assert abs(out["final_cost"] - s.final_cost) <= 1e-6 * max(abs(s.final_cost), 1e-12)
    untouched = [0, 3, 4] if mode == 0 else [1, 2, 5]
    assert np.array_equal(out["para"][untouched], rp0[untouched])
    # ---- the oracle's restatement
    ref_x, ref = oracle.icp_solve(mp, scan, c["map_pose"], c["pose0"], rp0, mode, thr, w, huber, prior_w=prior)
    assert np.allclose(out["para"], ref_x, rtol=1e-6, atol=1e-9)
    assert abs(out["final_cost"] - ref["final_cost"]) <= 1e-6 * abs(ref["final_cost"])
    m.close(); sc.close(); ctx.close()


def test_baseline_window_through_the_reference_text(dropin):
    """the BASELINE window itself (50 keyframes, 10 000 landmarks, 81 839 residual blocks): the reference's own Backend::BuildProblem -> adapt::Solve
    (3 LM iterations on the MI355X) against lvf_problem_solve on the flat batches of the same blocks"""
    from lvio_fusion_amd import api
    from tests.dropin_tick import baseline_window_inputs
    K = 3
    cfg, cams, args = baseline_window_inputs()
    out = dropin.backend_solve(cams["cam0"], cams["cam1"], syn.baseline(), max_num_iterations=K, **args)
    assert out["rc"] == 0, out["message"]
    assert out["recorded"] and out["num_frames"] == cfg["n_kf"]
    f32 = lambda x: np.asarray(x, np.float32).astype(np.float64)
    tc, tf = cfg["tc"], cfg["tf"]
    # BuildProblem's block census (backend.cpp:112-178): a TwoCamera block per landmark at its birth keyframe, a TwoFrame block per later observation,
    # an ImuError per consecutive pair, and ONE weak-constraint PoseError: keyframe 0 has no ImuError yet and no VisualError block when it is checked
    assert out["num_residual_blocks"] == len(tc["lm_idx"]) + len(tf["lm_idx"]) + (cfg["n_kf"] - 1) + 1
    ctx = api.Context(0)
    c2 = dict(cfg); c2["cam0"], c2["cam1"] = cams["cam0"], cams["cam1"]
    pre = api.preintegrate_or_none(ctx, c2)
    st = api.State(ctx, cfg["n_kf"], cfg["n_lm"])
    for field, val in ((api.POSES, cfg["poses"]), (api.VEL, cfg["vel"]), (api.BA, cfg["ba"]), (api.BG, cfg["bg"]), (api.INV_DEPTH, cfg["inv_depth"]), (api.W_VISUAL, f32(cfg["w_kf"]))):
        st.set(field, val)
    o = np.argsort(tc["kf_idx"], kind="stable")
    hs = [api.two_camera_batch(ctx, cams["cam0"], cams["cam1"], f32(tc["left_ob"])[o], f32(tc["right_ob"])[o], tc["lm_idx"][o], tc["kf_idx"][o]),
          api.two_frame_batch(ctx, cams["cam0"], cams["cam1"], f32(tf["first_ob"]), f32(tf["ob"]), tf["lm_idx"], tf["kf1_idx"], tf["kf2_idx"]),
          None, api.imu_batch(ctx, pre, [f["kf_i"] for f in cfg["imu"]], [f["kf_j"] for f in cfg["imu"]])]
    prob = api.Problem(ctx, st, *hs)
    bpr = api.pose_prior_batch(ctx, [-1], [0], np.array([cfg["poses"][0]]), np.array([100.0]), np.array([0.0]))
    prob.set_pose_priors(bpr)
    opt = api.default_solver_options(); opt.max_num_iterations = K
    s = prob.solve(opt)
    assert out["num_successful_steps"] == s.num_successful_steps and s.num_successful_steps >= 2
    assert abs(out["initial_cost"] - s.initial_cost) <= 1e-7 * s.initial_cost and abs(out["final_cost"] - s.final_cost) <= 1e-6 * s.final_cost
    assert_parity(out["pose"], st.get(api.POSES).reshape(-1, 7), "poses")
    assert_parity(out["inv_depth"], st.get(api.INV_DEPTH), "inverse depths")
    assert_parity(out["vel"], st.get(api.VEL).reshape(-1, 3), "velocities")
    assert_parity(out["ba"], st.get(api.BA).reshape(-1, 3), "ba"); assert_parity(out["bg"], st.get(api.BG).reshape(-1, 3), "bg")
    prob.close(); bpr.close()
    for h in hs + [st]:
        if h is not None:
            h.close()
    ctx.close()


# ---- the reference's CONTROL code (mapping.cpp, pose_graph.cpp, relocator.cpp: compiled unmodified into the drop-in library as well) with every
# ceres::Solve / adapt::Solve on the MI355X, against what the SAME text left behind on the CPU with the declared LM loop (tests/golden/ref_v5.npz)
def _r5():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_v5.npz"))


def _close(a, b, tol=1e-6):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return a.shape == b.shape and np.all(np.abs(a - b) <= tol * (1e-3 + np.abs(b)))


def test_mapping_optimize_from_the_reference_text_on_the_gpu(dropin):
    from tests import mapping_replay as mr
    c = mr.optimize_case()
    r = dropin.mapping_optimize(c["time"], c["pose"], c["ground"], c["surf"], c["first_active"], c["w_ground"], c["w_surf"], c["w_visual"], c["n_features_left"])
    R5 = _r5()
    assert _close(r["pose"], R5["optimize_pose"]), np.abs(r["pose"] - R5["optimize_pose"]).max(axis=1)
    assert np.array_equal(r["world_counts"], R5["optimize_world_counts"])


@pytest.mark.parametrize("name,kw", mr.RELOCATE_CASES)
def test_mapping_relocate_from_the_reference_text_on_the_gpu(dropin, name, kw):
    from tests import mapping_replay as mr
    c = mr.relocate_case(**kw)
    r = dropin.mapping_relocate(c["time"], c["pose"], c["ground"], c["surf"], c["old_index"], c["cur_ground"], c["cur_surf"], c["cur_pose"], c["rel_in"],
                                c["w_ground"], c["w_surf"], c["w_visual"])
    R5 = _r5()
    assert r["score"] == int(R5[name + "_score"])
    assert _close(r["relative_o_c"], R5[name + "_relative_o_c"]) and np.array_equal(r["map_counts"], R5[name + "_map_counts"])


def test_pose_graph_optimize_from_the_reference_text_on_the_gpu(dropin):
    from tests import mapping_replay as mr
    c = mr.pose_graph_case()
    r = dropin.pose_graph_optimize(c["time"], c["pose"], c["vw"], c["section_A"], c["submap_A"], c["submap_B"], c["start_after"])
    R5 = _r5()
    assert _close(r["pose"], R5["pose_graph_pose"]), np.abs(r["pose"] - R5["pose_graph_pose"]).max(axis=1)
    assert _close(r["vw"], R5["pose_graph_vw"])
    ns = len(c["section_A"])
    assert r["counts"] == (2 * ns + 1, ns + 2, ns + 2)


def test_update_new_submap_from_the_reference_text_on_the_gpu(dropin):
    from tests import mapping_replay as mr
    c = mr.submap_case()
    P = dropin.update_new_submap(c["time"], c["pose"], c["old_pose"], c["relative_o_c"], c["best"])
    assert _close(P, _r5()["submap_pose"]), np.abs(P - _r5()["submap_pose"]).max(axis=1)


def test_environment_optimize_from_the_reference_text_on_the_gpu(dropin):
    """Environment::Optimize (environment.cpp:18-115, compiled unmodified): the third adapt::Solve call site — the recorder follows its SetParameterBlockConstant
    calls, gpu::Solve holds the seven constant blocks"""
    from tests import mapping_replay as mr
    c = mr.environment_case()
    P = dropin.environment_optimize(c["cam0"], c["cam1"], c["baseline"], c["pose3"], c["vel3"], c["ba3"], c["bg3"], c["w_visual"], c["samples"], c["acc0"], c["gyr0"],
                                    c["noise4"], c["inv_depth"], c["right_ob"], c["left_ob"])
    assert _close(P, _r5()["environment_pose"]), np.abs(P - _r5()["environment_pose"])
