"""A scripted 12-keyframe drive through a 6-keyframe sliding window, shared by
  * tests/golden/make_ref_golden_backend.py — runs the REFERENCE's own Backend::BuildProblem (src/lvio_fusion/src/backend.cpp:96-183, compiled
    unmodified into oracle/_ref) on every tick and stores its block lists in tests/golden/ref_v4.npz,
  * tests/test_oracle_ref.py — checks the stored lists against the live reference (build container only),
  * tests/test_gpu_window.py — replays the same ticks through lvf_window_* and compares the device batches' block lists with them BIT FOR BIT.
The drive never depends on a solver result (the window is solved with max_num_iterations = 0), so the golden lists are a function of this file
and lvio_fusion_amd/synthetic.py alone.  Observations are rounded to float first: the reference keeps key points as cv::Point2f."""
import numpy as np

from lvio_fusion_amd import synthetic as syn

N_KF, WINDOW, N_LM, SEED = 12, 6, 150, 606
KF_ID0, LM_ID0 = 100, 5000
KINDS = ("TwoCamera", "PoseOnly", "TwoFrame", "ImuError", "PoseGraphError", "PoseError")


def f32(x):
    return np.asarray(x, np.float32).astype(np.float64)


class Drive:
    """Per tick t (0 .. N_KF - 1): keyframe t arrives with its landmarks and observations; at ticks 5 and 8 an older frame loses a feature
    (what the outlier gate does); the window then keeps the last WINDOW keyframes.  `with_imu`: every keyframe carries IMU state and each
    consecutive pair an ImuError block (Imu::Get()->initialized, frame->good_imu)."""

    def __init__(self, with_imu, seed=SEED):
        self.with_imu = with_imu
        self.cfg = cfg = syn.config4_window(n_kf=N_KF, n_lm=N_LM, n_prewindow=0, seed=seed, imu_samples=4)
        # the camera extrinsics as the reference holds them: Sophus::SE3d keeps a UNIT quaternion (estimator.cpp:36-83 builds them from a matrix
        # through SE3d's constructor, which normalises); kitti.yaml's 7-digit matrices give |q| - 1 = 2.7e-7 if taken over as they are, and the
        # reference's SE3d arithmetic (Landmark::ToWorld, Camera::Far) assumes |q| = 1
        for cam in ("cam0", "cam1"):
            c = dict(cfg[cam]); e = np.array(c["extrinsic"], np.float64); e[:4] /= np.linalg.norm(e[:4]); c["extrinsic"] = e; cfg[cam] = c
        tc, tf = cfg["tc"], cfg["tf"]
        rng = np.random.default_rng(seed + 1)
        # landmark ids are NOT in creation order: BuildProblem walks a frame's features in ascending landmark id (std::map)
        self.lm_id = LM_ID0 + rng.permutation(N_LM)
        self.birth = np.full(N_LM, -1, np.int32); self.birth[tc["lm_idx"]] = tc["kf_idx"]
        self.right_ob = np.zeros((N_LM, 2)); self.right_ob[tc["lm_idx"]] = f32(tc["right_ob"])
        self.left_ob = np.zeros((N_LM, 2)); self.left_ob[tc["lm_idx"]] = f32(tc["left_ob"])
        self.inv_depth = np.asarray(cfg["inv_depth"], np.float64)
        self.tf_ob = f32(tf["ob"])
        self.w_kf = f32(cfg["w_kf"])             # frame->weights.visual is a float (adapt/weights.h:10)
        self.obs = {k: {} for k in range(N_KF)}       # keyframe -> {landmark index: left ob}
        self.time = 10.0 + 0.5 * np.arange(N_KF)
        self.removed = []

    def tick(self, t):
        """events of tick t as a list of ('kf', t) / ('lm', l, t) / ('ob', l, t) / ('rm', l, k), then the first active keyframe"""
        cfg = self.cfg
        tf = cfg["tf"]
        ev = [("kf", t)]
        for l in np.flatnonzero(self.birth == t):
            ev.append(("lm", int(l), t)); self.obs[t][int(l)] = self.left_ob[l]
        for i in np.flatnonzero(tf["kf2_idx"] == t):
            l = int(tf["lm_idx"][i])
            ev.append(("ob", l, t, self.tf_ob[i])); self.obs[t][l] = self.tf_ob[i]
        if t in (5, 8):
            k_old = t - 2
            cand = [l for l in sorted(self.obs[k_old], key=lambda x: self.lm_id[x]) if self.birth[l] != k_old]
            if cand:
                l = cand[len(cand) // 2]
                ev.append(("rm", l, k_old)); del self.obs[k_old][l]; self.removed.append((l, k_old))
        return ev, max(0, t - WINDOW + 1)

    def reference_input(self, t, first):
        """the window of tick t as oracle/pyref.backend_build_problem's flat arrays (frames 0 .. t; [first:] active)"""
        cfg = self.cfg
        live = sorted({l for k in range(first, t + 1) for l in self.obs[k]})
        idx = {l: i for i, l in enumerate(live)}
        o_lm, o_fr, o_xy = [], [], []
        for k in range(first, t + 1):
            for l, ob in self.obs[k].items():
                o_lm.append(idx[l]); o_fr.append(k); o_xy.append(ob)
        return dict(time=self.time[:t + 1], pose=cfg["poses"][:t + 1], w_visual=self.w_kf[:t + 1], good_imu=np.full(t + 1, 1 if self.with_imu else 0, np.uint8),
                    first_active=first, imu_initialized=self.with_imu, lm_id=self.lm_id[live], lm_birth=self.birth[live], lm_inv_depth=self.inv_depth[live],
                    lm_right_ob=self.right_ob[live], obs_lm=o_lm, obs_frame=o_fr, obs_xy=np.array(o_xy).reshape(-1, 2)), live


def normalise_reference(rec_i, rec_d, live, lm_id):
    """the reference's insertion-ordered records -> per kind (ids [n][3] = landmark id, keyframe id a, keyframe id b; vals [n][8]; ProblemType [n])"""
    out = {}
    for kind in range(6):
        sel = rec_i[:, 0] == kind
        ri, rd = rec_i[sel], rec_d[sel]
        ids = np.full((len(ri), 3), -1, np.int64)
        if len(ri):
            has_lm = ri[:, 2] >= 0
            ids[has_lm, 0] = lm_id[np.asarray(live)[ri[has_lm, 2]]]
            ids[ri[:, 3] >= 0, 1] = KF_ID0 + ri[ri[:, 3] >= 0, 3]
            ids[ri[:, 4] >= 0, 2] = KF_ID0 + ri[ri[:, 4] >= 0, 4]
        if kind == 0 and len(ri):
            ids[:, 1] = -1
        out[KINDS[kind]] = dict(ids=ids, vals=rd.copy(), type=ri[:, 1].copy(), loss=ri[:, 5].copy())
    return out


def window_lists(win):
    """lvf_window_debug_blocks of the last solve in the same form (priors split into PoseGraphError / PoseError by their first keyframe)"""
    out = {}
    for kind, name in ((0, "TwoCamera"), (1, "PoseOnly"), (2, "TwoFrame"), (3, "ImuError")):
        ids, vals = win.debug_blocks(kind)
        out[name] = dict(ids=ids, vals=vals)
    ids, vals = win.debug_blocks(4)
    pg = ids[:, 1] >= 0
    out["PoseGraphError"] = dict(ids=ids[pg], vals=vals[pg]); out["PoseError"] = dict(ids=ids[~pg], vals=vals[~pg])
    return out


def replay_window(api, ctx, drive, device_assembly, oracle_preintegrate=None):
    """generator: (t, first, win) after every tick's lvf_window_solve(max_num_iterations = 0)"""
    cfg = drive.cfg
    win = api.Window(ctx, cfg["cam0"], cfg["cam1"], baseline=syn.baseline(), device_assembly=device_assembly)
    opt = api.default_solver_options(); opt.max_num_iterations = 0
    pre = None
    if drive.with_imu:
        pre = [oracle_preintegrate(f["samples"], f["acc0"], f["gyr0"], f["ba"], f["bg"], syn.IMU_NOISE) for f in cfg["imu"]]
    for t in range(N_KF):
        ev, first = drive.tick(t)
        for e in ev:
            if e[0] == "kf":
                win.add_keyframe(KF_ID0 + t, cfg["poses"][t], drive.w_kf[t])
                if drive.with_imu:
                    win.set_imu(KF_ID0 + t, cfg["vel"][t], cfg["ba"][t], cfg["bg"][t], pre[t - 1] if t > 0 else None)
            elif e[0] == "lm":
                l = e[1]
                win.add_landmark(int(drive.lm_id[l]), KF_ID0 + t, drive.left_ob[l], drive.right_ob[l], drive.inv_depth[l])
            elif e[0] == "ob":
                win.add_observation(int(drive.lm_id[e[1]]), KF_ID0 + t, e[3])
            else:
                win.remove_observation(int(drive.lm_id[e[1]]), KF_ID0 + e[2])
        win.slide(KF_ID0 + first)
        win.solve(opt)
        yield t, first, win
    win.close()
