"""Scripted cases for the reference's CONTROL code around the hot path, shared by
  * tests/golden/make_ref_golden_mapping.py — runs the REFERENCE's own Mapping::Optimize / Mapping::Relocate (src/mapping.cpp), PoseGraph::BuildProblem /
    Optimize (src/pose_graph.cpp) and Relocator::UpdateNewSubmap (src/relocator.cpp), compiled unmodified into oracle/_ref (oracle/ref_driver_mapping.cpp,
    ceres::Solve = the declared LM loop of oracle/ref_shim/ceres/solve_shim.h), and stores what they leave behind in tests/golden/ref_v5.npz,
  * tests/test_oracle_ref_mapping.py — the same calls live (build container) against the stored results and against the ORACLE compositions below
    (oracle/icp.h, oracle/loop.h, oracle/lm.h pieces put together by hand the way the reference's loops do),
  * tests/test_gpu_mapping.py — the same compositions over the C-ABI (lvf_scan_match, lvf_cloud_*, lvf_problem_solve with priors,
    lvf_relocate_rotation_solve, lvf_forward_update) against ref_v5.npz on the MI355X.
The cases are functions of this file and lvio_fusion_amd/synthetic.py alone."""
import numpy as np

from lvio_fusion_amd import synthetic as syn

RES = 0.2
N_FEATURES_LEFT = 40
EPSILON = 1e-3          # src/estimator.cpp: the reference's global `epsilon`


def _pad(a):
    out = np.zeros((a.shape[0], 4), np.float32); out[:, :3] = a.astype(np.float32); return out


def lidar_drive(n, seed, n_az=240, max_ground=900, max_surf=1300, noise=0.02):
    """n keyframes along a street-canyon drive with their BODY-frame feature clouds (ground / surf, float32 [m][4])"""
    rng = np.random.default_rng(seed)
    boxes = []
    for _ in range(40):
        c = np.array([rng.uniform(-10, 60), rng.uniform(-7, 7), -1.73])
        sz = rng.uniform(0.5, 3.0, 3)
        boxes.append((c - np.array([sz[0] / 2, sz[1] / 2, 0]), c + np.array([sz[0] / 2, sz[1] / 2, sz[2]])))
    poses = syn.drive_poses(n, rng)
    ground, surf = [], []
    for k in range(n):
        pl, g = syn._raycast_scene(poses[k, 4:], syn.rotmat(poses[k, :4]), rng, boxes, n_az=n_az)
        pl = pl + rng.normal(0, noise, pl.shape)
        gi, si = np.flatnonzero(g), np.flatnonzero(~g)
        gi = np.sort(rng.choice(gi, min(max_ground, gi.size), replace=False)); si = np.sort(rng.choice(si, min(max_surf, si.size), replace=False))
        ground.append(_pad(pl[gi])); surf.append(_pad(pl[si]))
    return dict(time=20.0 + 0.5 * np.arange(n), poses_true=poses, ground=ground, surf=surf, rng=rng)


def _perturb(pose, yaw_deg, dxyz):
    p = pose.copy()
    p[:4] = syn.quat_mul(pose[:4], syn.quat_from_ypr(np.deg2rad(yaw_deg), np.deg2rad(0.15), np.deg2rad(-0.1))); p[:4] /= np.linalg.norm(p[:4])
    p[4:] += np.asarray(dxyz)
    return p


def optimize_case():
    """Mapping::Optimize: 7 lidar keyframes, the first three are the map (already in the world), the last four are optimised one after the other;
    their estimates drift (the error grows along the drive), so ForwardUpdate after each frame matters for the next"""
    d = lidar_drive(7, 0x0A71)
    est = d["poses_true"].copy()
    for k in range(3, 7):
        est[k] = _perturb(d["poses_true"][k], 0.25 * (k - 2), [0.05 * (k - 2), -0.04 * (k - 2), 0.02 * (k - 2)])
    # features_left.size() sets the prior's weight (association.cpp:323,:381: size * weights.visual): with the reference's usual dozens of features the prior
    # holds the pose to micrometres; two keyframes with (almost) no features let the lidar terms move them by centimetres
    return dict(time=d["time"], pose=est, ground=d["ground"], surf=d["surf"], first_active=3, n_features_left=np.array([N_FEATURES_LEFT] * 3 + [0, 1, N_FEATURES_LEFT, 0], np.int32),
                w_ground=syn.W_LIDAR_GROUND, w_surf=syn.W_LIDAR_SURF, w_visual=syn.W_VISUAL)


RELOCATE_CASES = (("relocate", {}), ("relocate_sparse", dict(seed=0x0A75, keep=(140, 260))), ("relocate_poor", dict(seed=0x0A76, keep=(30, 45))),
                  ("relocate_large", dict(seed=0x0A78, drive=dict(n_az=1400, max_ground=5000, max_surf=8000))))      # 15 k + 24 k map points, 5 k + 8 k scan points


def relocate_case(seed=0x0A72, keep=None, drive=None):
    """Mapping::Relocate: three old keyframes (the loop's old frame in the middle), the current frame sees the same place from 0.9 m further on with a
    perturbed initial relative pose"""
    d = lidar_drive(4, seed, **(drive or {}))
    cur_true = d["poses_true"][3]
    g, s = d["ground"][3], d["surf"][3]
    if keep is not None:
        g, s = g[:keep[0]], s[:keep[1]]
    old = d["poses_true"][1]
    cur0 = _perturb(cur_true, 0.5, [0.10, -0.10, 0.05])
    return dict(time=d["time"][:3], pose=d["poses_true"][:3].copy(), ground=d["ground"][:3], surf=d["surf"][:3], old_index=1, cur_ground=g, cur_surf=s,
                cur_pose=cur0, rel_in=syn.se3_mul(syn.se3_inv(old), cur0), cur_true=cur_true,
                w_ground=syn.W_LIDAR_GROUND, w_surf=syn.W_LIDAR_SURF, w_visual=syn.W_VISUAL)


def pose_graph_case(n=13, seed=0x0A73):
    """PoseGraph::BuildProblem / Optimize: 13 keyframes; the loop's old frame is keyframe 0, the new sub-map starts at keyframe 12 (relocated by the
    loop closure); turning sections begin at keyframes 2, 5 and 9; the keyframes between the sections follow through ForwardUpdate"""
    rng = np.random.default_rng(seed)
    P = syn.drive_poses(n, rng, step=6.0)
    corr = np.concatenate([syn.quat_from_ypr(np.deg2rad(3.0), np.deg2rad(0.4), np.deg2rad(-0.3)), [1.5, -0.8, 0.2]])
    after = syn.se3_mul(corr, P[-1]); after[:4] /= np.linalg.norm(after[:4])
    time = 50.0 + 1.0 * np.arange(n)
    vw = rng.normal(0, 1.0, (n, 3))
    # `pose`: the drifted estimates BuildProblem sees (its edge targets are the CURRENT relative poses, pose_graph.cpp:186-197); `start_after`: where
    # Relocator::UpdateNewSubmap then puts the sub-map's start frame — a constant block of the problem — before PoseGraph::Optimize runs (relocator.cpp:214-216)
    return dict(time=time, pose=P, start_after=after, vw=vw, section_A=time[[2, 5, 9]], section_idx=[2, 5, 9], submap_A=time[0], submap_B=time[n - 1])


def submap_case(n=6, seed=0x0A74):
    """Relocator::UpdateNewSubmap: six keyframes of the new sub-map, each with its loop closure (old frame pose, relative_o_c); best frame = 3"""
    rng = np.random.default_rng(seed)
    P = syn.drive_poses(n, rng, step=1.5)
    drift = np.concatenate([syn.quat_from_ypr(np.deg2rad(2.0), np.deg2rad(0.3), np.deg2rad(-0.2)), [0.8, -0.5, 0.1]])
    old = np.stack([syn.se3_mul(drift, p) for p in P])            # where the old pass saw the same places
    for o in old:
        o[:4] /= np.linalg.norm(o[:4])
    rel = []
    for k in range(n):
        e = np.concatenate([syn.quat_from_ypr(*np.deg2rad(rng.normal(0, 0.3, 3))), rng.normal(0, 0.05, 3)])
        r = syn.se3_mul(e, np.array([0, 0, 0, 1.0, 0.4, 0.1, 0.0])); r[:4] /= np.linalg.norm(r[:4])
        rel.append(r)
    return dict(time=70.0 + 0.5 * np.arange(n), pose=P, old_pose=old, relative_o_c=np.stack(rel), best=3)


def environment_case(seed=0x0A77, n_lm=60):
    """Environment::Optimize (environment.cpp:18-115; the RL episodes' per-step solve, one independent window per environment): the current keyframe's pose
    is the only free block — PoseOnly blocks for its features (landmarks born two keyframes earlier), one ImuError to its predecessor with every other block
    constant.  Frames [birth, last, cur]; observations rounded to float (cv::Point2f)."""
    cfg = syn.config4_window(n_kf=3, n_lm=n_lm, n_prewindow=0, seed=seed, imu_samples=6)
    cams = {}
    for cam in ("cam0", "cam1"):
        c = dict(cfg[cam]); e = np.array(c["extrinsic"], np.float64); e[:4] /= np.linalg.norm(e[:4]); c["extrinsic"] = e; cams[cam] = c
    f32 = lambda x: np.asarray(x, np.float32).astype(np.float64)
    P = cfg["poses_true"].copy()
    cur = _perturb(P[2], 0.6, [0.08, -0.05, 0.03])
    rng = np.random.default_rng(seed + 1)
    # landmarks born in frame 0 (right-image observation + inverse depth there), seen by the left camera of frame 2
    d = rng.uniform(6.0, 40.0, n_lm)
    u = rng.uniform(100, 1100, n_lm); v = rng.uniform(60, 320, n_lm)
    c1 = cams["cam1"]
    right_ob = f32(np.stack([u, v], -1))
    ps = np.stack([(right_ob[:, 0] - c1["cx"]) / c1["fx"] * d, (right_ob[:, 1] - c1["cy"]) / c1["fy"] * d, d], -1)
    pw = syn.se3_apply(P[0], syn.se3_apply(c1["extrinsic"], ps))
    px, z = syn.project(cams["cam0"], np.tile(P[2], (n_lm, 1)), pw)
    keep = (z > 2.0) & (px[:, 0] > 0) & (px[:, 0] < 1241) & (px[:, 1] > 0) & (px[:, 1] < 376)
    left_ob = f32(px[keep] + rng.normal(0, 0.5, px[keep].shape))
    f = cfg["imu"][1]
    pose3 = np.stack([P[0], P[1], cur])
    return dict(cam0=cams["cam0"], cam1=cams["cam1"], baseline=syn.baseline(), pose3=pose3, vel3=cfg["vel"], ba3=np.stack([cfg["ba"][0], f["ba"], cfg["ba"][2]]),
                bg3=np.stack([cfg["bg"][0], f["bg"], cfg["bg"][2]]), w_visual=float(np.float32(syn.W_VISUAL)), samples=f["samples"], acc0=f["acc0"], gyr0=f["gyr0"],
                noise4=syn.IMU_NOISE, inv_depth=1.0 / d[keep], right_ob=right_ob[keep], left_ob=left_ob, pw=pw[keep], pose_true=P[2])


def environment_optimize(B, c):
    """Environment::Optimize by hand: a two-keyframe window [last, cur] with last's pose and all velocity / bias blocks constant (environment.cpp:62-68)"""
    return B.environment_solve(c)


# ------------------------------------------------------------------------------------------------ compositions
# `B` is a backend: an object with  transform(cloud, pose) -> world cloud (Mapping::MergeScan),  scan_match(map_ground, map_surf, scan_ground, scan_surf, map_pose,
# frame_pose, outer, prior_weight) -> (pose, score_ground, score_surf),  se3_mul / se3_inv,  forward_update(T, poses, vw) -> (poses, vw),
# pose_graph_solve(P, const, priors) -> P,  rotation_solve(relocated, unrelocated) -> q4.
def mapping_optimize(B, c):
    """Mapping::Optimize (mapping.cpp:139-191) by hand: per active keyframe — map = the last three lidar keyframes' WORLD clouds merged (BuildMapFrame :114-137,
    map pose = the latest of them), ground then surf sub-problem with the prior PoseErrorRPZ/YXY(features_left.size() * weights.visual), ForwardUpdate of every
    LATER keyframe by new * old^-1 (:181-183, time + epsilon), then the frame's own clouds go to the world (ToWorld :185)."""
    n = len(c["time"])
    pose = np.array(c["pose"], np.float64).copy()
    world_g = {k: B.transform(c["ground"][k], pose[k]) for k in range(c["first_active"])}
    world_s = {k: B.transform(c["surf"][k], pose[k]) for k in range(c["first_active"])}
    for k in range(c["first_active"], n):
        last = [j for j in range(k) if j in world_g][-3:]
        if last:
            mg = np.concatenate([world_g[j] for j in last]); ms = np.concatenate([world_s[j] for j in last])
            old = pose[k].copy()
            new, _, _ = B.scan_match(mg, ms, c["ground"][k], c["surf"][k], pose[last[-1]], pose[k], 1, c["n_features_left"][k] * c["w_visual"])
            pose[k] = new
            T = B.se3_mul(new, B.se3_inv(old))
            if k + 1 < n:
                pose[k + 1:], _ = B.forward_update(T, pose[k + 1:], None)
        world_g[k] = B.transform(c["ground"][k], pose[k]); world_s[k] = B.transform(c["surf"][k], pose[k])
    return pose


def mapping_relocate(B, c):
    """Mapping::Relocate (mapping.cpp:251-300) by hand: map = previous + old + subsequent lidar keyframes' world clouds in time order (BuildOldMapFrame :78-112,
    map pose = the earliest of them), clone pose = old pose * relative_o_c, four outer passes without prior; score = int(score_ground + score_surf)."""
    i = c["old_index"]
    idx = [j for j in (i - 1, i, i + 1) if 0 <= j < len(c["time"])]
    mg = np.concatenate([B.transform(c["ground"][j], c["pose"][j]) for j in idx]); ms = np.concatenate([B.transform(c["surf"][j], c["pose"][j]) for j in idx])
    start = B.se3_mul(c["pose"][i], c["rel_in"])
    pose, sg, ss = B.scan_match(mg, ms, c["cur_ground"], c["cur_surf"], c["pose"][idx[0]], start, 4, 0.0)
    return int(sg + ss), B.se3_mul(B.se3_inv(c["pose"][i]), pose), c["pose"][idx[0]], (len(mg), len(ms))


def pose_graph_priors(B, c):
    """the blocks PoseGraph::BuildProblem adds (pose_graph.cpp:163-199) in the prior-batch form of oracle/lm.h / lvf_pose_prior_create: keyframes renumbered
    old = 0, sections 1.., start = last"""
    P = np.array(c["pose"], np.float64)
    chain = [0] + list(c["section_idx"]) + [len(P) - 1]
    Pc = P[chain]
    m = len(chain)
    ka, kb, tgt, w, v = [], [], [], [], []
    for k in range(1, m - 1):
        ka.append(k - 1); kb.append(k); tgt.append(np.concatenate([B.pose_graph_target(Pc[k - 1], Pc[k]), [0.0]])); w.append(1.0); v.append(1.0)      # PoseGraphError(last, A)
        ka.append(-2); kb.append(k); tgt.append(Pc[k].copy()); w.append(1.0); v.append(0.0)                                                             # RError(A)
    ka.append(m - 2); kb.append(m - 1); tgt.append(np.concatenate([B.pose_graph_target(Pc[m - 2], Pc[m - 1]), [0.0]])); w.append(1.0); v.append(1.0)
    return chain, Pc, dict(kf_a=np.array(ka, np.int32), kf_b=np.array(kb, np.int32), target=np.array(tgt), weight=np.array(w), v=np.array(v))


def pose_graph_optimize(B, c):
    """PoseGraph::Optimize (pose_graph.cpp:201-224) by hand: the solve over (old, sections, start) with both ends constant, then per section the keyframes up to
    the next section (the last one: up to the start frame) follow by new_A * old_A^-1; Vw is rotated along."""
    chain, Pc, pr = pose_graph_priors(B, c)
    P = np.array(c["pose"], np.float64).copy(); vw = np.array(c["vw"], np.float64).copy()
    const = np.zeros(len(chain), np.uint8); const[0] = const[-1] = 1
    Pc = Pc.copy(); Pc[-1] = c["start_after"]; P[-1] = c["start_after"]
    Pn = B.pose_graph_solve(Pc, const, pr)
    for s in range(1, len(chain) - 1):
        a, b = chain[s], chain[s + 1]
        T = B.se3_mul(Pn[s], B.se3_inv(c["pose"][a]))
        P[a] = Pn[s]
        if b - a > 1:
            P[a + 1:b], vw[a + 1:b] = B.forward_update(T, P[a + 1:b], vw[a + 1:b])
    return P, vw


def update_new_submap(B, c):
    """Relocator::UpdateNewSubmap (relocator.cpp:247-282) by hand"""
    P = np.array(c["pose"], np.float64).copy()
    b = c["best"]
    base = P[b].copy()
    target_b = B.se3_mul(c["old_pose"][b], c["relative_o_c"][b])
    relocated = np.stack([B.se3_mul(B.se3_inv(target_b), B.se3_mul(c["old_pose"][k], c["relative_o_c"][k])) for k in range(len(P))])
    # (the loop runs over the best frame too, whose pose has ALREADY been replaced by its loop-closure target: relocator.cpp:253,:261)
    unrelocated = np.stack([B.se3_mul(B.se3_inv(base), target_b if k == b else P[k]) for k in range(len(P))])
    q = B.rotation_solve(relocated, unrelocated)
    new_b = B.se3_mul(target_b, np.concatenate([q, [0.0, 0.0, 0.0]]))
    T = B.se3_mul(new_b, B.se3_inv(base))
    others = [k for k in range(len(P)) if k != b]
    P[others], _ = B.forward_update(T, P[others], None)
    P[b] = new_b
    return P


class OracleBackend:
    """the compositions over the CPU oracle (oracle/pyoracle.py)"""
    def __init__(self, oracle):
        self.o = oracle
        self.se3_mul, self.se3_inv, self.pose_graph_target = oracle.se3_mul, oracle.se3_inv, oracle.pose_graph_target

    def transform(self, cloud, pose):
        return self.o.cloud_transform(cloud, pose)

    def scan_match(self, mg, ms, qg, qs, map_pose, frame_pose, outer, prior_w):
        o = self.o
        pose = np.array(frame_pose, np.float64).copy()
        sg = ss = 0.0
        for _ in range(outer):
            x = o.se3_to_rpyxyz(o.se3_mul(o.se3_inv(map_pose), pose))
            if len(mg):
                x, g = o.icp_solve(mg, qg, map_pose, pose, x, 0, RES * RES * 100, syn.W_LIDAR_GROUND, 0.0, prior_w=prior_w)
                pose = o.se3_mul(map_pose, o.rpyxyz_to_se3(x))
                sg = min(g["num_residual_blocks"] / 10, 20.0) - 2 * g["final_cost"] / g["num_residual_blocks"]
            if len(ms):
                x, s = o.icp_solve(ms, qs, map_pose, pose, x, 1, RES * RES * 25, syn.W_LIDAR_SURF, 0.1, prior_w=prior_w)
                pose = o.se3_mul(map_pose, o.rpyxyz_to_se3(x))
                ss = min(s["num_residual_blocks"] / 10, 30.0) - 2 * s["final_cost"] / s["num_residual_blocks"]
        return pose, sg, ss

    def forward_update(self, T, poses, vw):
        return self.o.forward_update(T, poses, vw)

    def pose_graph_solve(self, Pc, const, pr):
        n = len(Pc)
        cams = syn.kitti_cameras()
        cfg = dict(n_kf=n, n_lm=0, poses=Pc, vel=np.zeros((n, 3)), ba=np.zeros((n, 3)), bg=np.zeros((n, 3)), inv_depth=np.zeros(0), w_kf=np.ones(n), cam0=cams[0], cam1=cams[1],
                   tc=dict(left_ob=np.zeros((0, 2)), right_ob=np.zeros((0, 2)), lm_idx=np.zeros(0, np.int32), kf_idx=np.zeros(0, np.int32)),
                   tf=dict(first_ob=np.zeros((0, 2)), ob=np.zeros((0, 2)), lm_idx=np.zeros(0, np.int32), kf1_idx=np.zeros(0, np.int32), kf2_idx=np.zeros(0, np.int32)),
                   po=dict(ob=np.zeros((0, 2)), kf_idx=np.zeros(0, np.int32), pw_idx=np.zeros(0, np.int32), pw=np.zeros((1, 3))), imu=[])
        win = self.o.Window(cfg, np.zeros((0, self.o.PREINT_DOUBLES)), pose_const=const, use=(), priors=pr)
        win.solve()           # ceres::Solver::Options defaults (pose_graph.cpp:203-206)
        return win.poses.copy()

    def rotation_solve(self, relocated, unrelocated):
        q, _ = self.o.relocate_rotation_solve(relocated, unrelocated, [0, 0, 0, 1.0])
        return np.asarray(q, np.float64)

    def environment_solve(self, c):
        o = self.o
        n = len(c["inv_depth"])
        pre = o.imu_preintegrate(c["samples"], c["acc0"], c["gyr0"], c["ba3"][1], c["bg3"][1], c["noise4"])
        cfg = dict(n_kf=2, n_lm=0, poses=c["pose3"][1:], vel=c["vel3"][1:], ba=c["ba3"][1:], bg=c["bg3"][1:], inv_depth=np.zeros(0), w_kf=np.full(2, c["w_visual"]),
                   cam0=c["cam0"], cam1=c["cam1"],
                   tc=dict(left_ob=np.zeros((0, 2)), right_ob=np.zeros((0, 2)), lm_idx=np.zeros(0, np.int32), kf_idx=np.zeros(0, np.int32)),
                   tf=dict(first_ob=np.zeros((0, 2)), ob=np.zeros((0, 2)), lm_idx=np.zeros(0, np.int32), kf1_idx=np.zeros(0, np.int32), kf2_idx=np.zeros(0, np.int32)),
                   po=dict(ob=c["left_ob"], kf_idx=np.ones(n, np.int32), pw_idx=np.arange(n, dtype=np.int32), pw=world_points(o, c)), imu=[dict(kf_i=0, kf_j=1)])
        win = o.Window(cfg, np.stack([pre]), pose_const=np.array([1, 0], np.uint8), vbb_const=np.array([7, 7], np.uint8), use=("po", "imu"))
        win.solve()
        return win.poses[1].copy()


def world_points(o, c):
    """Landmark::ToWorld (src/landmark.cpp:15-19) of every landmark: Pixel2Robot of the right-image observation at depth 1 / inv_depth, then the birth frame's pose"""
    c1 = c["cam1"]
    d = 1.0 / np.asarray(c["inv_depth"], np.float64)
    ps = np.stack([(c["right_ob"][:, 0] - c1["cx"]) * d / c1["fx"], (c["right_ob"][:, 1] - c1["cy"]) * d / c1["fy"], d], -1)
    return np.stack([o.se3_apply(c["pose3"][0], o.se3_apply(c1["extrinsic"], p)) for p in ps])
