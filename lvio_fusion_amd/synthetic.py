"""Deterministic synthetic KITTI-shaped inputs for the hot path.

Pure numpy; no oracle and no GPU code is touched here.  Constants come from the reference's
KITTI configuration (src/lvio_fusion_node/config/kitti.yaml:23-32 intrinsics, :35-45 lidar,
:48-52 IMU noise, :55-82 extrinsics) and default factor weights (src/lvio_fusion/src/frame.cpp:13-15).
Pose layout everywhere: Sophus SE3d::data() = [qx,qy,qz,qw,tx,ty,tz]; Twc maps body->world.
"""
import numpy as np

FX = FY = 718.856
CX, CY = 607.1928, 185.2157
BODY_TO_CAM0 = np.array([[0.00875117, -0.00479608, 0.99995, 1.10224],
                         [-0.999865, -0.0140025, 0.00868325, -0.319072],
                         [0.0139602, -0.999891, -0.00491796, 0.746066]])
BODY_TO_CAM1 = np.array([[0.00875117, -0.00479608, 0.99995, 1.10695],
                         [-0.999865, -0.0140025, 0.00868325, -0.856165],
                         [0.0139602, -0.999891, -0.00491796, 0.753565]])
W_VISUAL = FX / 10.0          # frame.cpp:13
W_LIDAR_GROUND = 1.0          # frame.cpp:14
W_LIDAR_SURF = 0.01           # frame.cpp:15
IMU_NOISE = (0.1, 0.01, 1e-3, 1e-4)   # acc_n, gyr_n, acc_w, gyr_w  kitti.yaml:48-51
GRAVITY = np.array([0.0, 0.0, 9.81007])
LIDAR_RES = 0.2               # kitti.yaml:45
THR_GROUND = LIDAR_RES * LIDAR_RES * 100   # association.cpp:285
THR_SURF = LIDAR_RES * LIDAR_RES * 25      # association.cpp:343

SEED_CFG2 = 0x10F051
SEED_CFG3 = 0x1CB
SEED_CFG4 = 0xBA50


# ----------------------------------------------------------------------------- quaternion / SE3 (x,y,z,w)
def quat_from_rotmat(R):
    """Eigen's Quaternion(Matrix3) branch structure; returns [x,y,z,w]."""
    t = R[0, 0] + R[1, 1] + R[2, 2]
    if t > 0:
        s = np.sqrt(t + 1.0)
        w = 0.5 * s
        s = 0.5 / s
        return np.array([(R[2, 1] - R[1, 2]) * s, (R[0, 2] - R[2, 0]) * s, (R[1, 0] - R[0, 1]) * s, w])
    i = int(np.argmax([R[0, 0], R[1, 1], R[2, 2]]))
    j, k = (i + 1) % 3, (i + 2) % 3
    s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
    q = np.zeros(4)
    q[i] = 0.5 * s
    s = 0.5 / s
    q[3] = (R[k, j] - R[j, k]) * s
    q[j] = (R[j, i] + R[i, j]) * s
    q[k] = (R[k, i] + R[i, k]) * s
    return q


def rotmat(q):
    q = np.asarray(q, dtype=np.float64)
    q = q / np.linalg.norm(q, axis=-1, keepdims=True)
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    R = np.empty(q.shape[:-1] + (3, 3))
    R[..., 0, 0] = 1 - 2 * (y * y + z * z); R[..., 0, 1] = 2 * (x * y - w * z); R[..., 0, 2] = 2 * (x * z + w * y)
    R[..., 1, 0] = 2 * (x * y + w * z); R[..., 1, 1] = 1 - 2 * (x * x + z * z); R[..., 1, 2] = 2 * (y * z - w * x)
    R[..., 2, 0] = 2 * (x * z - w * y); R[..., 2, 1] = 2 * (y * z + w * x); R[..., 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def quat_mul(a, b):
    ax, ay, az, aw = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bx, by, bz, bw = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx,
                     aw * bw - ax * bx - ay * by - az * bz], axis=-1)


def quat_from_ypr(yaw, pitch, roll):
    """Z-Y-X; matches base.hpp:110-132 (rpy[0]=yaw)."""
    z, y, x = np.asarray(yaw) / 2, np.asarray(pitch) / 2, np.asarray(roll) / 2
    cz, sz, cy, sy, cx, sx = np.cos(z), np.sin(z), np.cos(y), np.sin(y), np.cos(x), np.sin(x)
    w = cz * cy * cx + sz * sy * sx
    qx = cz * cy * sx - sz * sy * cx
    qy = cz * sy * cx + sz * cy * sx
    qz = sz * cy * cx - cz * sy * sx
    return np.stack([qx, qy, qz, w], axis=-1)


def se3_apply(T, p):
    return np.einsum("...ij,...j->...i", rotmat(T[..., :4]), p) + T[..., 4:]


def se3_inv(T):
    R = rotmat(T[..., :4])
    q = T[..., :4] * np.array([-1.0, -1.0, -1.0, 1.0])
    t = -np.einsum("...ji,...j->...i", R, T[..., 4:])
    return np.concatenate([q, t], axis=-1)


def se3_mul(A, B):
    q = quat_mul(A[..., :4], B[..., :4])
    t = se3_apply(A, B[..., 4:])
    return np.concatenate([q, t], axis=-1)


def camera_extrinsic(M34):
    return np.concatenate([quat_from_rotmat(M34[:, :3]), M34[:, 3]])


def kitti_cameras():
    """(cam0, cam1) as dicts {fx,fy,cx,cy,extrinsic[7]} ; estimator.cpp:36-83."""
    c0 = dict(fx=FX, fy=FY, cx=CX, cy=CY, extrinsic=camera_extrinsic(BODY_TO_CAM0))
    c1 = dict(fx=FX, fy=FY, cx=CX, cy=CY, extrinsic=camera_extrinsic(BODY_TO_CAM1))
    return c0, c1


def baseline():
    return float(np.linalg.norm(BODY_TO_CAM0[:, 3] - BODY_TO_CAM1[:, 3]))   # estimator.cpp:84


def project(cam, T_wc_body, pw):
    """Reference Reprojection in matrix form (visual_error.hpp:10-23); returns pixels, depth."""
    pb = se3_apply(se3_inv(T_wc_body), pw)
    pc = se3_apply(se3_inv(cam["extrinsic"]), pb)
    return np.stack([cam["fx"] * pc[..., 0] / pc[..., 2] + cam["cx"], cam["fy"] * pc[..., 1] / pc[..., 2] + cam["cy"]], -1), pc[..., 2]


def drive_poses(n_kf, rng, step=1.2):
    """KITTI-like forward drive: x forward 1.2 m/KF, small lateral/vertical jitter, yaw random walk."""
    yaw = np.cumsum(rng.normal(0, np.deg2rad(2.0), n_kf))
    pitch = rng.normal(0, np.deg2rad(0.5), n_kf)
    roll = rng.normal(0, np.deg2rad(0.5), n_kf)
    q = quat_from_ypr(yaw, pitch, roll)
    t = np.stack([step * np.arange(n_kf), rng.normal(0, 0.05, n_kf), rng.normal(0, 0.02, n_kf)], -1)
    return np.concatenate([q, t], -1)


def perturb_poses(poses, rng, sig_rot_deg=0.5, sig_t=0.05):
    n = poses.shape[0]
    dq = quat_from_ypr(*(rng.normal(0, np.deg2rad(sig_rot_deg), (3, n))))
    out = poses.copy()
    out[:, :4] = quat_mul(poses[:, :4], dq)
    out[:, :4] /= np.linalg.norm(out[:, :4], axis=1, keepdims=True)
    out[:, 4:] += rng.normal(0, sig_t, (n, 3))
    return out


# ----------------------------------------------------------------------------- config 2
def config2_pose_only(n_lm=10000, n_kf=50, seed=SEED_CFG2, pix_sigma=0.5):
    """Dense 10k x 50 PoseOnlyReprojection batch, sorted by keyframe."""
    rng = np.random.default_rng(seed)
    cam0, _ = kitti_cameras()
    poses = drive_poses(n_kf, rng)
    # landmarks in the frustum of the LAST keyframe's left camera, depth 4..80 m => in front of every KF
    d = rng.uniform(4.0, 80.0, n_lm)
    u = rng.uniform(0, 1241, n_lm); v = rng.uniform(0, 376, n_lm)
    pc = np.stack([(u - CX) / FX * d, (v - CY) / FY * d, d], -1)
    pw = se3_apply(poses[-1], se3_apply(cam0["extrinsic"], pc))
    kf_idx = np.repeat(np.arange(n_kf, dtype=np.int32), n_lm)
    pw_idx = np.tile(np.arange(n_lm, dtype=np.int32), n_kf)
    px, _ = project(cam0, poses[kf_idx], pw[pw_idx])
    ob = px + rng.normal(0, pix_sigma, px.shape)
    est = perturb_poses(poses, rng)   # the state the factors are evaluated at
    return dict(cam0=cam0, poses_true=poses, poses=est, pw=pw, ob=ob, kf_idx=kf_idx, pw_idx=pw_idx,
                w_kf=np.full(n_kf, W_VISUAL), n_kf=n_kf, n_lm=n_lm)


# ----------------------------------------------------------------------------- IMU
def synth_imu_samples(pose_i, pose_j, v_i, v_j, ba, bg, n_samples, dt_total, rng, noise=True):
    """n_samples (dt, acc, gyr) rows consistent with constant body-rate / constant world accel
    between two keyframes, plus white noise.  Returns samples[n][7], acc0[3], gyr0[3]."""
    dt = dt_total / n_samples
    Ri, Rj = rotmat(pose_i[:4]), rotmat(pose_j[:4])
    dR = Ri.T @ Rj
    ang = np.arccos(np.clip((np.trace(dR) - 1) / 2, -1, 1))
    axis = np.array([dR[2, 1] - dR[1, 2], dR[0, 2] - dR[2, 0], dR[1, 0] - dR[0, 1]])
    w_body = axis / (2 * np.sin(ang)) * ang / dt_total if ang > 1e-9 else np.zeros(3)
    a_world = (v_j - v_i) / dt_total
    rows = []
    th = np.linalg.norm(w_body)
    for s in range(n_samples + 1):
        tt = s * dt
        if th > 1e-12:
            k = w_body / th
            K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
            Rt = Ri @ (np.eye(3) + np.sin(th * tt) * K + (1 - np.cos(th * tt)) * K @ K)
        else:
            Rt = Ri
        acc = Rt.T @ (a_world + GRAVITY) + ba
        gyr = w_body + bg
        if noise:
            acc = acc + rng.normal(0, 0.02, 3)
            gyr = gyr + rng.normal(0, 0.002, 3)
        rows.append(np.concatenate([[dt], acc, gyr]))
    rows = np.array(rows)
    return rows[1:], rows[0, 1:4], rows[0, 4:7]


# ----------------------------------------------------------------------------- config 4
def config4_window(n_kf=50, n_lm=10000, n_prewindow=2000, seed=SEED_CFG4, imu_samples=10, kf_dt=0.1 * 10,
                   pix_sigma=0.5, p_geom=0.1, ids_by_birth=False):
    """Full sliding-window problem: TwoCamera + TwoFrame + PoseOnly + ImuError block lists
    (block mix of backend.cpp:96-183)."""
    rng = np.random.default_rng(seed)
    cam0, cam1 = kitti_cameras()
    poses = drive_poses(n_kf, rng)
    # landmarks: born at KF f, defined by right-image pixel + depth in cam1 (Pixel2Robot, visual_error.hpp:25-33)
    birth = rng.integers(0, n_kf, n_lm).astype(np.int32)
    if ids_by_birth:       # landmark ids handed out in creation order (what a live front-end does): neighbouring blocks of a
        birth.sort()       # keyframe's feature list then share their first keyframe
    depth = rng.uniform(4.0, 80.0, n_lm)
    u1 = rng.uniform(0, 1241, n_lm); v1 = rng.uniform(0, 376, n_lm)
    ps = np.stack([(u1 - CX) / FX * depth, (v1 - CY) / FY * depth, depth], -1)
    pb = se3_apply(cam1["extrinsic"], ps)
    pw = se3_apply(poses[birth], pb)
    first_ob_right = np.stack([u1, v1], -1)                       # exact by construction
    left_px, _ = project(cam0, poses[birth], pw)
    left_ob_birth = left_px + rng.normal(0, pix_sigma, left_px.shape)
    inv_depth_true = 1.0 / depth
    # tracks
    length = np.minimum(1 + rng.geometric(p_geom, n_lm), n_kf - birth)
    tf_lm, tf_k1, tf_k2 = [], [], []
    for l in range(n_lm):
        ks = np.arange(birth[l] + 1, birth[l] + length[l])
        if ks.size:
            tf_lm.append(np.full(ks.size, l)); tf_k1.append(np.full(ks.size, birth[l])); tf_k2.append(ks)
    cat = lambda xs: np.concatenate(xs).astype(np.int32) if xs else np.zeros(0, np.int32)
    tf_lm, tf_k1, tf_k2 = cat(tf_lm), cat(tf_k1), cat(tf_k2)
    px, z = project(cam0, poses[tf_k2], pw[tf_lm])
    keep = z > 2.0
    tf_lm, tf_k1, tf_k2, px = tf_lm[keep], tf_k1[keep], tf_k2[keep], px[keep]
    tf_ob = px + rng.normal(0, pix_sigma, px.shape)
    order = np.lexsort((tf_lm, tf_k2))        # sort by current KF, then landmark
    tf_lm, tf_k1, tf_k2, tf_ob = tf_lm[order], tf_k1[order], tf_k2[order], tf_ob[order]
    # pre-window landmarks seen by the first 10 KFs (PoseOnly, constant world point)
    dpw = rng.uniform(6.0, 80.0, n_prewindow)
    upw = rng.uniform(0, 1241, n_prewindow); vpw = rng.uniform(0, 376, n_prewindow)
    pcw = np.stack([(upw - CX) / FX * dpw, (vpw - CY) / FY * dpw, dpw], -1)
    pw_pre = se3_apply(poses[0], se3_apply(cam0["extrinsic"], pcw))
    nfirst = min(10, n_kf)
    vis = rng.random((nfirst, n_prewindow)) < 0.5
    po_kf, po_pw = np.nonzero(vis)
    pxp, zp = project(cam0, poses[po_kf], pw_pre[po_pw])
    keep = zp > 2.0
    po_kf, po_pw, pxp = po_kf[keep].astype(np.int32), po_pw[keep].astype(np.int32), pxp[keep]
    po_ob = pxp + rng.normal(0, pix_sigma, pxp.shape)
    # IMU
    t_kf = np.arange(n_kf) * kf_dt
    vel = np.gradient(poses[:, 4:], t_kf, axis=0)
    ba = rng.normal(0, 0.02, (n_kf, 3)); bg = rng.normal(0, 0.002, (n_kf, 3))
    imu = []
    for i in range(n_kf - 1):
        s, a0, g0 = synth_imu_samples(poses[i], poses[i + 1], vel[i], vel[i + 1], ba[i], bg[i], imu_samples, kf_dt, rng)
        imu.append(dict(samples=s, acc0=a0, gyr0=g0, ba=ba[i].copy(), bg=bg[i].copy(), kf_i=i, kf_j=i + 1))
    est = perturb_poses(poses, rng)
    inv_depth = inv_depth_true * (1 + rng.normal(0, 0.05, n_lm))
    lm_all = np.arange(n_lm, dtype=np.int32)
    return dict(cam0=cam0, cam1=cam1, n_kf=n_kf, n_lm=n_lm, poses_true=poses, poses=est,
                vel=vel + rng.normal(0, 0.05, vel.shape), ba=ba + rng.normal(0, 0.005, ba.shape),
                bg=bg + rng.normal(0, 0.0005, bg.shape), inv_depth=inv_depth, inv_depth_true=inv_depth_true,
                w_kf=np.full(n_kf, W_VISUAL),
                tc=dict(left_ob=left_ob_birth, right_ob=first_ob_right, lm_idx=lm_all, kf_idx=birth),
                tf=dict(first_ob=first_ob_right[tf_lm], ob=tf_ob, lm_idx=tf_lm, kf1_idx=tf_k1, kf2_idx=tf_k2),
                po=dict(ob=po_ob, kf_idx=po_kf, pw_idx=po_pw, pw=pw_pre),
                imu=imu)


# ----------------------------------------------------------------------------- config 3 (lidar)
def _raycast_scene(origin, R_wl, rng, boxes, n_rings=64, n_az=2900, ang_bottom=24.9, ang_res_y=0.427,
                   min_range=5.0, max_range=30.0, ground_z=-1.73, wall_y=8.0):
    """64-beam sweep against ground plane + two walls + axis-aligned boxes. Returns points in the
    LIDAR/body frame and a ground flag."""
    el = np.deg2rad(-ang_bottom + ang_res_y * np.arange(n_rings))
    az = np.linspace(-np.pi, np.pi, n_az, endpoint=False)
    E, A = np.meshgrid(el, az, indexing="ij")
    d_l = np.stack([np.cos(E) * np.cos(A), np.cos(E) * np.sin(A), np.sin(E)], -1).reshape(-1, 3)
    d_w = d_l @ R_wl.T
    o = origin
    best = np.full(d_w.shape[0], np.inf); is_ground = np.zeros(d_w.shape[0], bool)
    with np.errstate(divide="ignore", invalid="ignore"):
        t = (ground_z - o[2]) / d_w[:, 2]
        ok = (t > 0) & (t < best); best[ok] = t[ok]; is_ground[ok] = True
        for s in (+1.0, -1.0):
            t = (s * wall_y - o[1]) / d_w[:, 1]
            ok = (t > 0) & (t < best); best[ok] = t[ok]; is_ground[ok] = False
        for (lo, hi) in boxes:
            t1 = (lo - o) / d_w; t2 = (hi - o) / d_w
            tn = np.nanmax(np.minimum(t1, t2), axis=1); tf_ = np.nanmin(np.maximum(t1, t2), axis=1)
            ok = (tn > 0) & (tn <= tf_) & (tn < best); best[ok] = tn[ok]; is_ground[ok] = False
    keep = np.isfinite(best) & (best >= min_range) & (best <= max_range)
    pts_l = d_l[keep] * best[keep, None]
    return pts_l, is_ground[keep]


def config3_icp(seed=SEED_CFG3, n_query=100000, noise=0.02, n_az=2900):
    """Scan-to-map association inputs: query scan in the body frame, map = 3 previous scans merged in
    world frame.  Float32 xyz + pad (16 B/pt).  The azimuth sampling (2900 steps x 64 rings)
    is denser than kitti.yaml's horizon_scan so that, after the 5-30 m range gate, the query has the stated
    100 000 points and the 3-scan map ~300 000."""
    rng = np.random.default_rng(seed)
    boxes = []
    for _ in range(40):
        c = np.array([rng.uniform(-10, 60), rng.uniform(-7, 7), -1.73])
        sz = rng.uniform(0.5, 3.0, 3)
        boxes.append((c - np.array([sz[0] / 2, sz[1] / 2, 0]), c + np.array([sz[0] / 2, sz[1] / 2, sz[2]])))
    poses = drive_poses(4, rng)
    clouds, grounds = [], []
    for i in range(4):
        pl, g = _raycast_scene(poses[i, 4:], rotmat(poses[i, :4]), rng, boxes, n_az=n_az)
        pl = pl + rng.normal(0, noise, pl.shape)
        clouds.append(pl); grounds.append(g)
    map_w = np.concatenate([se3_apply(poses[i], clouds[i]) for i in range(3)])
    map_ground = np.concatenate(grounds[:3])
    q, qg = clouds[3], grounds[3]
    if q.shape[0] > n_query:
        sel = np.sort(rng.choice(q.shape[0], n_query, replace=False)); q, qg = q[sel], qg[sel]
    # initial pose error: yaw 0.5 deg, 0.10 m xy, pitch/roll 0.2 deg, z 0.05 m
    err_q = quat_from_ypr(np.deg2rad(0.5), np.deg2rad(0.2), np.deg2rad(-0.2))
    pose0 = poses[3].copy()
    pose0[:4] = quat_mul(poses[3, :4], err_q); pose0[:4] /= np.linalg.norm(pose0[:4])
    pose0[4:] += np.array([0.10, -0.10, 0.05])

    def pad(a):
        out = np.zeros((a.shape[0], 4), np.float32); out[:, :3] = a.astype(np.float32); return out
    return dict(query=pad(q), query_ground=qg, map=pad(map_w), map_ground=map_ground, pose_true=poses[3], pose0=pose0,
                map_pose=poses[2], thr_ground=THR_GROUND, thr_surf=THR_SURF)


# ----------------------------------------------------------------------------- config 5 (loop-closure candidates)
def config5_candidates(n=8, seed=0x5CA7, n_query=20000, n_az=600, overlap="varied"):
    """n independent relocalisation candidates: each is a config-3 style scene with its own seed;
    the 'old' keyframe is the map pose, the candidate's initial pose is the perturbed query pose.

    Mapping::Relocate's score (mapping.cpp:279-294) is min(N_ground / 10, 20) + min(N_surf / 10, 30) - 2 cost / N per sub-problem,
    truncated to int: with thousands of accepted correspondences every candidate saturates at 49.  overlap = "varied" (default) gives the
    candidates DIFFERENT overlap with their old sub-map — candidate i keeps only KEEP[i % 8] ground / surf query points (a loop closure
    seen from further away, or a mostly occluded one) — so the scores spread from "rejected" (score - 20 <= 0) to saturated and the
    arg-max has something to decide.  Three slots saturate on purpose: the reference's `>=` keeps the LATER of equal scores
    (relocator.cpp:200).  overlap = "full": every candidate keeps its whole scan (round 1-2 behaviour)."""
    KEEP = [(60, 110), (400, 900), (140, 260), (90, 200), (30, 45), (180, 150), (100000, 100000), (100000, 100000)]
    out = []
    for i in range(n):
        c = config3_icp(seed=seed + i, n_query=n_query, n_az=n_az)
        q, qg, init = c["query"], c["query_ground"], c["pose0"]
        if overlap == "varied":
            rng = np.random.default_rng(seed + 7919 * (i + 1))
            keep = KEEP[i % len(KEEP)]
            gi, si = np.flatnonzero(qg), np.flatnonzero(~qg)
            gi = np.sort(rng.choice(gi, min(keep[0], gi.size), replace=False)); si = np.sort(rng.choice(si, min(keep[1], si.size), replace=False))
            sel = np.sort(np.concatenate([gi, si]))
            q, qg = q[sel], qg[sel]
        out.append(dict(map=c["map"], map_ground=c["map_ground"], query=q, query_ground=qg,
                        map_pose=c["map_pose"], last_pose=c["map_pose"], init_pose=init, pose_true=c["pose_true"]))
    return out


# ----------------------------------------------------------------------------- raw sensor-frame scan (feature extraction)
def raw_scan(seed=0x5CA9, n_az=1800, noise=0.01, max_return=120.0):
    """One 64-beam revolution in the SENSOR frame in acquisition order (azimuth sweeps clockwise like a Velodyne, all rings per
    firing), against the street scene of config 3; rays with no return within max_return are NaN (pcl::removeNaNFromPointCloud has
    work to do), nothing is range-gated (Preprocess does that).  float32 [n][4] (x, y, z, 0)."""
    rng = np.random.default_rng(seed)
    boxes = []
    for _ in range(40):
        c = np.array([rng.uniform(-30, 30), rng.uniform(-7, 7), -1.73])
        sz = rng.uniform(0.5, 3.0, 3)
        boxes.append((c - np.array([sz[0] / 2, sz[1] / 2, 0]), c + np.array([sz[0] / 2, sz[1] / 2, sz[2]])))
    el = np.deg2rad(-24.9 + 0.427 * (np.arange(64) + rng.uniform(0.35, 0.65)))     # rings sit inside their rows, not on the boundaries
    # clockwise (-atan2(y, x) increases); firings sit near the CENTRES of the range-image columns (column = round(angle / 0.2 deg)),
    # away from the rounding boundaries where an ulp of atan2f would decide the pixel
    az = np.pi - 2 * np.pi * (np.arange(n_az) + 1.0 + rng.uniform(-0.25, 0.25)) / n_az
    A, E = np.meshgrid(az, el, indexing="ij")                                      # azimuth-major
    d = np.stack([np.cos(E) * np.cos(A), np.cos(E) * np.sin(A), np.sin(E)], -1).reshape(-1, 3)
    best = np.full(d.shape[0], np.inf)
    with np.errstate(divide="ignore", invalid="ignore"):
        t = -1.73 / d[:, 2]; ok = (t > 0) & (t < best); best[ok] = t[ok]
        for s in (+1.0, -1.0):
            t = (s * 8.0) / d[:, 1]; ok = (t > 0) & (t < best); best[ok] = t[ok]
        for (lo, hi) in boxes:
            t1 = lo / d; t2 = hi / d
            tn = np.nanmax(np.minimum(t1, t2), axis=1); tf_ = np.nanmin(np.maximum(t1, t2), axis=1)
            ok = (tn > 0) & (tn <= tf_) & (tn < best); best[ok] = tn[ok]
    best = best * (1 + rng.normal(0, noise / 10, best.shape))
    pts = d * best[:, None]
    pts[~(best < max_return)] = np.nan
    out = np.zeros((pts.shape[0], 4), np.float32)
    out[:, :3] = pts.astype(np.float32)
    return out


def lidar_extrinsic():
    """body_to_lidar of config/kitti.yaml:74-80 as an SE3 (sensor -> robot)."""
    return np.array([0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0]) if False else np.concatenate([quat_from_ypr(np.deg2rad(0.4), np.deg2rad(-0.3), np.deg2rad(0.2)), [0.27, 0.0, -0.08]])
