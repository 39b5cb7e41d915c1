"""Thin numpy-facing handles over the C-ABI (include/lvf.h).  Used by tests/ and bench.py; the C++ adapter
(include/lvf_ceres_adapter.hpp) is the reference-shaped host interface.  Everything here runs on the GPU —
errors from the library are raised, never papered over."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import LidarExtractDebug, LidarParams, WindowOptions, Camera, IcpOptions, IcpSummary, ScanMatchJob, ScanMatchOptions, ScanMatchResult, SolverOptions, SolverSummary

POSES, VEL, BA, BG, INV_DEPTH, W_VISUAL = range(6)
IMU_BLOCK_SIZES = (7, 3, 3, 3, 7, 3, 3, 3)


class LvfError(RuntimeError):
    pass


def _chk(rc):
    if rc != 0:
        raise LvfError(f"lvf error {rc}: {_lib.lib().lvf_last_error().decode()}")


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _dp(a):
    return a.ctypes.data_as(_lib.c_double_p)


def _ip(a):
    return a.ctypes.data_as(_lib.c_int_p)


def make_camera(c):
    k = Camera()
    k.fx, k.fy, k.cx, k.cy = c["fx"], c["fy"], c["cx"], c["cy"]
    for i in range(7):
        k.extrinsic[i] = float(c["extrinsic"][i])
    return k


class Context:
    def __init__(self, device=0, stream=None):
        self.h = C.c_void_p()
        _chk(_lib.lib().lvf_ctx_create(int(device), C.c_void_p(stream) if stream else None, C.byref(self.h)))
        self.L = _lib.lib()

    def synchronize(self):
        _chk(self.L.lvf_ctx_synchronize(self.h))

    def timer_begin(self):
        _chk(self.L.lvf_timer_begin(self.h))

    def timer_end(self):
        _chk(self.L.lvf_timer_end(self.h))

    def timer_ms(self):
        ms = C.c_float()
        _chk(self.L.lvf_timer_elapsed_ms(self.h, C.byref(ms)))
        return ms.value

    def close(self):
        if self.h:
            self.L.lvf_ctx_destroy(self.h)
            self.h = C.c_void_p()


class State:
    def __init__(self, ctx, n_kf, n_lm):
        self.ctx, self.n_kf, self.n_lm = ctx, n_kf, n_lm
        self.h = C.c_void_p()
        _chk(ctx.L.lvf_state_create(ctx.h, n_kf, n_lm, C.byref(self.h)))

    def _size(self, field):
        return {POSES: 7 * self.n_kf, VEL: 3 * self.n_kf, BA: 3 * self.n_kf, BG: 3 * self.n_kf, INV_DEPTH: self.n_lm,
                W_VISUAL: self.n_kf}[field]

    def set(self, field, arr):
        a = _d(arr)
        if a.size != self._size(field):
            raise ValueError(f"state field {field}: expected {self._size(field)} doubles, got {a.size}")
        _chk(self.ctx.L.lvf_state_set(self.h, field, _dp(a)))

    def get(self, field):
        out = np.empty(self._size(field))
        _chk(self.ctx.L.lvf_state_get(self.h, field, _dp(out)))
        return out

    def copy_from(self, other):
        """device-to-device copy of every field (asynchronous)"""
        _chk(self.ctx.L.lvf_state_copy(self.h, other.h))

    def close(self):
        if self.h:
            self.ctx.L.lvf_state_destroy(self.h)
            self.h = C.c_void_p()


class Batch:
    """One functor type's residual blocks.  evaluate() == batched CostFunction::Evaluate."""

    def __init__(self, ctx, h, n, n_res, block_sizes):
        self.ctx, self.h, self.n, self.n_res, self.block_sizes = ctx, h, n, n_res, tuple(block_sizes)

    def evaluate(self, state=None, rpyxyz=None, jacobians=True):
        r = _d(rpyxyz) if rpyxyz is not None else None
        _chk(self.ctx.L.lvf_batch_evaluate(self.h, state.h if state is not None else None, _dp(r) if r is not None else None,
                                           1 if jacobians else 0))

    def residuals(self):
        out = np.empty((self.n, self.n_res))
        _chk(self.ctx.L.lvf_batch_download_residuals(self.h, _dp(out)))
        return out

    def jacobian(self, block):
        out = np.empty((self.n, self.n_res, self.block_sizes[block]))
        _chk(self.ctx.L.lvf_batch_download_jacobian(self.h, block, _dp(out)))
        return out

    def evaluate_local(self, state, huber_a=0.0, jacobians=True):
        """Problem::Evaluate-shaped outputs: robustified residuals [n][R] and LOCAL Jacobians [n][R][L]."""
        L = self.ctx.L.lvf_batch_local_columns(self.h)
        r = np.empty((self.n, self.n_res)); J = np.empty((self.n, self.n_res, L)) if jacobians else None
        _chk(self.ctx.L.lvf_batch_evaluate_local(self.h, state.h, C.c_double(huber_a), _dp(r), _dp(J) if jacobians else None))
        return r, J

    def set_block_weights(self, weight):
        w = None if weight is None else _d(weight)
        _chk(self.ctx.L.lvf_two_camera_set_block_weights(self.h, _dp(w) if w is not None else None))

    def normals(self):
        out = np.empty((self.n, 3))
        _chk(self.ctx.L.lvf_batch_download_normals(self.h, _dp(out)))
        return out

    def close(self):
        if self.h:
            self.ctx.L.lvf_batch_destroy(self.h)
            self.h = C.c_void_p()


def pose_only_batch(ctx, cam0, ob, kf_idx, pw_idx, pw):
    ob, pw, kf_idx, pw_idx = _d(ob), _d(pw), _i(kf_idx), _i(pw_idx)
    h = C.c_void_p()
    cam = make_camera(cam0)
    _chk(ctx.L.lvf_pose_only_create(ctx.h, C.byref(cam), ob.shape[0], _dp(ob), _ip(kf_idx), _ip(pw_idx), pw.shape[0], _dp(pw), C.byref(h)))
    return Batch(ctx, h, ob.shape[0], 2, (7,))


def two_frame_batch(ctx, left, right, first_ob, ob, lm_idx, kf1_idx, kf2_idx):
    first_ob, ob = _d(first_ob), _d(ob)
    lm_idx, kf1_idx, kf2_idx = _i(lm_idx), _i(kf1_idx), _i(kf2_idx)
    h = C.c_void_p()
    cl, cr = make_camera(left), make_camera(right)
    _chk(ctx.L.lvf_two_frame_create(ctx.h, C.byref(cl), C.byref(cr), ob.shape[0], _dp(first_ob), _dp(ob), _ip(lm_idx), _ip(kf1_idx),
                                    _ip(kf2_idx), C.byref(h)))
    return Batch(ctx, h, ob.shape[0], 2, (1, 7, 7))


def two_camera_batch(ctx, left, right, left_ob, right_ob, lm_idx, kf_idx):
    left_ob, right_ob, lm_idx, kf_idx = _d(left_ob), _d(right_ob), _i(lm_idx), _i(kf_idx)
    h = C.c_void_p()
    cl, cr = make_camera(left), make_camera(right)
    _chk(ctx.L.lvf_two_camera_create(ctx.h, C.byref(cl), C.byref(cr), left_ob.shape[0], _dp(left_ob), _dp(right_ob), _ip(lm_idx),
                                     _ip(kf_idx), C.byref(h)))
    return Batch(ctx, h, left_ob.shape[0], 2, (1,))


def imu_batch(ctx, pre, kf_i, kf_j):
    pre, kf_i, kf_j = _d(pre), _i(kf_i), _i(kf_j)
    assert pre.ndim == 2 and pre.shape[1] == 467
    h = C.c_void_p()
    _chk(ctx.L.lvf_imu_create(ctx.h, pre.shape[0], _dp(pre), _ip(kf_i), _ip(kf_j), C.byref(h)))
    return Batch(ctx, h, pre.shape[0], 15, IMU_BLOCK_SIZES)


def lidar_plane_batch(ctx, mode, p, pa, pb, pc, Twc1, weight):
    p, pa, pb, pc, Twc1 = map(_d, (p, pa, pb, pc, Twc1))
    h = C.c_void_p()
    _chk(ctx.L.lvf_lidar_plane_create(ctx.h, int(mode), p.shape[0], _dp(p), _dp(pa), _dp(pb), _dp(pc), _dp(Twc1), float(weight), C.byref(h)))
    return Batch(ctx, h, p.shape[0], 1, (1, 1, 1))


def pose_prior_batch(ctx, kf_a, kf_b, target, weight, v):
    """kf_a[i] < 0: PoseError on pose kf_b[i] with origin target[i][:7]; else PoseGraphError(kf_a, kf_b) with
    target[i][:6] = rpyxyz_ (see relative_rpyxyz)."""
    kf_a, kf_b, target, weight, v = _i(kf_a), _i(kf_b), _d(target), _d(weight), _d(v)
    n = kf_a.shape[0]
    assert target.shape == (n, 7)
    h = C.c_void_p()
    _chk(ctx.L.lvf_pose_prior_create(ctx.h, n, _ip(kf_a), _ip(kf_b), _dp(target), _dp(weight), _dp(v), C.byref(h)))
    return Batch(ctx, h, n, 6, (7, 7))


def relative_rpyxyz(last_pose, pose):
    a, b, out = _d(last_pose), _d(pose), np.empty(6)
    _chk(_lib.lib().lvf_relative_rpyxyz(_dp(a), _dp(b), _dp(out)))
    return out


def relocate_r_evaluate(ctx, relocated, unrelocated, q4, jacobians=True):
    """RelocateRError <7,4> batched: returns r[n][7], J[n][7][4] (ambient) or None."""
    a, b, q = _d(relocated), _d(unrelocated), _d(q4)
    n = a.shape[0]
    r = np.empty((n, 7)); J = np.empty((n, 7, 4)) if jacobians else None
    _chk(ctx.L.lvf_relocate_r_evaluate(ctx.h, n, _dp(a), _dp(b), _dp(q), _dp(r), _dp(J) if jacobians else None))
    return r, J


def relocate_rotation_solve(ctx, relocated, unrelocated, q4, opt=None):
    """Relocator::UpdateNewSubmap's rotation solve; returns (q4_new, SolverSummary)."""
    a, b = _d(relocated), _d(unrelocated)
    q = _d(q4).copy()
    opt = opt or default_solver_options()
    summ = _lib.SolverSummary()
    _chk(ctx.L.lvf_relocate_rotation_solve(ctx.h, a.shape[0], _dp(a), _dp(b), _dp(q), C.byref(opt), C.byref(summ)))
    return q, summ


def forward_update(ctx, transform, poses, vw=None):
    """PoseGraph::ForwardUpdate on host arrays; returns updated copies."""
    T = _d(transform); P = _d(poses).copy(); V = None if vw is None else _d(vw).copy()
    _chk(ctx.L.lvf_forward_update(ctx.h, _dp(T), P.shape[0], _dp(P), _dp(V) if V is not None else None))
    return P, V


def prior3_evaluate(ctx, mode, target3, weight, x3):
    """PoseErrorRPZ / PoseErrorYXY stand-alone (lvf_prior3_evaluate): returns r[3], J[3 blocks][3 rows]."""
    t, x = _d(target3), _d(x3)
    r, J = np.empty(3), np.empty(9)
    _chk(ctx.L.lvf_prior3_evaluate(ctx.h, int(mode), _dp(t), C.c_double(weight), _dp(x), _dp(r), _dp(J)))
    return r, J.reshape(3, 3)


def preintegrate(ctx, samples_list, acc0, gyr0, ba, bg, noise4):
    n = len(samples_list)
    offset = np.zeros(n + 1, np.int32)
    offset[1:] = np.cumsum([s.shape[0] for s in samples_list])
    samples = _d(np.concatenate(samples_list)) if n else np.zeros((0, 7))
    acc0, gyr0, ba, bg, noise4 = map(_d, (acc0, gyr0, ba, bg, noise4))
    out = np.zeros((n, 467))
    _chk(ctx.L.lvf_preintegrate(ctx.h, n, _ip(offset), _dp(samples), _dp(acc0), _dp(gyr0), _dp(ba), _dp(bg), _dp(noise4), _dp(out)))
    return out


def preintegrate_or_none(ctx, cfg):
    """Device pre-integration of a synthetic window's IMU samples (None when the window has no IMU factors)."""
    from . import synthetic as syn
    imu = cfg["imu"]
    if not imu:
        return None
    return preintegrate(ctx, [f["samples"] for f in imu], np.stack([f["acc0"] for f in imu]), np.stack([f["gyr0"] for f in imu]),
                        np.stack([f["ba"] for f in imu]), np.stack([f["bg"] for f in imu]), syn.IMU_NOISE)


class Cloud:
    """Device-resident (x, y, z, intensity) float32 cloud; every method returns a new device cloud."""

    def __init__(self, ctx, xyzi=None, _h=None):
        self.ctx = ctx
        if _h is not None:
            self.h = _h
        else:
            a = _f(xyzi)
            assert a.ndim == 2 and a.shape[1] >= 3
            self.h = C.c_void_p()
            _chk(ctx.L.lvf_cloud_create(ctx.h, a.ctypes.data_as(_lib.c_float_p), a.shape[0], a.shape[1], 3 if a.shape[1] > 3 else -1, C.byref(self.h)))

    def __len__(self):
        return self.ctx.L.lvf_cloud_size(self.h)

    def download(self):
        out = np.empty((len(self), 4), np.float32)
        _chk(self.ctx.L.lvf_cloud_download(self.h, out.ctypes.data_as(_lib.c_float_p)))
        return out

    def _new(self, fn, *args):
        h = C.c_void_p()
        _chk(fn(self.h, *args, C.byref(h)))
        return Cloud(self.ctx, _h=h)

    def transform(self, pose):
        p = _d(pose)
        return self._new(self.ctx.L.lvf_cloud_transform, _dp(p))

    def voxel_filter(self, leaf):
        return self._new(self.ctx.L.lvf_cloud_voxel_filter, float(leaf))

    def radius_outlier_filter(self, radius, min_neighbors):
        return self._new(self.ctx.L.lvf_cloud_radius_outlier_filter, float(radius), int(min_neighbors))

    def segment_plane(self, thr, max_iterations=100, seed=12345):
        h = C.c_void_p(); co = np.empty(4); it = C.c_int()
        _chk(self.ctx.L.lvf_cloud_segment_plane(self.h, float(thr), int(max_iterations), int(seed), C.byref(h), _dp(co), C.byref(it)))
        return Cloud(self.ctx, _h=h), co, it.value

    @staticmethod
    def concat(ctx, parts):
        arr = (C.c_void_p * len(parts))(*[p.h for p in parts])
        h = C.c_void_p()
        _chk(ctx.L.lvf_cloud_concat(ctx.h, arr, len(parts), C.byref(h)))
        return Cloud(ctx, _h=h)

    @staticmethod
    def align_scan(pc1, stamp1, pc2, stamp2, cycle_time, time):
        """FeatureAssociation::AlignScan on device: (Cloud, True) or (empty Cloud, False) where the reference returns false."""
        h = C.c_void_p(); ok = C.c_int()
        _chk(pc1.ctx.L.lvf_cloud_align_scan(pc1.h, float(stamp1), pc2.h, float(stamp2), float(cycle_time), float(time), C.byref(h), C.byref(ok)))
        return Cloud(pc1.ctx, _h=h), bool(ok.value)

    def close(self):
        if self.h:
            self.ctx.L.lvf_cloud_destroy(self.h)
            self.h = C.c_void_p()


def lidar_params(**kw):
    p = LidarParams()
    _lib.lib().lvf_lidar_params_default(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def extract_host_counts(ctx, on):
    """test / A-B hook: route lidar_extract through the host-counted path (process-wide); returns the previous setting"""
    return bool(ctx.L.lvf_debug_extract_host_counts(1 if on else 0))


def lidar_extract(ctx, points, extrinsic, params=None, debug=False):
    """FeatureAssociation::Process on device: raw sensor-frame scan -> (ground Cloud, surf Cloud[, debug dict])."""
    a = _f(points); e = _d(extrinsic)
    prm = params if params is not None else lidar_params()
    hg, hs = C.c_void_p(), C.c_void_p()
    dbg = None
    if debug:
        npix = prm.num_scans * prm.horizon_scan
        cap = max(npix, 1)
        keep = dict(label_mat=np.zeros(npix, np.int32), ground_mat=np.zeros(npix, np.int8), range_mat=np.zeros(npix, np.float32),
                    ground_raw=np.zeros((cap, 4), np.float32), surf_raw=np.zeros((cap, 4), np.float32))
        dbg = LidarExtractDebug()
        dbg.label_mat = keep["label_mat"].ctypes.data_as(C.POINTER(C.c_int32)); dbg.ground_mat = keep["ground_mat"].ctypes.data_as(C.POINTER(C.c_int8))
        dbg.range_mat = keep["range_mat"].ctypes.data_as(_lib.c_float_p); dbg.ground_raw = keep["ground_raw"].ctypes.data_as(_lib.c_float_p)
        dbg.surf_raw = keep["surf_raw"].ctypes.data_as(_lib.c_float_p)
    _chk(ctx.L.lvf_lidar_extract(ctx.h, a.ctypes.data_as(_lib.c_float_p), a.shape[0], a.shape[1], C.byref(prm), _dp(e), C.byref(hg), C.byref(hs),
                                 C.byref(dbg) if dbg is not None else None))
    g, s = Cloud(ctx, _h=hg), Cloud(ctx, _h=hs)
    if not debug:
        return g, s
    R, Cn = prm.num_scans, prm.horizon_scan
    out = dict(label_mat=keep["label_mat"].reshape(R, Cn), ground_mat=keep["ground_mat"].reshape(R, Cn), range_mat=keep["range_mat"].reshape(R, Cn),
               ground_raw=keep["ground_raw"][:dbg.n_ground_raw].copy(), surf_raw=keep["surf_raw"][:dbg.n_surf_raw].copy(),
               n_filtered=dbg.n_filtered, n_segmented=dbg.n_segmented)
    return g, s, out


class Map:
    def __init__(self, ctx, xyz, max_radius2):
        self.ctx = ctx
        self.h = C.c_void_p()
        if isinstance(xyz, Cloud):
            self.M = len(xyz)
            _chk(ctx.L.lvf_map_create_from_cloud(xyz.h, float(max_radius2), C.byref(self.h)))
            return
        a = _f(xyz)
        self.ctx, self.M = ctx, a.shape[0]
        _chk(ctx.L.lvf_map_create(ctx.h, a.ctypes.data_as(_lib.c_float_p), a.shape[0], a.shape[1] if a.ndim == 2 else 3,
                                  float(max_radius2), C.byref(self.h)))

    def close(self):
        if self.h:
            self.ctx.L.lvf_map_destroy(self.h)
            self.h = C.c_void_p()

    @classmethod
    def create_batch(cls, ctx, clouds, max_radius2):
        """lvf_map_create_batch: one Map per cloud (host arrays of one column count, or device-resident Cloud objects: no upload), the host
        waits shared between them"""
        n = len(clouds)
        if n and all(isinstance(c, Cloud) for c in clouds):
            thr = np.ascontiguousarray(np.broadcast_to(np.asarray(max_radius2, np.float32), (n,)))
            hc = (C.c_void_p * n)(*[c.h for c in clouds])
            hs = (C.c_void_p * n)()
            _chk(ctx.L.lvf_map_create_batch_from_clouds(ctx.h, n, hc, thr.ctypes.data_as(_lib.c_float_p), hs))
            out = []
            for i in range(n):
                m = cls.__new__(cls)
                m.ctx, m.M, m.h = ctx, len(clouds[i]), C.c_void_p(hs[i])
                out.append(m)
            return out
        arrs = [_f(c) for c in clouds]
        thr = np.ascontiguousarray(np.broadcast_to(np.asarray(max_radius2, np.float32), (n,)))
        if n == 0:
            return []
        stride = arrs[0].shape[1] if arrs[0].ndim == 2 else 3
        assert all((a.shape[1] if a.ndim == 2 else 3) == stride for a in arrs), "clouds of one batch share their column count"
        ptrs = (_lib.c_float_p * n)(*[a.ctypes.data_as(_lib.c_float_p) for a in arrs])
        Ms = np.array([a.shape[0] for a in arrs], np.int32)
        hs = (C.c_void_p * n)()
        _chk(ctx.L.lvf_map_create_batch(ctx.h, n, ptrs, Ms.ctypes.data_as(_lib.c_int_p), stride, thr.ctypes.data_as(_lib.c_float_p), hs))
        out = []
        for i in range(n):
            m = cls.__new__(cls)
            m.ctx, m.M, m.h = ctx, int(Ms[i]), C.c_void_p(hs[i])
            out.append(m)
        return out


class Scan:
    def __init__(self, ctx, xyz):
        self.ctx = ctx
        self.h = C.c_void_p()
        if isinstance(xyz, Cloud):
            self.Q = len(xyz)
            _chk(ctx.L.lvf_scan_create_from_cloud(xyz.h, C.byref(self.h)))
            return
        a = _f(xyz)
        self.ctx, self.Q = ctx, a.shape[0]
        _chk(ctx.L.lvf_scan_create(ctx.h, a.ctypes.data_as(_lib.c_float_p), a.shape[0], a.shape[1] if a.ndim == 2 else 3, C.byref(self.h)))

    def download(self):
        idx = np.empty((self.Q, 3), np.int32); d2 = np.empty((self.Q, 3), np.float32); valid = np.empty(self.Q, np.uint8)
        _chk(self.ctx.L.lvf_scan_download(self.h, _ip(idx), d2.ctypes.data_as(_lib.c_float_p), valid.ctypes.data_as(_lib.c_u8_p)))
        return idx, d2, valid

    def close(self):
        if self.h:
            self.ctx.L.lvf_scan_destroy(self.h)
            self.h = C.c_void_p()


def knn3(map_, scan, pose, thr):
    p = _d(pose)
    _chk(map_.ctx.L.lvf_knn3(map_.h, scan.h, _dp(p), float(thr)))


def icp_solve(map_, scan, map_pose, frame_pose, rpyxyz, mode, thr, weight, huber_a, prior_weight=0.0, max_num_iterations=4):
    """rpyxyz is updated IN PLACE (numpy float64[6]); returns the summary struct."""
    assert rpyxyz.dtype == np.float64 and rpyxyz.flags.c_contiguous and rpyxyz.size == 6
    mp, fp = _d(map_pose), _d(frame_pose)
    opt = IcpOptions(int(mode), float(thr), float(weight), float(huber_a), float(prior_weight), int(max_num_iterations))
    summ = IcpSummary()
    _chk(map_.ctx.L.lvf_icp_solve(map_.h, scan.h, _dp(mp), _dp(fp), _dp(rpyxyz), C.byref(opt), C.byref(summ)))
    return summ


def scan_match_options(resolution=0.2, outer_iterations=1, prior_weight=0.0):
    o = ScanMatchOptions()
    _lib.lib().lvf_scan_match_options_default(C.byref(o), float(resolution))
    o.outer_iterations, o.prior_weight = int(outer_iterations), float(prior_weight)
    return o


def scan_match(map_ground, scan_ground, map_surf, scan_surf, map_pose, frame_pose, opt, last_pose=None):
    """Mapping::Optimize's per-frame body (outer_iterations=1, prior) / Mapping::Relocate (outer_iterations=4, no prior)."""
    mp, fp = _d(map_pose), _d(frame_pose)
    lp = _d(last_pose) if last_pose is not None else None
    res = ScanMatchResult()
    anyh = map_ground if map_ground is not None else map_surf
    h = lambda x: x.h if x is not None else None
    _chk(anyh.ctx.L.lvf_scan_match(h(map_ground), h(scan_ground), h(map_surf), h(scan_surf), _dp(mp), _dp(fp), _dp(lp) if lp is not None else None,
                                   C.byref(opt), C.byref(res)))
    return res


def scan_match_batch(ctx, jobs, opt, relocate_base_score=20):
    """Many scan-to-map updates side by side (lvf_scan_match_batch).  jobs: list of dicts with map_ground / scan_ground / map_surf /
    scan_surf (handles or None), map_pose, frame_pose, last_pose (or None).  Returns (results list, best index or -1)."""
    n = len(jobs)
    arr = (ScanMatchJob * max(n, 1))()
    h = lambda x: x.h if x is not None else None
    for j, a in zip(jobs, arr):
        a.map_ground, a.scan_ground, a.map_surf, a.scan_surf = h(j.get("map_ground")), h(j.get("scan_ground")), h(j.get("map_surf")), h(j.get("scan_surf"))
        a.map_pose[:] = list(_d(j["map_pose"])); a.frame_pose[:] = list(_d(j["frame_pose"]))
        lp = j.get("last_pose")
        a.has_last_pose = 0 if lp is None else 1
        if lp is not None:
            a.last_pose[:] = list(_d(lp))
    res = (ScanMatchResult * max(n, 1))()
    best = C.c_int32(-1)
    _chk(ctx.L.lvf_scan_match_batch(ctx.h, arr, n, C.byref(opt), int(relocate_base_score), res, C.byref(best)))
    return list(res)[:n], int(best.value)


def lidar_solve(batch, rpyxyz, huber_a, prior_weight=0.0, max_num_iterations=4):
    """3-DoF LM over a caller-built lidar batch; rpyxyz (float64[6]) is updated IN PLACE."""
    assert rpyxyz.dtype == np.float64 and rpyxyz.flags.c_contiguous and rpyxyz.size == 6
    opt = IcpOptions(0, 0.0, 0.0, float(huber_a), float(prior_weight), int(max_num_iterations))
    summ = IcpSummary()
    _chk(batch.ctx.L.lvf_lidar_solve(batch.h, _dp(rpyxyz), C.byref(opt), C.byref(summ)))
    return summ


class Window:
    """Persistent sliding window (lvf_window_*): Backend::BuildProblem's assembly kept incrementally across ticks."""

    def __init__(self, ctx, left, right, baseline=None, weak_visual_threshold=20, device_assembly=None):
        self.ctx = ctx
        o = WindowOptions()
        ctx.L.lvf_window_options_default(C.byref(o))
        if baseline is not None:
            o.baseline = float(baseline)
        o.weak_visual_threshold = int(weak_visual_threshold)
        if device_assembly is not None:
            o.device_assembly = 1 if device_assembly else 0
        self.h = C.c_void_p()
        cl, cr = make_camera(left), make_camera(right)
        _chk(ctx.L.lvf_window_create(ctx.h, C.byref(cl), C.byref(cr), C.byref(o), C.byref(self.h)))

    def add_keyframe(self, kf_id, pose, w_visual):
        p = _d(pose)
        _chk(self.ctx.L.lvf_window_add_keyframe(self.h, int(kf_id), _dp(p), float(w_visual)))

    def set_imu(self, kf_id, vel, ba, bg, pre=None):
        v, a, g = _d(vel), _d(ba), _d(bg)
        pr = _d(pre) if pre is not None else None
        _chk(self.ctx.L.lvf_window_set_imu(self.h, int(kf_id), _dp(v), _dp(a), _dp(g), _dp(pr) if pr is not None else None))

    def add_landmark(self, lm_id, birth_kf, left_ob, right_ob, inv_depth):
        l, r = _d(left_ob), _d(right_ob)
        _chk(self.ctx.L.lvf_window_add_landmark(self.h, int(lm_id), int(birth_kf), _dp(l), _dp(r), float(inv_depth)))

    def add_observation(self, lm_id, kf_id, ob):
        o = _d(ob)
        _chk(self.ctx.L.lvf_window_add_observation(self.h, int(lm_id), int(kf_id), _dp(o)))

    def remove_observation(self, lm_id, kf_id):
        _chk(self.ctx.L.lvf_window_remove_observation(self.h, int(lm_id), int(kf_id)))

    def reject_outliers(self, max_px=10.0, capacity=4096):
        """Backend::Optimize's outlier gate; returns the removed (landmark id, keyframe id) pairs."""
        lm = (C.c_int64 * capacity)(); kf = (C.c_int64 * capacity)(); n = C.c_int(0)
        _chk(self.ctx.L.lvf_window_reject_outliers(self.h, C.c_double(max_px), lm, kf, capacity, C.byref(n)))
        return [(int(lm[i]), int(kf[i])) for i in range(min(n.value, capacity))], n.value

    def slide(self, first_active_kf):
        _chk(self.ctx.L.lvf_window_slide(self.h, int(first_active_kf)))

    def solve(self, opt):
        s = SolverSummary()
        _chk(self.ctx.L.lvf_window_solve(self.h, C.byref(opt), C.byref(s)))
        return s

    def pose(self, kf_id):
        out = np.empty(7)
        _chk(self.ctx.L.lvf_window_get_pose(self.h, int(kf_id), _dp(out)))
        return out

    def set_pose(self, kf_id, pose):
        p = _d(pose)
        _chk(self.ctx.L.lvf_window_set_pose(self.h, int(kf_id), _dp(p)))

    def imu(self, kf_id):
        v, a, g = np.empty(3), np.empty(3), np.empty(3)
        _chk(self.ctx.L.lvf_window_get_imu(self.h, int(kf_id), _dp(v), _dp(a), _dp(g)))
        return v, a, g

    def inv_depth(self, lm_id):
        d = C.c_double()
        _chk(self.ctx.L.lvf_window_get_inv_depth(self.h, int(lm_id), C.byref(d)))
        return d.value

    def debug_blocks(self, kind):
        """(ids [n][3] int64, vals [n][8]) of the last solve's blocks of one kind (lvf_window_debug_blocks: 0 TwoCamera, 1 PoseOnly, 2 TwoFrame,
        3 ImuError, 4 priors)"""
        n = C.c_int()
        _chk(self.ctx.L.lvf_window_debug_blocks(self.h, int(kind), 0, None, None, C.byref(n)))
        ids = np.zeros((max(n.value, 1), 3), np.int64); vals = np.zeros((max(n.value, 1), 8))
        _chk(self.ctx.L.lvf_window_debug_blocks(self.h, int(kind), n.value, ids.ctypes.data_as(C.c_void_p), _dp(vals), C.byref(n)))
        return ids[:n.value], vals[:n.value]

    def counts(self):
        c = np.zeros(8, np.int32)
        _chk(self.ctx.L.lvf_window_counts(self.h, _ip(c)))
        return dict(zip(("kf", "lm", "tc", "tf", "po", "imu", "prior", "lm_known"), c.tolist()))

    def close(self):
        if self.h:
            self.ctx.L.lvf_window_destroy(self.h)
            self.h = C.c_void_p()


def event_pair_us(ctx, reps=25):
    """Microseconds a HIP event pair measures with NOTHING enqueued between its two records (median of `reps`): the marker's own cost,
    which every event-bracketed stage time carries once (bench.py subtracts it so stage times agree with rocprofv3's kernel durations)."""
    ctx.synchronize()
    v = []
    for _ in range(reps):
        ctx.timer_begin(); ctx.timer_end()
        v.append(1e3 * ctx.timer_ms())
    return float(np.median(v))


def box_calibration(ctx):
    """lvf_box_calibration as a dict (see include/lvf.h)"""
    v = np.zeros(8)
    _chk(ctx.L.lvf_box_calibration(ctx.h, _dp(v)))
    return {"ns_per_dependent_fp64_fma": float(v[0]), "shader_clocks_per_dependent_fp64_fma": float(v[1]), "effective_sclk_mhz": float(v[2]),
            "empty_launch_us_back_to_back": float(v[3]), "empty_launch_plus_wait_us": float(v[4]), "rated_sclk_mhz": float(v[5]),
            "rated_mclk_mhz": float(v[6]), "compute_units": int(v[7])}


def comm_unique_id():
    """lvf_comm_get_unique_id: the 128 bytes rank 0 hands to the other ranks out of band"""
    buf = (C.c_ubyte * 128)()
    _chk(_lib.lib().lvf_comm_get_unique_id(C.cast(buf, C.c_void_p)))
    return bytes(buf)


class Comm:
    """lvf_comm_*: the path's one exchange through the C-ABI (RCCL opened by the library; world_size 1 needs none)"""
    def __init__(self, ctx, world_size=1, rank=0, unique_id=None):
        self.ctx, self.h = ctx, C.c_void_p()
        idbuf = None
        if unique_id is not None:
            if len(unique_id) != 128:
                raise ValueError("unique id must be the 128 bytes of comm_unique_id()")
            idbuf = C.cast((C.c_ubyte * 128).from_buffer_copy(unique_id), C.c_void_p)
        _chk(ctx.L.lvf_comm_create(ctx.h, int(world_size), int(rank), idbuf, C.byref(self.h)))

    @property
    def world_size(self):
        return int(self.ctx.L.lvf_comm_world_size(self.h))

    @property
    def rank(self):
        return int(self.ctx.L.lvf_comm_rank(self.h))

    def allgather(self, send):
        a = _d(send).ravel()
        out = np.empty((self.world_size, a.size))
        _chk(self.ctx.L.lvf_comm_allgather(self.h, _dp(a), int(a.size), _dp(out)))
        return out

    def close(self):
        if self.h:
            self.ctx.L.lvf_comm_destroy(self.h)
            self.h = C.c_void_p()


def default_solver_options():
    o = SolverOptions()
    _lib.lib().lvf_solver_options_default(C.byref(o))
    return o


class Problem:
    def __init__(self, ctx, state, two_camera=None, two_frame=None, pose_only=None, imu=None):
        self.ctx, self.state = ctx, state
        self.h = C.c_void_p()
        hs = [b.h if b is not None else None for b in (two_camera, two_frame, pose_only, imu)]
        _chk(ctx.L.lvf_problem_create(ctx.h, state.h, hs[0], hs[1], hs[2], hs[3], C.byref(self.h)))

    def set_pose_priors(self, batch):
        _chk(self.ctx.L.lvf_problem_set_pose_priors(self.h, batch.h if batch is not None else None))

    def set_pose_constant(self, kf, const=True):
        _chk(self.ctx.L.lvf_problem_set_pose_constant(self.h, int(kf), 1 if const else 0))

    def set_vbb_constant(self, kf, v=True, ba=True, bg=True):
        _chk(self.ctx.L.lvf_problem_set_vbb_constant(self.h, int(kf), int(bool(v)), int(bool(ba)), int(bool(bg))))

    def cost(self, opt):
        c = C.c_double()
        _chk(self.ctx.L.lvf_problem_cost(self.h, C.byref(opt), C.byref(c)))
        return c.value

    def lm_iteration(self, opt, radius, decrease_factor=2.0):
        r, d, c0, c1, acc = C.c_double(radius), C.c_double(decrease_factor), C.c_double(), C.c_double(), C.c_int()
        _chk(self.ctx.L.lvf_problem_lm_iteration(self.h, C.byref(opt), C.byref(r), C.byref(d), C.byref(c0), C.byref(c1), C.byref(acc)))
        return dict(radius=r.value, decrease_factor=d.value, cost_before=c0.value, cost_after=c1.value, accepted=bool(acc.value))

    def solve(self, opt):
        s = SolverSummary()
        _chk(self.ctx.L.lvf_problem_solve(self.h, C.byref(opt), C.byref(s)))
        return s

    def debug_force_handover_timeout(self, n=1):
        """test hook: the next n chained hand-overs of this problem time out (lvf_problem_debug_force_handover_timeout)"""
        _chk(self.ctx.L.lvf_problem_debug_force_handover_timeout(self.h, int(n)))

    def stage_times(self, opt, radius=1e4, reps=10, spans=False):
        """[(stage name, average microseconds, launches)] of `reps` LM iterations from the current state: the sum of the stage's KERNEL
        durations (start / stop events recorded with every launch — the dispatch timestamps rocprofv3's kernel trace reports).
        spans=True appends the between-stage event span (kernels + gaps + marker cost) as a fourth entry."""
        L = self.ctx.L
        n = L.lvf_problem_stage_count()
        us = np.zeros(n); sp = np.zeros(n); la = (C.c_int * n)()
        _chk(L.lvf_problem_stage_times2(self.h, C.byref(opt), float(radius), int(reps), _dp(us), _dp(sp), la))
        if spans:
            return [(L.lvf_problem_stage_name(i).decode(), float(us[i]), int(la[i]), float(sp[i])) for i in range(n)]
        return [(L.lvf_problem_stage_name(i).decode(), float(us[i]), int(la[i])) for i in range(n)]

    def gradient(self, opt):
        d = self.ctx.L.lvf_problem_reduced_dim(self.h)
        gc = np.empty(d); gl = np.empty(max(self.state.n_lm, 1))
        _chk(self.ctx.L.lvf_problem_gradient(self.h, C.byref(opt), _dp(gc), _dp(gl)))
        return gc, gl[:self.state.n_lm]

    def reduced_system(self):
        d = self.ctx.L.lvf_problem_reduced_dim(self.h)
        S = np.empty((d, d)); rhs = np.empty(d)
        _chk(self.ctx.L.lvf_problem_download_reduced(self.h, _dp(S), _dp(rhs)))
        return S, rhs

    def close(self):
        if self.h:
            self.ctx.L.lvf_problem_destroy(self.h)
            self.h = C.c_void_p()


class ProblemBatch:
    """W independent windows advanced by one chain of launches per LM iteration (lvf_problem_batch_*)."""

    def __init__(self, ctx, problems):
        self.ctx, self.problems = ctx, list(problems)
        self.h = C.c_void_p()
        arr = (C.c_void_p * len(self.problems))(*[p.h for p in self.problems])
        _chk(ctx.L.lvf_problem_batch_create(ctx.h, arr, len(self.problems), C.byref(self.h)))

    def uses_tables(self, opt):
        return self.ctx.L.lvf_problem_batch_uses_tables(self.h, C.byref(opt))

    def lm_iteration(self, opt, radius, decrease_factor):
        n = len(self.problems)
        r, d = _d(radius).copy(), _d(decrease_factor).copy()
        c0, c1, acc = np.empty(n), np.empty(n), np.zeros(n, np.int32)
        _chk(self.ctx.L.lvf_problem_batch_lm_iteration(self.h, C.byref(opt), _dp(r), _dp(d), _dp(c0), _dp(c1), _ip(acc)))
        return [dict(radius=r[i], decrease_factor=d[i], cost_before=c0[i], cost_after=c1[i], accepted=bool(acc[i])) for i in range(n)]

    def solve(self, opt):
        out = (SolverSummary * len(self.problems))()
        _chk(self.ctx.L.lvf_problem_batch_solve(self.h, C.byref(opt), out))
        return list(out)

    def close(self):
        if self.h:
            self.ctx.L.lvf_problem_batch_destroy(self.h)
            self.h = C.c_void_p()
