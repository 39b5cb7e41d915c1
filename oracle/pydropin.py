"""TEST INFRASTRUCTURE ONLY.  ctypes front-end of oracle/_ref/liblvf_dropin.so — the COMPILED DROP-IN: the reference's backend.cpp /
association.cpp / landmark.cpp / preintegration.cpp compiled UNMODIFIED with include/reference_patch ahead of the reference's include
directory, linked against lvio_fusion_amd/liblvf_hip.so (recipe: oracle/Makefile target `dropin`, driver: oracle/ref_driver_dropin.cpp).
`Backend::BuildProblem -> adapt::Solve` and `ScanToMapWithGround/Segmented -> adapt::Solve` run ON THE MI355X from the reference's text.

/root/reference only exists in the build container: the library is built there (__graft_entry__.build) and travels to the GPU box
with the snapshot (oracle/_ref/ is git-ignored, not gpurun-ignored).  Used by tests/test_gpu_dropin.py only."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_ref", "liblvf_dropin.so")
REFERENCE_INCLUDE = "/root/reference/src/lvio_fusion/include"


def can_build():
    return os.path.isdir(os.path.join(REFERENCE_INCLUDE, "lvio_fusion", "ceres"))


def build(force=False):
    if can_build():
        subprocess.check_call(["make", "-C", _HERE, "-s", "dropin"] + (["-B"] if force else []))
    return _SO if os.path.exists(_SO) else None


def available():
    return os.path.exists(_SO) or can_build()


_lib = None


def lib():
    global _lib
    if _lib is None:
        if build() is None:
            raise RuntimeError("oracle/_ref/liblvf_dropin.so is not built and /root/reference is absent")
        _lib = C.CDLL(_SO)
        _lib.lvd_sources.restype = C.c_char_p
    return _lib


class _Camera(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double), ("extrinsic", C.c_double * 7)]


class _Input(C.Structure):
    _fields_ = [("n_frames", C.c_int), ("first_active", C.c_int), ("time", C.c_void_p), ("pose", C.c_void_p), ("w_visual", C.c_void_p), ("good_imu", C.c_void_p),
                ("imu_initialized", C.c_int), ("vel", C.c_void_p), ("ba", C.c_void_p), ("bg", C.c_void_p), ("imu_ns", C.c_void_p), ("imu_samples", C.c_void_p),
                ("imu_acc0", C.c_void_p), ("imu_gyr0", C.c_void_p), ("pre_ba", C.c_void_p), ("pre_bg", C.c_void_p), ("imu_noise4", C.c_void_p),
                ("n_lm", C.c_int), ("lm_id", C.c_void_p), ("lm_birth", C.c_void_p), ("lm_inv_depth", C.c_void_p), ("lm_right_ob", C.c_void_p),
                ("n_obs", C.c_int), ("obs_lm", C.c_void_p), ("obs_frame", C.c_void_p), ("obs_xy", C.c_void_p)]


def _cam(c):
    o = _Camera(); o.fx, o.fy, o.cx, o.cy = c["fx"], c["fy"], c["cx"], c["cy"]
    for i in range(7):
        o.extrinsic[i] = float(c["extrinsic"][i])
    return o


def backend_solve(cam0, cam1, baseline, time, pose, w_visual, good_imu, first_active, imu_initialized, lm_id, lm_birth, lm_inv_depth, lm_right_ob, obs_lm, obs_frame,
                  obs_xy, max_num_iterations, vel=None, ba=None, bg=None, imu=None, imu_noise=None):
    """The reference's Backend::BuildProblem over the window given as flat arrays (oracle/pyref.backend_build_problem's layout), then adapt::Solve =
    gpu::Solve with Backend::Optimize's options.  `imu`: per frame k >= 1 a dict(samples [ns][7], acc0, gyr0, ba, bg) of the pre-integration (k-1 -> k), or None.
    Returns dict(pose [n][7], inv_depth [n_lm], vel, ba, bg, initial_cost, final_cost, num_successful_steps, num_unsuccessful_steps, num_residual_blocks,
    termination_type, num_frames, recorded, message)."""
    n = len(time)
    z3 = np.zeros((n, 3))
    vel = z3 if vel is None else vel; ba = z3 if ba is None else ba; bg = z3 if bg is None else bg
    ns = np.zeros(n, np.int32); samples = []; acc0 = np.zeros((n, 3)); gyr0 = np.zeros((n, 3)); pba = np.zeros((n, 3)); pbg = np.zeros((n, 3))
    if imu is not None:
        for k in range(1, n):
            f = imu[k]
            if f is None:
                continue
            s = np.asarray(f["samples"], np.float64).reshape(-1, 7)
            ns[k] = len(s); samples.append(s); acc0[k] = f["acc0"]; gyr0[k] = f["gyr0"]; pba[k] = f["ba"]; pbg[k] = f["bg"]
    samples = np.concatenate(samples) if samples else np.zeros((0, 7))
    noise = np.ascontiguousarray(imu_noise if imu_noise is not None else np.zeros(4), np.float64)
    keep = [np.ascontiguousarray(a, t) for a, t in ((time, np.float64), (pose, np.float64), (w_visual, np.float64), (good_imu, np.uint8), (vel, np.float64), (ba, np.float64),
                                                   (bg, np.float64), (ns, np.int32), (samples, np.float64), (acc0, np.float64), (gyr0, np.float64), (pba, np.float64),
                                                   (pbg, np.float64), (noise, np.float64), (lm_id, np.int64), (lm_birth, np.int32), (lm_inv_depth, np.float64),
                                                   (lm_right_ob, np.float64), (obs_lm, np.int32), (obs_frame, np.int32), (obs_xy, np.float64))]
    vp = lambda x: x.ctypes.data_as(C.c_void_p)
    a = _Input()
    a.n_frames, a.first_active, a.imu_initialized, a.n_lm, a.n_obs = n, int(first_active), int(bool(imu_initialized)), len(keep[14]), len(keep[18])
    (a.time, a.pose, a.w_visual, a.good_imu, a.vel, a.ba, a.bg, a.imu_ns, a.imu_samples, a.imu_acc0, a.imu_gyr0, a.pre_ba, a.pre_bg, a.imu_noise4, a.lm_id, a.lm_birth,
     a.lm_inv_depth, a.lm_right_ob, a.obs_lm, a.obs_frame, a.obs_xy) = map(vp, keep)
    pose_o = np.empty((n, 7)); invd_o = np.empty(a.n_lm); vel_o = np.zeros((n, 3)); ba_o = np.zeros((n, 3)); bg_o = np.zeros((n, 3)); s8 = np.zeros(8)
    msg = C.create_string_buffer(512); t4 = np.zeros(4)
    c0, c1 = _cam(cam0), _cam(cam1)
    L = lib()
    L.lvd_backend_solve.restype = C.c_int
    rc = L.lvd_backend_solve(C.byref(c0), C.byref(c1), C.c_double(baseline), C.byref(a), int(max_num_iterations), vp(pose_o), vp(invd_o), vp(vel_o), vp(ba_o), vp(bg_o), vp(s8),
                             msg, 512, vp(t4))
    return dict(rc=rc, pose=pose_o, inv_depth=invd_o, vel=vel_o, ba=ba_o, bg=bg_o, initial_cost=s8[0], final_cost=s8[1], num_successful_steps=int(s8[2]),
                num_unsuccessful_steps=int(s8[3]), num_residual_blocks=int(s8[4]), termination_type=int(s8[5]), num_frames=int(s8[6]), recorded=bool(s8[7]),
                message=msg.value.decode(errors="replace"),
                times_ms=dict(object_graph=t4[0], build_problem=t4[1], adapt_solve=t4[2], destroy_and_read_back=t4[3]))


def scan_to_map_solve(mode, scan, map_pts, frame_pose, map_pose, para6, w_ground, w_surf, w_visual, n_features_left, relocate, resolution, max_num_iterations):
    """FeatureAssociation::ScanToMapWithGround (mode 0) / WithSegmented (mode 1) + the solve of mapping.cpp:158-164 through adapt::Solve = gpu::Solve.
    scan / map_pts: [n][4] float32 (x, y, z, intensity).  Returns dict(para [6] (the updated rpyxyz), final_cost, num_residual_blocks_reduced, termination_type, n_lidar, message)."""
    scan = np.ascontiguousarray(scan, np.float32); map_pts = np.ascontiguousarray(map_pts, np.float32)
    fp = np.ascontiguousarray(frame_pose, np.float64); mp = np.ascontiguousarray(map_pose, np.float64)
    para = np.array(para6, np.float64).copy()
    s4 = np.zeros(4); msg = C.create_string_buffer(512)
    vp = lambda x: x.ctypes.data_as(C.c_void_p)
    L = lib()
    L.lvd_scan_to_map_solve.restype = C.c_int
    rc = L.lvd_scan_to_map_solve(int(mode), vp(scan), len(scan), vp(map_pts), len(map_pts), vp(fp), vp(mp), vp(para), C.c_double(w_ground), C.c_double(w_surf),
                                 C.c_double(w_visual), int(n_features_left), int(bool(relocate)), C.c_double(resolution), int(max_num_iterations), vp(s4), msg, 512)
    return dict(rc=rc, para=para, final_cost=s4[0], num_residual_blocks_reduced=int(s4[1]), termination_type=int(s4[2]), n_lidar=int(s4[3]), message=msg.value.decode(errors="replace"))


# ---- the reference's CONTROL code on the GPU: mapping.cpp, pose_graph.cpp, relocator.cpp compiled unmodified into this library too (ceres::Solve =
# gpu::Solve).  Same entry points and marshalling as oracle/pyref.py's CPU pins (oracle/ref_driver_mapping.cpp serves both builds).
def _through_dropin(fn_name, *args, **kw):
    from . import pyref
    saved = pyref._lib
    pyref._lib = lib()
    try:
        return getattr(pyref, fn_name)(*args, **kw)
    finally:
        pyref._lib = saved


def mapping_optimize(*a, **k):
    """Mapping::Optimize (mapping.cpp:139-191) from the reference's text, every adapt::Solve on the MI355X"""
    return _through_dropin("mapping_optimize", *a, **k)


def mapping_relocate(*a, **k):
    """Mapping::Relocate (mapping.cpp:251-300) from the reference's text, every ceres::Solve on the MI355X"""
    return _through_dropin("mapping_relocate", *a, **k)


def pose_graph_optimize(*a, **k):
    """PoseGraph::BuildProblem + Optimize (pose_graph.cpp:163-224) from the reference's text, ceres::Solve on the MI355X"""
    return _through_dropin("pose_graph_optimize", *a, **k)


def update_new_submap(*a, **k):
    """Relocator::UpdateNewSubmap (relocator.cpp:247-282) from the reference's text: RelocateRError::Create -> gpu::RelocateRError, ceres::Solve ->
    lvf_relocate_rotation_solve"""
    return _through_dropin("update_new_submap", *a, **k)


def environment_optimize(*a, **k):
    """Environment::Optimize (environment.cpp:18-115) from the reference's text, adapt::Solve on the MI355X"""
    return _through_dropin("environment_optimize", *a, **k)
