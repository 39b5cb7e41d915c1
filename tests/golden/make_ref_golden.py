#!/usr/bin/env python
"""Regenerates tests/golden/ref_v1.npz — input/output vectors produced by the REFERENCE's own cost functors.

The functor headers /root/reference/src/lvio_fusion/include/lvio_fusion/ceres/{base,visual_error,lidar_error,pose_error}.hpp
are compiled UNMODIFIED into oracle/_ref/liblvf_ref.so (oracle/Makefile target `ref`, stand-in third-party headers in
oracle/ref_shim/) and evaluated through their own `X::Create(...)` + `CostFunction::Evaluate` on the seeded inputs below.
/root/reference exists only in the build container, so the outputs are committed here as fixtures: the oracle (CPU test,
bit-exact) and the HIP path (GPU test, 1e-6 relative) must both reproduce them.  Quaternions are deliberately NOT unit
length in half of the cases (the functors normalise inside QuaternionRotatePoint; the Jacobians carry the projector).
Run from the repo root, in the build container:  python tests/golden/make_ref_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lvio_fusion_amd import synthetic as syn   # noqa: E402  (input generators only: numpy)
from oracle import pyref as pr                 # noqa: E402
from oracle.pyoracle import Camera             # noqa: E402  (the ctypes struct only)

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_v1.npz")


def cam(c):
    return Camera.make(c["fx"], c["fy"], c["cx"], c["cy"], c["extrinsic"])


def cam_vec(c):
    return np.concatenate([[c["fx"], c["fy"], c["cx"], c["cy"]], c["extrinsic"]])


def inputs():
    """Everything the reference functors are evaluated on (pure numpy, seeded)."""
    g = {}
    cfg = syn.config4_window(n_kf=9, n_lm=120, n_prewindow=40, seed=777, imu_samples=4)
    g["cam0"], g["cam1"] = cam_vec(cfg["cam0"]), cam_vec(cfg["cam1"])
    poses = cfg["poses"].copy()
    poses[::2, :4] *= np.random.default_rng(11).uniform(0.5, 1.9, (poses[::2].shape[0], 1))     # every other pose: non-unit quaternion
    g["poses"] = poses
    for k in ("inv_depth", "w_kf"):
        g[k] = np.asarray(cfg[k], np.float64)
    for name in ("tc", "tf", "po"):
        for k, v in cfg[name].items():
            g[f"{name}_{k}"] = v
    rng = np.random.default_rng(12)
    n = 256
    p = rng.uniform(-25, 25, (n, 3)); p[:, 2] = rng.uniform(-2, 3, n)
    pa = p + rng.normal(0, 0.3, (n, 3)); pb = pa + rng.normal(0, 0.8, (n, 3)); pc = pa + rng.normal(0, 0.8, (n, 3))
    g["lidar_p"], g["lidar_pa"], g["lidar_pb"], g["lidar_pc"] = p, pa, pb, pc
    g["lidar_Twc1"] = np.array([0.013, -0.021, 0.31, 0.95, 12.0, -3.5, 0.4])        # not normalised on purpose
    g["lidar_rpyxyz"] = np.array([0.043, -0.017, 0.009, 0.31, -0.22, 0.07])
    g["lidar_Twc2"] = np.array([0.11, 0.19, -0.07, 0.93, 3.1, 4.2, 0.5])
    A = np.array([0.1, -0.2, 0.3, 0.9, 1.0, 2.0, 3.0]); B = np.array([-0.15, 0.1, 0.2, 0.95, 1.5, 2.2, 2.9])
    An, Bn = A.copy(), B.copy()
    An[:4] /= np.linalg.norm(An[:4]); Bn[:4] /= np.linalg.norm(Bn[:4])
    g["pose_A"], g["pose_B"], g["pose_An"], g["pose_Bn"] = A, B, An, Bn
    g["rel"] = np.array([0.02, -0.03, 0.05, 0.998, 1.1, -0.1, 0.03]); g["rel"][:4] /= np.linalg.norm(g["rel"][:4])
    g["q4"] = np.array([0.05, -0.02, 0.11, 0.97])
    return cfg, g


def main():
    assert pr.can_build(), "needs /root/reference (run in the build container)"
    pr.build(force=True)
    cfg, g = inputs()
    c0, c1 = cam(cfg["cam0"]), cam(cfg["cam1"])
    tc, tf, pol = cfg["tc"], cfg["tf"], cfg["po"]
    P, rho, w = g["poses"], g["inv_depth"], g["w_kf"]
    g["tc_r"], g["tc_J"] = pr.two_camera(tc["left_ob"], tc["right_ob"], tc["lm_idx"], tc["kf_idx"], rho, w, c0, c1)
    g["tf_r"], g["tf_Jd"], g["tf_J1"], g["tf_J2"] = pr.two_frame(tf["first_ob"], tf["ob"], tf["lm_idx"], tf["kf1_idx"], tf["kf2_idx"], rho, P, w, c0, c1)
    g["po_r"], g["po_J"] = pr.pose_only(pol["ob"], pol["kf_idx"], pol["pw_idx"], pol["pw"], P, w, c0)
    g["po_r_nojac"], _ = pr.pose_only(pol["ob"], pol["kf_idx"], pol["pw_idx"], pol["pw"], P, w, c0, jac=False)
    for mode in (0, 1):
        g[f"lidar_r{mode}"], g[f"lidar_J{mode}"] = pr.lidar_plane(mode, g["lidar_p"], g["lidar_pa"], g["lidar_pb"], g["lidar_pc"], g["lidar_Twc1"],
                                                                  g["lidar_rpyxyz"], 0.7)
    g["plane_r"], g["plane_J"] = pr.lidar_plane_se3(g["lidar_p"], g["lidar_pa"], g["lidar_pb"], g["lidar_pc"], g["lidar_Twc2"])
    g["pg_r"], g["pg_J1"], g["pg_J2"] = pr.pose_graph_rel(g["rel"], 100.0, 0.5, g["pose_A"], g["pose_B"])
    g["pg2_r"], g["pg2_J1"], g["pg2_J2"] = pr.pose_graph(g["pose_An"], g["pose_Bn"], 100.0, 1.0, g["pose_A"], g["pose_B"])
    g["pp_r"], g["pp_J"] = pr.pose_prior(g["pose_An"], 100.0, 0.3, g["pose_B"])
    g["re_r"], g["re_J"] = pr.r_error(g["pose_An"], 3.0, g["pose_B"])
    g["te_r"], g["te_J"] = pr.t_error(g["pose_An"][4:], 2.0, g["pose_B"])
    for mode in (0, 1):
        x3 = g["lidar_rpyxyz"][[1, 2, 5]] if mode == 0 else g["lidar_rpyxyz"][[0, 3, 4]]
        g[f"p3_r{mode}"], g[f"p3_J{mode}"] = pr.prior3(mode, g["lidar_rpyxyz"] * 1.1, 2.5, x3)
    g["rr_r"], g["rr_J"] = pr.relocate_r(g["pose_Bn"], g["pose_An"], g["q4"])
    g["h_rpyxyz"] = pr.se3_to_rpyxyz(g["pose_An"]); g["h_se3"] = pr.rpyxyz_to_se3(g["lidar_rpyxyz"])
    g["h_mul"] = pr.se3_mul(g["pose_A"], g["pose_B"]); g["h_inv"] = pr.se3_inv(g["pose_A"])
    pts = np.random.default_rng(13).uniform(-30, 30, (64, 3))
    g["h_pts"] = pts
    g["h_apply"] = np.stack([pr.se3_apply(g["pose_A"], q) for q in pts])
    g["h_apply_f32"] = np.stack([pr.se3_apply_f32(g["pose_An"].astype(np.float32), q.astype(np.float32)) for q in pts])   # the association's transform
    np.savez_compressed(OUT, **g)
    print("wrote", OUT, os.path.getsize(OUT), "bytes,", len(g), "arrays;", pr.lib().lvr_sources().decode())


if __name__ == "__main__":
    main()
