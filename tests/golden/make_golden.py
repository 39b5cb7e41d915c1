#!/usr/bin/env python
"""Regenerates tests/golden/hotpath_v1.npz — small seeded input/output vectors for every functor on the path.

PARITY UNPINNED: the reference ships no tests or golden vectors and cannot be built here (no Ceres/Eigen/Sophus/PCL), so
these vectors are produced by the ORACLE (oracle/, a restatement of the reference functors that is itself cross-checked
by mpmath / closed-form / scipy tests).  Their job is drift detection: both the oracle (CPU test) and the HIP path (GPU
test) must keep reproducing them.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lvio_fusion_amd import synthetic as syn   # noqa: E402  (input generators only: numpy)
from oracle import pyoracle as po              # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hotpath_v1.npz")


def cam(c):
    return po.Camera.make(c["fx"], c["fy"], c["cx"], c["cy"], c["extrinsic"])


def cam_vec(c):
    return np.concatenate([[c["fx"], c["fy"], c["cx"], c["cy"]], c["extrinsic"]])


def main():
    po.build()
    g = {}
    cfg = syn.config4_window(n_kf=6, n_lm=40, n_prewindow=12, seed=2024, imu_samples=4)
    c0, c1 = cam(cfg["cam0"]), cam(cfg["cam1"])
    g["cam0"], g["cam1"] = cam_vec(cfg["cam0"]), cam_vec(cfg["cam1"])
    for k in ("poses", "vel", "ba", "bg", "inv_depth", "w_kf"):
        g[k] = np.asarray(cfg[k], np.float64)
    tc, tf, pol = cfg["tc"], cfg["tf"], cfg["po"]
    for name, d in (("tc", tc), ("tf", tf), ("po", pol)):
        for k, v in d.items():
            g[f"{name}_{k}"] = v
    g["tc_r"], g["tc_J"] = po.two_camera(tc["left_ob"], tc["right_ob"], tc["lm_idx"], tc["kf_idx"], cfg["inv_depth"], cfg["w_kf"], c0, c1)
    g["tf_r"], g["tf_Jd"], g["tf_J1"], g["tf_J2"] = po.two_frame(tf["first_ob"], tf["ob"], tf["lm_idx"], tf["kf1_idx"], tf["kf2_idx"], cfg["inv_depth"],
                                                                 cfg["poses"], cfg["w_kf"], c0, c1)
    g["po_r"], g["po_J"] = po.pose_only(pol["ob"], pol["kf_idx"], pol["pw_idx"], pol["pw"], cfg["poses"], cfg["w_kf"], c0)
    # IMU: samples -> pre-integration -> ImuError
    imu = cfg["imu"]
    g["imu_samples"] = np.stack([f["samples"] for f in imu]); g["imu_acc0"] = np.stack([f["acc0"] for f in imu]); g["imu_gyr0"] = np.stack([f["gyr0"] for f in imu])
    g["imu_ba"] = np.stack([f["ba"] for f in imu]); g["imu_bg"] = np.stack([f["bg"] for f in imu])
    g["imu_kf_i"] = np.array([f["kf_i"] for f in imu], np.int32); g["imu_kf_j"] = np.array([f["kf_j"] for f in imu], np.int32)
    pre = np.stack([po.imu_preintegrate(f["samples"], f["acc0"], f["gyr0"], f["ba"], f["bg"], syn.IMU_NOISE) for f in imu])
    g["imu_pre"] = pre
    g["imu_r"], g["imu_J"] = po.imu_eval(pre, g["imu_kf_i"], g["imu_kf_j"], cfg["poses"], cfg["vel"], cfg["ba"], cfg["bg"])
    # pose priors
    rng = np.random.default_rng(5)
    last = cfg["poses"][1].copy(); last[4:] += rng.normal(0, 0.05, 3)
    g["prior_kf_a"] = np.array([-1, 1, 3], np.int32); g["prior_kf_b"] = np.array([0, 2, 4], np.int32)
    tgt = np.zeros((3, 7))
    tgt[0] = cfg["poses_true"][0]
    tgt[1, :6] = po.pose_graph_target(last, cfg["poses"][2]); tgt[2, :6] = po.pose_graph_target(cfg["poses_true"][3], cfg["poses_true"][4])
    g["prior_target"], g["prior_weight"], g["prior_v"] = tgt, np.array([100.0, 100.0, 7.0]), np.array([0.0, 1.0, 0.5])
    pr, pja, pjb = [], [], []
    for i in range(3):
        if g["prior_kf_a"][i] < 0:
            r, J = po.pose_prior(tgt[i], g["prior_weight"][i], g["prior_v"][i], cfg["poses"][g["prior_kf_b"][i]])
            pr.append(r); pja.append(np.zeros((6, 7))); pjb.append(J)
        else:
            r, J1, J2 = po.pose_graph(tgt[i, :6], g["prior_weight"][i], g["prior_v"][i], cfg["poses"][g["prior_kf_a"][i]], cfg["poses"][g["prior_kf_b"][i]])
            pr.append(r); pja.append(J1); pjb.append(J2)
    g["prior_r"], g["prior_Ja"], g["prior_Jb"] = np.array(pr), np.array(pja), np.array(pjb)
    # one LM iteration of the whole window (with the priors)
    win = po.Window(cfg, pre, priors=dict(kf_a=g["prior_kf_a"], kf_b=g["prior_kf_b"], target=tgt, weight=g["prior_weight"], v=g["prior_v"]))
    it = win.lm_iteration(1e4, 2.0)
    g["lm_cost_before"], g["lm_cost_after"], g["lm_radius"] = np.array(it["cost_before"]), np.array(it["cost_after"]), np.array(it["radius"])
    g["lm_accepted"] = np.array(int(it["accepted"]))
    g["lm_poses"], g["lm_inv_depth"], g["lm_vel"] = win.poses.copy(), win.inv_depth.copy(), win.vel.copy()
    # lidar: association + plane factors + one sub-problem solve
    c3 = syn.config3_icp(seed=99, n_query=1500, n_az=160)
    qg, mg = c3["query"][c3["query_ground"]][:400], c3["map"][c3["map_ground"]]
    g["knn_map"], g["knn_query"], g["knn_pose"], g["knn_thr"] = mg, qg, c3["pose0"], np.array(c3["thr_ground"], np.float32)
    idx, d2, valid = po.knn3(mg, qg, c3["pose0"], c3["thr_ground"])
    g["knn_idx"], g["knn_d2"], g["knn_valid"] = idx, d2, valid
    v = valid > 0
    p = qg[v, :3].astype(np.float64); pa, pb, pc = (mg[idx[v, k], :3].astype(np.float64) for k in range(3))
    g["lidar_p"], g["lidar_pa"], g["lidar_pb"], g["lidar_pc"], g["lidar_Twc1"] = p, pa, pb, pc, c3["map_pose"]
    rp = po.se3_to_rpyxyz(po.se3_mul(po.se3_inv(c3["map_pose"]), c3["pose0"]))
    g["lidar_rpyxyz"] = rp
    nrm = po.plane_normals(pa, pb, pc)
    g["lidar_nrm"] = nrm
    for mode in (0, 1):
        r, J = po.lidar_plane(mode, p, pa, nrm, c3["map_pose"], rp, 0.7)
        g[f"lidar_r{mode}"], g[f"lidar_J{mode}"] = r, J
    x, summ = po.icp_solve(mg, qg, c3["map_pose"], c3["pose0"], rp, 0, c3["thr_ground"], syn.W_LIDAR_GROUND, 0.0, prior_w=50.0)
    g["icp_x"], g["icp_costs"] = x, np.array([summ["initial_cost"], summ["final_cost"]])
    g["icp_counts"] = np.array([summ["num_residual_blocks"], summ["num_iterations"], summ["num_successful_steps"]], np.int32)
    np.savez_compressed(OUT, **g)
    print("wrote", OUT, os.path.getsize(OUT), "bytes,", len(g), "arrays")


if __name__ == "__main__":
    main()
