#!/usr/bin/env python
"""Regenerates tests/golden/ref_v3.npz — input/output vectors of the REFERENCE's own LiDAR front half.

/root/reference/src/lvio_fusion/src/projection.cpp (ImageProjection::{FindStartEndAngle, ProjectPointCloud, RemoveGround, Segment,
LabelComponents}) and src/association.cpp (FeatureAssociation::{Preprocess, AdjustDistortion, CalculateSmoothness, ExtractFeatures,
Sensor2Robot, AlignScan, ScanToMapWithGround, ScanToMapWithSegmented}) are compiled UNMODIFIED into oracle/_ref/liblvf_ref.so
(oracle/Makefile target `ref`, driver oracle/ref_driver_lidar.cpp) against container stand-ins: cv::Mat as a typed array,
pcl::PointCloud as a vector, the PCL filters as pass-throughs (so ExtractFeatures' own picks come out), KdTreeFLANN as the declared exact
brute-force search, ceres::Problem as a recorder whose blocks are evaluated through CostFunction::Evaluate.
/root/reference exists only in the build container, so the outputs are committed as fixtures: oracle/extract.h (libm form) must
reproduce them bit for bit, the association + factor restatement to 1e-12, on every box.

Cases: two raw 64-beam revolutions (1800 / 600 columns; NaN returns, out-of-range points, ground, walls, boxes), the second also through a
non-identity Sensor2Robot; AlignScan over two revolutions (covered / not covered); ground and surf scan-to-map problems with and without
the visual prior block.
Run from the repo root, in the build container:  python tests/golden/make_ref_golden_lidar.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lvio_fusion_amd import synthetic as syn   # noqa: E402  (input generators only: numpy)
from oracle import pyref as pr                 # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_v3.npz")
TAPS = ("filtered", "range_mat", "ground_mat", "label_mat", "segmented", "seg_ground", "seg_col", "seg_range", "start_ring", "end_ring", "curvature",
        "ground_raw", "surf_raw", "orientation")
SCANS = {"a": dict(seed=0x5CA9, n_az=1800, horizon_scan=1800), "b": dict(seed=0x5CB3, n_az=360, horizon_scan=360)}


def digest(a):
    import hashlib
    a = np.ascontiguousarray(a)
    return hashlib.sha256(a.tobytes()).hexdigest() + ":" + str(a.dtype) + ":" + "x".join(str(d) for d in a.shape)


def scan_inputs():
    return {k: syn.raw_scan(seed=v["seed"], n_az=v["n_az"]) for k, v in SCANS.items()}


def icp_inputs():
    c = syn.config5_candidates(1, seed=4242, n_query=3000, n_az=240, overlap="full")[0]
    g = dict(map=c["map"], map_ground=c["map_ground"], query=c["query"], query_ground=c["query_ground"], map_pose=c["map_pose"], frame_pose=c["init_pose"])
    return g


def main():
    out = {}
    # (the inputs are regenerated from their seeds by the tests; their digests are stored so a drifting generator is noticed.  The full-size
    # scan "a" is stored as digests of every tap — 5 MB of arrays otherwise — the smaller scan "b" in full)
    pts_all = scan_inputs()
    for k, pts in pts_all.items():
        out[f"scan_{k}_points_sha256"] = np.array(digest(pts))
        r = pr.lidar_extract(pts, horizon_scan=SCANS[k]["horizon_scan"])
        for t in TAPS:
            if k == "a":
                out[f"scan_{k}_{t}_sha256"] = np.array(digest(r[t]))
            else:
                out[f"scan_{k}_{t}"] = r[t]
        out[f"scan_{k}_counts"] = np.array([r["n_filtered"], r["n_segmented"], len(r["ground_raw"]), len(r["surf_raw"]), r["label_count"]])
    ext = syn.lidar_extrinsic()
    r = pr.lidar_extract(pts_all["b"], extrinsic=ext, horizon_scan=SCANS["b"]["horizon_scan"])
    out["scan_b_extrinsic"] = ext; out["scan_b_ground_robot"] = r["ground_raw"]; out["scan_b_surf_robot"] = r["surf_raw"]
    # AlignScan
    rng = np.random.default_rng(77)
    pc1 = np.zeros((1000, 4), np.float32); pc1[:, :3] = rng.normal(0, 10, (1000, 3))
    pc2 = np.zeros((1300, 4), np.float32); pc2[:, :3] = rng.normal(0, 10, (1300, 3))
    out["align_pc1"], out["align_pc2"] = pc1, pc2
    out["align_args"] = np.array([10.0, 10.1036, 0.1036])
    # (a time at or past the LAST stamp makes the reference dereference raw_point_clouds_.end() — association.cpp:41-45: undefined behaviour, not a case)
    for name, t in (("mid", 10.06), ("mid2", 10.0518), ("early", 10.0), ("uncovered", 9.9)):
        ok, cl = pr.align_scan(pc1, 10.0, pc2, 10.1036, 0.1036, t)
        out[f"align_{name}_time"] = np.array([t]); out[f"align_{name}_ok"] = np.array([int(ok)]); out[f"align_{name}_cloud"] = cl
    # scan-to-map problems
    g = icp_inputs()
    for key, v in g.items():
        out[f"icp_{key}"] = np.asarray(v)
    from oracle import pyoracle as po       # (host SE3 helpers only: the inputs' para is se32rpyxyz(map_pose^-1 * frame_pose))
    para = po.se3_to_rpyxyz(po.se3_mul(po.se3_inv(g["map_pose"]), g["frame_pose"]))
    out["icp_para"] = para
    mg, ms = g["map"][g["map_ground"]], g["map"][~g["map_ground"]]
    qg, qs = g["query"][g["query_ground"]], g["query"][~g["query_ground"]]
    for mode, (q, m) in enumerate(((qg, mg), (qs, ms))):
        for relocate in (1, 0):
            r = pr.scan_to_map(mode, q, m, g["frame_pose"], g["map_pose"], para, n_features_left=137, relocate=bool(relocate))
            tag = f"icp_m{mode}_r{relocate}"
            out[tag + "_residuals"] = r["residuals"]; out[tag + "_jacobians"] = r["jacobians"]
            out[tag + "_meta"] = np.array([r["huber_a"], r["n_lidar"], r["n_other"], r["n_param_blocks"], r["n_lidar_type"]]); out[tag + "_prior"] = r["prior"]
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", {k: v.shape for k, v in out.items() if k.endswith("ground_raw") or k.endswith("_residuals")})


if __name__ == "__main__":
    main()
