#!/usr/bin/env python
"""Regenerates tests/golden/ref_v2.npz — input/output vectors of the REFERENCE's own IMU text.

/root/reference/src/lvio_fusion/include/lvio_fusion/ceres/imu_error.hpp (ImuError::Evaluate :17-113),
imu/preintegration.h + src/preintegration.cpp (Preintegration::{Append, MidPointIntegration, Propagate, Repropagate, Evaluate})
and utility.h:99-140 (q_delta, skew_symmetric, q_left, q_right) are compiled UNMODIFIED into oracle/_ref/liblvf_ref.so
(oracle/Makefile target `ref`) against the fixed-size Matrix / Quaternion / LLT / inverse stand-in oracle/ref_shim/Eigen/Core, and
driven through `Preintegration::Create(bias)` + `Append(...)` and `ImuError::Create(pre)->Evaluate(...)` — the sequence
backend.cpp:150-152 and the front end drive.  /root/reference exists only in the build container, so the outputs are committed as
fixtures: the oracle (CPU test, bit for bit) and the HIP path (GPU test, 1e-6 relative) must both reproduce them.

Cases: ragged sample counts (0, 1, 4, 10, 100 per keyframe pair), re-propagation with new biases, unit and NON-unit pose
quaternions (Qi.inverse() divides by the squared norm, toRotationMatrix() does not normalise — both sides must follow that),
bias offsets from the linearisation point, and a covariance = identity variant whose sqrt_info is exactly I: its outputs are the
UNWEIGHTED residual and the pre-weighting 15 x 32 Jacobian of imu_error.hpp.
Run from the repo root, in the build container:  python tests/golden/make_ref_golden_imu.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lvio_fusion_amd import synthetic as syn   # noqa: E402  (input generators only: numpy)
from oracle import pyref as pr                 # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_v2.npz")
COUNTS = [10, 1, 0, 100, 4, 10, 7, 10, 3, 10]          # samples per keyframe pair (pair 2 has none, pair 3 is a 100 Hz gap)


def inputs(n_kf=11, seed=2024):
    """Pure numpy, seeded: a config-4 style window's IMU part with ragged sample counts."""
    cfg = syn.config4_window(n_kf=n_kf, n_lm=20, n_prewindow=4, seed=seed, imu_samples=10)
    rng = np.random.default_rng(seed + 1)
    g = {}
    n = len(cfg["imu"])
    assert n == len(COUNTS)
    samples, start = [], [0]
    for f, c in zip(cfg["imu"], COUNTS):
        s = np.concatenate([f["samples"]] * 10)[:c] if c else np.zeros((0, 7))
        if c:
            s = s.copy(); s[:, 1:] += rng.normal(0, 0.02, s[:, 1:].shape)           # every repetition differs
        samples.append(s); start.append(start[-1] + c)
    g["imu_samples"] = np.concatenate(samples); g["imu_start"] = np.asarray(start, np.int32)
    g["imu_acc0"] = np.stack([f["acc0"] for f in cfg["imu"]]); g["imu_gyr0"] = np.stack([f["gyr0"] for f in cfg["imu"]])
    g["imu_ba"] = np.stack([f["ba"] for f in cfg["imu"]]) + rng.normal(0, 0.02, (n, 3))
    g["imu_bg"] = np.stack([f["bg"] for f in cfg["imu"]]) + rng.normal(0, 0.002, (n, 3))
    g["imu_new_ba"] = g["imu_ba"] + rng.normal(0, 0.05, (n, 3)); g["imu_new_bg"] = g["imu_bg"] + rng.normal(0, 0.005, (n, 3))
    g["imu_kf_i"] = np.asarray([f["kf_i"] for f in cfg["imu"]], np.int32); g["imu_kf_j"] = np.asarray([f["kf_j"] for f in cfg["imu"]], np.int32)
    g["noise4"] = np.asarray(syn.IMU_NOISE, np.float64)
    g["poses"] = cfg["poses"].copy()
    g["poses_nonunit"] = cfg["poses"].copy()
    g["poses_nonunit"][:, :4] *= rng.uniform(0.8, 1.25, (n_kf, 1))
    g["vel"] = np.asarray(cfg["vel"], np.float64) + rng.normal(0, 0.05, (n_kf, 3))
    g["ba"] = np.asarray(cfg["ba"], np.float64) + rng.normal(0, 0.03, (n_kf, 3))       # away from the linearisation point: the bias-correction
    g["bg"] = np.asarray(cfg["bg"], np.float64) + rng.normal(0, 0.003, (n_kf, 3))      # terms dp_dba, dq_dbg ... are exercised
    return g


def split(g, f):
    return g["imu_samples"][g["imu_start"][f]:g["imu_start"][f + 1]]


def main():
    assert pr.can_build(), "needs /root/reference (run in the build container)"
    pr.build(force=True)
    g = inputs()
    n = len(g["imu_kf_i"]); nz = g["noise4"]
    g["pre"] = np.stack([pr.imu_preintegrate(split(g, f), g["imu_acc0"][f], g["imu_gyr0"][f], g["imu_ba"][f], g["imu_bg"][f], nz) for f in range(n)])
    g["pre_reprop"] = np.stack([pr.imu_repropagate(split(g, f), g["imu_acc0"][f], g["imu_gyr0"][f], g["imu_ba"][f], g["imu_bg"][f],
                                                   g["imu_new_ba"][f], g["imu_new_bg"][f], nz) for f in range(n)])
    # a pair without samples has a zero covariance and a single mid-point step a singular one (rank 12): covariance.inverse() is not
    # finite there and the reference's ImuError returns NaN.  Such pairs are pre-integration cases only.
    ok = np.asarray([c >= 3 for c in COUNTS])
    g["eval_pairs"] = np.flatnonzero(ok).astype(np.int32)
    pre, ki, kj = g["pre"][ok], g["imu_kf_i"][ok], g["imu_kf_j"][ok]
    for tag, P in (("unit", g["poses"]), ("nonunit", g["poses_nonunit"])):
        g[f"r_{tag}"], g[f"J_{tag}"] = pr.imu_eval(pre, ki, kj, P, g["vel"], g["ba"], g["bg"], nz)
        g[f"r_nojac_{tag}"], _ = pr.imu_eval(pre, ki, kj, P, g["vel"], g["ba"], g["bg"], nz, jac=False)
        g[f"raw_{tag}"] = pr.imu_raw_residual(pre, ki, kj, P, g["vel"], g["ba"], g["bg"], nz)           # Preintegration::Evaluate
        preI = pre.copy(); preI[:, 242:] = np.eye(15).ravel()                                           # covariance = I  =>  sqrt_info = I exactly
        g[f"rI_{tag}"], g[f"JI_{tag}"] = pr.imu_eval(preI, ki, kj, P, g["vel"], g["ba"], g["bg"], nz)   # unweighted residual, pre-weighting Jacobian
        assert np.array_equal(g[f"rI_{tag}"], g[f"raw_{tag}"])
        assert all(np.all(np.isfinite(g[k])) for k in (f"r_{tag}", f"J_{tag}", f"raw_{tag}", f"JI_{tag}"))
    np.savez_compressed(OUT, **g)
    print("wrote", OUT, os.path.getsize(OUT), "bytes,", len(g), "arrays;", pr.lib().lvr_sources().decode())


if __name__ == "__main__":
    main()
