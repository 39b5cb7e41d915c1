"""Shared helpers for the GPU parity tests."""
import numpy as np

RTOL = 1e-6          # north_star: residuals/Jacobians within 1e-6 relative, fp64
ATOL_SCALE = 1e-10   # absolute floor relative to the largest magnitude of the compared array (cancellation zeros)


def assert_parity(gpu, ref, what=""):
    gpu = np.asarray(gpu, dtype=np.float64); ref = np.asarray(ref, dtype=np.float64)
    assert gpu.shape == ref.shape, f"{what}: shape {gpu.shape} vs {ref.shape}"
    assert np.all(np.isfinite(gpu)), f"{what}: non-finite GPU output"
    scale = float(np.max(np.abs(ref))) if ref.size else 0.0
    err = np.abs(gpu - ref)
    tol = RTOL * np.abs(ref) + ATOL_SCALE * scale
    bad = err > tol
    if np.any(bad):
        k = np.unravel_index(np.argmax(err - tol), err.shape)
        raise AssertionError(f"{what}: {bad.sum()} / {bad.size} elements beyond rtol {RTOL}; worst at {k}: gpu={gpu[k]!r} ref={ref[k]!r}")
    return float(np.max(err / (np.abs(ref) + ATOL_SCALE * scale + 1e-300))) if ref.size else 0.0


def ocam(oracle, c):
    return oracle.Camera.make(c["fx"], c["fy"], c["cx"], c["cy"], c["extrinsic"])
