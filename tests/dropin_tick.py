"""What the LITERAL drop-in costs per tick on the BASELINE window (50 keyframes, 10 000 landmarks, ~82 k residual blocks): the reference's own
Backend::BuildProblem (src/backend.cpp:96-183, compiled unmodified into oracle/_ref/liblvf_dropin.so with include/reference_patch ahead of the reference's
headers) walking its pointer graph — std::map<time, Frame>, std::map<id, Feature>, weak_ptr locks, one heap cost function per block — followed by
adapt::Solve = gpu::Solve with max_num_iterations = 1, against lvf_window_solve on the same window.  Test infrastructure (it loads oracle/_ref): a tool,
not a bench.py leg (it lives under tests/ because it loads oracle/_ref).      python tests/dropin_tick.py [reps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lvio_fusion_amd import api, synthetic as syn      # noqa: E402
from oracle import pydropin                            # noqa: E402


def baseline_window_inputs():
    """the BASELINE window (50 keyframes, 10 000 landmarks, no pre-window landmarks) as the flat arrays oracle/pydropin.backend_solve takes:
    -> (cfg with float-rounded observations / weights and unit extrinsic quaternions as the reference holds them, cams, kwargs)"""
    cfg = syn.config4_window(n_prewindow=0)
    n_kf, n_lm = cfg["n_kf"], cfg["n_lm"]
    tc, tf = cfg["tc"], cfg["tf"]
    f32 = lambda x: np.asarray(x, np.float32).astype(np.float64)
    birth = np.full(n_lm, -1, np.int32); birth[tc["lm_idx"]] = tc["kf_idx"]
    right = np.zeros((n_lm, 2)); right[tc["lm_idx"]] = f32(tc["right_ob"])
    obs_lm = np.concatenate([tc["lm_idx"], tf["lm_idx"]]); obs_fr = np.concatenate([tc["kf_idx"], tf["kf2_idx"]]); obs_xy = np.concatenate([f32(tc["left_ob"]), f32(tf["ob"])])
    cams = {}
    for cam in ("cam0", "cam1"):
        c = dict(cfg[cam]); e = np.array(c["extrinsic"], np.float64); e[:4] /= np.linalg.norm(e[:4]); c["extrinsic"] = e; cams[cam] = c
    args = dict(time=10.0 + 0.5 * np.arange(n_kf), pose=cfg["poses"], w_visual=f32(cfg["w_kf"]), good_imu=np.ones(n_kf, np.uint8), first_active=0, imu_initialized=True,
                lm_id=5000 + np.arange(n_lm), lm_birth=birth, lm_inv_depth=cfg["inv_depth"], lm_right_ob=right, obs_lm=obs_lm, obs_frame=obs_fr, obs_xy=obs_xy,
                vel=cfg["vel"], ba=cfg["ba"], bg=cfg["bg"], imu=[None] + cfg["imu"], imu_noise=syn.IMU_NOISE)
    return cfg, cams, args


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    cfg, cams, args = baseline_window_inputs()
    pydropin.lib()
    ts, parts = [], []
    for _ in range(reps + 1):
        t0 = time.perf_counter()
        out = pydropin.backend_solve(cams["cam0"], cams["cam1"], syn.baseline(), max_num_iterations=1, **args)
        ts.append(time.perf_counter() - t0); parts.append(out["times_ms"])
    assert out["rc"] == 0, out["message"]
    print({"blocks": out["num_residual_blocks"], "recorded": out["recorded"], "cost": [out["initial_cost"], out["final_cost"]],
           "dropin_tick_ms_incl_graph_build_median": 1e3 * float(np.median(ts[1:])), "all_ms": [round(1e3 * t, 2) for t in ts],
           "parts_ms_median": {k: round(float(np.median([p[k] for p in parts[1:]])), 3) for k in parts[0]},
           "reference_tick_ms": round(float(np.median([p["build_problem"] + p["adapt_solve"] + p["destroy_and_read_back"] for p in parts[1:]])), 3),
           "note": "python -> ctypes -> the driver builds the reference's object graph (frames, landmarks, features: heap objects, std::map inserts) -> Backend::BuildProblem -> "
                   "adapt::Solve (1 LM iteration on the MI355X) -> read back; LVF_ADAPTER_TIMING=1 prints gpu::Solve's own split"})


if __name__ == "__main__":
    main()
