#!/usr/bin/env python
"""Regenerates tests/golden/ref_v4.npz — the block lists of the REFERENCE's own Backend::BuildProblem.

/root/reference/src/lvio_fusion/src/backend.cpp (the whole file; BuildProblem is :96-183) and src/landmark.cpp are compiled UNMODIFIED into
oracle/_ref/liblvf_ref.so (oracle/Makefile target `ref`, driver oracle/ref_driver_backend.cpp) against the container stand-ins: ceres::Problem as
a recorder, the reference's own Frame / Feature / Landmark / Camera / adapt::Problem classes as the containers.  The driver builds the object
graph of every tick of tests/window_replay.py's scripted drive (12 keyframes through a 6-keyframe window, landmark ids not in creation order,
two removed observations, TwoFrame -> PoseOnly conversion when birth frames leave, with and without IMU), calls BuildProblem and reads the
recorded residual blocks back: functor kind, ProblemType (VisualError / WeakError: Camera::Far), landmark, keyframes, the weight and the
observations handed to X::Create, in insertion order.
/root/reference exists only in the build container, so the lists are committed as a fixture: lvf_window_*'s assembly (host walk and device
kernels) must reproduce them BIT FOR BIT (ids, order, weights, observations) on the GPU box — tests/test_gpu_window.py.
Run from the repo root, in the build container:  python tests/golden/make_ref_golden_backend.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lvio_fusion_amd import synthetic as syn   # noqa: E402  (input generator only: numpy)
from oracle import pyref as pr                 # noqa: E402
from tests import window_replay as wr          # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_v4.npz")


def reference_ticks(with_imu):
    """{(t, kind): dict(ids, vals, type, loss)} + per-tick (num_frames, num_parameter_blocks, first) from the live reference"""
    drive = wr.Drive(with_imu)
    out, meta = {}, {}
    for t in range(wr.N_KF):
        _, first = drive.tick(t)
        inp, live = drive.reference_input(t, first)
        r = pr.backend_build_problem(drive.cfg["cam0"], drive.cfg["cam1"], syn.baseline(), **inp)
        norm = wr.normalise_reference(r["rec_i"], r["rec_d"], live, drive.lm_id)
        for kind, v in norm.items():
            out[(t, kind)] = v
        meta[t] = (r["num_frames"], r["num_parameter_blocks"], first, len(r["rec_i"]))
    return out, meta


def main():
    store = {}
    for with_imu in (True, False):
        ticks, meta = reference_ticks(with_imu)
        tag = f"imu{int(with_imu)}"
        store[f"{tag}_meta"] = np.array([meta[t] for t in range(wr.N_KF)], np.int64)
        census = {k: 0 for k in wr.KINDS}
        weak = 0
        for (t, kind), v in ticks.items():
            store[f"{tag}_t{t}_{kind}_ids"] = v["ids"]; store[f"{tag}_t{t}_{kind}_vals"] = v["vals"]
            store[f"{tag}_t{t}_{kind}_type"] = v["type"].astype(np.int32); store[f"{tag}_t{t}_{kind}_loss"] = v["loss"].astype(np.int32)
            census[kind] += len(v["ids"])
            weak += int((v["type"] == pr.BP_TYPES.index("WeakError")).sum())
        print(tag, census, "WeakError blocks:", weak)
        assert census["PoseOnly"] > 0 and census["TwoFrame"] > 0 and weak > 0
        if not with_imu:
            assert census["PoseGraphError"] > 0 and census["PoseError"] > 0, "the drive must exercise the weak-constraint rule (backend.cpp:164-178)"
            n_kf_total = sum(int(store[f"{tag}_meta"][t][0]) for t in range(wr.N_KF))
            assert census["PoseGraphError"] + census["PoseError"] < n_kf_total, "... and keyframes with >= 20 near features too"
    np.savez_compressed(OUT, **store)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
