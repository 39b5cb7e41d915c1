"""Two host threads, each with its own context (= its own HIP stream and device allocator), in flight at once — the situation of
Backend::Optimize and Relocator -> Mapping::Relocate (src/lvio_fusion/src/relocator.cpp:188 takes the backend lock only at :210).
Results must equal the single-threaded ones; error strings are thread-local."""
import threading

import numpy as np
import pytest

from lvio_fusion_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def ba_job(api, oracle_pre, cfg, out, key, reps):
    ctx = api.Context(0)
    try:
        for _ in range(reps):
            st = api.State(ctx, cfg["n_kf"], cfg["n_lm"])
            for f, k in ((api.POSES, "poses"), (api.VEL, "vel"), (api.BA, "ba"), (api.BG, "bg"), (api.INV_DEPTH, "inv_depth"), (api.W_VISUAL, "w_kf")):
                st.set(f, cfg[k])
            tc, tf, po = cfg["tc"], cfg["tf"], cfg["po"]
            hs = [api.two_camera_batch(ctx, cfg["cam0"], cfg["cam1"], tc["left_ob"], tc["right_ob"], tc["lm_idx"], tc["kf_idx"]),
                  api.two_frame_batch(ctx, cfg["cam0"], cfg["cam1"], tf["first_ob"], tf["ob"], tf["lm_idx"], tf["kf1_idx"], tf["kf2_idx"]),
                  api.pose_only_batch(ctx, cfg["cam0"], po["ob"], po["kf_idx"], po["pw_idx"], po["pw"]),
                  api.imu_batch(ctx, oracle_pre, [f["kf_i"] for f in cfg["imu"]], [f["kf_j"] for f in cfg["imu"]])]
            prob = api.Problem(ctx, st, *hs)
            opt = api.default_solver_options(); opt.max_num_iterations = 5
            s = prob.solve(opt)
            out[key] = (s.final_cost, st.get(api.POSES).copy())
            prob.close()
            for h in hs:
                h.close()
            st.close()
    except Exception as e:      # surfaced by the main thread
        out[key] = e
    finally:
        ctx.close()


def icp_job(api, cand, out, key, reps):
    from lvio_fusion_amd import relocalize as rl
    ctx = api.Context(0)
    try:
        for _ in range(reps):
            res = rl.evaluate_candidate(api, ctx, cand)
            out[key] = (res.score, np.array(res.pose[:]))
    except Exception as e:
        out[key] = e
    finally:
        ctx.close()


def test_concurrent_contexts_match_serial(oracle):
    from lvio_fusion_amd import api
    cfg = syn.config4_window(n_kf=10, n_lm=300, n_prewindow=40, seed=3, imu_samples=4)
    pre = np.stack([oracle.imu_preintegrate(f["samples"], f["acc0"], f["gyr0"], f["ba"], f["bg"], syn.IMU_NOISE) for f in cfg["imu"]])
    cand = syn.config5_candidates(1, seed=8, n_query=6000, n_az=300, overlap="full")[0]
    serial = {}
    ba_job(api, pre, cfg, serial, "ba", 1); icp_job(api, cand, serial, "icp", 1)
    assert not isinstance(serial["ba"], Exception) and not isinstance(serial["icp"], Exception), serial
    par = {}
    ts = [threading.Thread(target=ba_job, args=(api, pre, cfg, par, "ba", 6)), threading.Thread(target=icp_job, args=(api, cand, par, "icp", 6)),
          threading.Thread(target=ba_job, args=(api, pre, cfg, par, "ba2", 6))]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
        assert not t.is_alive()
    for k in ("ba", "icp", "ba2"):
        assert not isinstance(par[k], Exception), par[k]
    for k in ("ba", "ba2"):
        assert abs(par[k][0] - serial["ba"][0]) <= 1e-6 * serial["ba"][0]
        assert np.allclose(par[k][1], serial["ba"][1], rtol=1e-6, atol=1e-9)
    assert par["icp"][0] == serial["icp"][0] and np.allclose(par["icp"][1], serial["icp"][1], rtol=1e-6, atol=1e-9)


def test_error_strings_are_thread_local():
    from lvio_fusion_amd import api, _lib
    ctx = api.Context(0)
    seen = {}

    def bad():
        c2 = api.Context(0)
        try:
            api.State(c2, -1, 0)
        except api.LvfError as e:
            seen["t"] = str(e)
        c2.close()
    with pytest.raises(api.LvfError) as e1:
        api.Map(ctx, np.zeros((4, 4), np.float32), -1.0)
    t = threading.Thread(target=bad); t.start(); t.join()
    assert "negative size" in seen["t"]
    assert "max_radius2" in str(e1.value)
    assert b"max_radius2" in _lib.lib().lvf_last_error()        # this thread's message was not overwritten by the other thread
    ctx.close()
