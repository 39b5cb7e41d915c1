"""bench.py's multi-rank branch (world > 1: the timing all-gather, rank 0's solo reference, the sharded relocalisation leg and its one
gather) executed before the driver does: two ranks under torch.distributed.run on THIS box's GPU with the gloo
backend (RCCL refuses two ranks on one device), CPU tensors in the collectives, everything else as the driver launches it.  No scaling
claim is read from it — the product backend is nccl = RCCL over xGMI, one GPU per rank (relocator.cpp:196-206 is the shard site)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


LINE_LIMIT = 6000        # the driver keeps the last 8 KB of stdout: the final line must fit with room to spare


def _run(extra, nproc=2, timeout=900, tmp=None):
    """-> (the compact stdout line, the full record from the sidecar file)"""
    legs_file = os.path.join(tmp or ROOT, f"bench_legs_test_{os.getpid()}.json")
    if nproc > 1:
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--steps", "20", "--warmup", "5", "--dist-backend", "gloo"]
    else:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5"]
    cmd += ["--legs-file", legs_file] + extra
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    assert p.returncode == 0 and len(lines) == 1, f"rc {p.returncode}; stdout tail: {p.stdout[-1500:]}; stderr tail: {p.stderr[-3000:]}"
    assert p.stdout.rstrip().splitlines()[-1] == lines[0], "the record must be the LAST stdout line"
    assert len(lines[0]) < LINE_LIMIT, f"bench line is {len(lines[0])} bytes"
    try:
        full = json.load(open(legs_file))
    finally:
        if os.path.exists(legs_file):
            os.remove(legs_file)
    return json.loads(lines[0]), full


@pytest.fixture(scope="module")
def one_rank_records():
    """the same eight candidates through the 1-rank batched path (what bench.py's relocalize leg runs on one GPU)"""
    from lvio_fusion_amd import api, relocalize as rl, synthetic as syn
    ctx = api.Context(0)
    cands = syn.config5_candidates(8)
    for c in cands:
        rl.split_candidate(c)
    best, rec = rl.relocalize(api, ctx, cands, batched=True)
    ctx.close()
    live = rec[rec[:, 8] >= 0]
    return best, live[np.argsort(live[:, 8])]


def test_two_ranks_gloo_line_and_records(one_rank_records):
    best1, rec1 = one_rank_records
    line, out = _run([])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling", "dtype", "config", "roofline", "ranks_seen"):
        assert k in line, k
    assert line["n_gpus"] == 2 and line["value"] == pytest.approx(out["value"], rel=1e-3)
    seen = out["ranks_seen"]
    assert seen["group_world_size"] == 2 and seen["backend"] == "gloo" and len(seen["device_uuids"]) == 2 and seen["table_rows"] == 8
    assert line["ranks_seen"]["table_crc32"] == seen["table_crc32"]
    assert out["n_gpus"] == 2 and out["steps"] == 20 and out["warmup"] == 5 and out["scaling"] == "weak" and out["dist_backend"] == "gloo"
    assert "gloo dry run" in out["dist_backend_note"]
    assert out["value"] > 0 and out["ms_per_step"] > 0 and out["steps_executed"] == 20
    assert len(out["per_rank_median_seconds_for_K_steps"]) == 2 and all(t > 0 for t in out["per_rank_median_seconds_for_K_steps"])
    assert out["single_gpu_seconds_same_work"] > 0 and out["scaling_efficiency_T1_over_TN"] > 0
    # value = the units all ranks processed / the slowest rank's time
    assert abs(out["value"] - 2 * out["steps_executed"] / (1e-3 * out["ms_per_step"] * out["steps_executed"])) <= 1e-6 * out["value"]
    r = out["legs"]["relocalize_8_candidates"]
    assert r["ranks"] == 2 and r["error"] is None and r["errors_by_rank"] == [None, None]
    assert r["candidates"] == list(range(8))
    assert np.array_equal(np.array(r["scores"]), rec1[:, 0])
    assert np.allclose(np.array(r["relative_o_c"]), rec1[:, 1:8], rtol=0, atol=1e-9)
    assert (r["best"] is None) == (best1 is None)
    if best1 is not None:
        assert r["best"]["candidate"] == best1[0] and r["best"]["score"] == best1[1]


def test_a_failing_rank_does_not_hang_the_collective(one_rank_records):
    """rank 1 raises inside its share: barriers and the gather still run, rank 0 reports its own four records and the error"""
    _, rec1 = one_rank_records
    _, out = _run(["--fail-rank", "1"])
    r = out["legs"]["relocalize_8_candidates"]
    assert out["n_gpus"] == 2 and r["errors_by_rank"][0] is None and "scripted failure" in r["errors_by_rank"][1]
    assert r["candidates"] == [0, 2, 4, 6]
    assert np.array_equal(np.array(r["scores"]), rec1[[0, 2, 4, 6], 0])


def test_one_rank_line_is_short_and_complete():
    """the line the driver parses: < 6 000 bytes, json round trip, the contract's keys + roofline + cpu_baseline (legs bounded to keep the test short)"""
    line, full = _run(["--legs", "icp,batched_windows_8"], nproc=1)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["steps"] == 20 and line["warmup"] == 5 and line["dtype"] == "f64" and "workload" in line["config"]
    assert line["value"] == pytest.approx(1e3 / line["ms_per_step"], rel=1e-3)
    r = line["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(r) and r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-3)
    c = line["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(c) and c["value"] > 0
    assert line["speedup_vs_cpu"]["combined_ba_plus_icp"] > 0 and line["speedup_vs_cpu"]["target"] == 10.0
    assert line["roofline_icp"]["frac"] > 0 and line["icp_mpairs_per_sec"] > 0
    assert all(line["verified"].values())
    assert json.loads(json.dumps(line)) == line
    assert full["value"] == pytest.approx(line["value"], rel=1e-3) and "legs" in full
