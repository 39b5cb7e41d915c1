"""The N>1 path on CPU: world_size-2 and world_size-4 gloo processes exercise the sharding, the single all_gather and the arg-max of the
loop-closure candidate evaluation (lvio_fusion_amd/relocalize.py) with scripted per-candidate results — the collective
logic is device-independent; the per-candidate solve itself is covered on the GPU (tests/test_gpu_relocalize.py)."""
import os
import socket
import subprocess
import sys

import numpy as np

from lvio_fusion_amd import relocalize as rl

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ownership_covers_every_candidate_once():
    for n in (1, 5, 8, 13):
        for world in (1, 2, 4, 8):
            got = sorted(c for r in range(world) for c in rl.owned(n, r, world))
            assert got == list(range(n))
            assert all(len(rl.owned(n, r, world)) <= rl.slots(n, world) for r in range(world))


def test_choose_best_follows_relocator_rules():
    recs = rl.empty_records(5)
    ident = [0, 0, 0, 1, 0, 0, 0]
    recs[0] = rl.make_record(0, 25.9, ident)     # int(25.9) - 20 = 5
    recs[1] = rl.make_record(1, 20.0, ident)     # 0: not > 0 -> Relocate() returned false
    recs[2] = rl.make_record(2, 31.0, ident)     # 11
    recs[3] = rl.make_record(3, 31.7, ident)     # 11: ties go to the LATER candidate (>=)
    best = rl.choose_best(recs)
    assert best[0] == 3 and best[1] == 11.0
    assert rl.choose_best(rl.empty_records(3)) is None
    only_bad = rl.empty_records(2); only_bad[0] = rl.make_record(0, 12.0, ident)
    assert rl.choose_best(only_bad) is None


WORKER = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from lvio_fusion_amd import relocalize as rl
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
n = 7
scores = [22.0, 45.5, 19.0, 45.2, 30.0, 21.0, 44.9]          # scripted Mapping::Relocate results
table = rl.empty_records(rl.slots(n, world))
for s, cid in enumerate(rl.owned(n, rank, world)):
    rel = np.array([0, 0, np.sin(0.01 * cid), np.cos(0.01 * cid), cid, 2.0 * cid, 0.5])
    table[s] = rl.make_record(cid, scores[cid], rel)
allrec = rl.gather_records(table, world)
best = rl.choose_best(allrec)
print(json.dumps({"rank": rank, "best": [best[0], best[1]] + best[2].tolist(), "n_rec": int((allrec[:, 8] >= 0).sum())}))
dist.destroy_process_group()
'''


def test_gloo_world2_gather_and_argmax(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=180)
        assert p.returncode == 0, e
        outs.append(__import__("json").loads(o.strip().splitlines()[-1]))
    # both ranks see all 7 records and agree on the winner: candidates 1 and 3 tie at int(45.x) - 20 = 25 -> the later one
    assert all(o["n_rec"] == 7 for o in outs)
    assert outs[0]["best"] == outs[1]["best"]
    assert outs[0]["best"][0] == 3 and outs[0]["best"][1] == 25.0
    assert np.allclose(outs[0]["best"][2:], [0, 0, np.sin(0.03), np.cos(0.03), 3, 6.0, 0.5])


WORKER4 = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from lvio_fusion_amd import relocalize as rl
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
fail_rank, n = int(os.environ["FAIL_RANK"]), int(os.environ["N_CAND"])
dist.init_process_group("gloo", rank=rank, world_size=world)
scores = [22.0 + 3.0 * ((7 * c) % 5) for c in range(n)]
table = rl.empty_records(rl.slots(n, world))
err = None
try:      # the per-GPU part may fail on one rank: the collective below must still be entered by everyone (bench.py's structure)
    for s, cid in enumerate(rl.owned(n, rank, world)):
        if rank == fail_rank and s == 1:
            raise RuntimeError("device lost mid-evaluation")
        table[s] = rl.make_record(cid, scores[cid], np.array([0, 0, 0, 1.0, cid, 0, 0]))
except Exception as e:
    err = repr(e)
dist.barrier()
allrec = rl.gather_records(table, world)
dist.barrier()
best = rl.choose_best(allrec)
live = sorted(int(x) for x in allrec[allrec[:, 8] >= 0][:, 8])
print(json.dumps({"rank": rank, "err": err, "live": live, "best": None if best is None else [best[0], best[1]]}))
dist.destroy_process_group()
'''


def _run_world(tmp_path, world, n, fail_rank):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "worker4.py"
    script.write_text(WORKER4)
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FAIL_RANK=str(fail_rank), N_CAND=str(n))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=240)
        assert p.returncode == 0, e
        outs.append(__import__("json").loads(o.strip().splitlines()[-1]))
    return outs


def test_gloo_world4_uneven_candidate_counts(tmp_path):
    """7 candidates over 4 ranks: shares of 2, 2, 2, 1; the short rank's unused slot must not disturb the arg-max."""
    outs = _run_world(tmp_path, 4, 7, -1)
    assert all(o["live"] == list(range(7)) and o["err"] is None for o in outs)
    scores = [int(22.0 + 3.0 * ((7 * c) % 5)) - 20 for c in range(7)]
    top = max(scores)
    want = max(c for c in range(7) if scores[c] == top)          # `>=`: the later candidate wins a tie
    assert all(o["best"] == [want, float(top)] for o in outs)


def test_gloo_world4_a_rank_failing_mid_evaluation_does_not_hang(tmp_path):
    """Rank 2 throws after its first candidate: every rank still reaches the barriers and the all_gather, the failed rank's
    remaining candidate simply has no record, and all ranks agree on the winner among the rest."""
    outs = _run_world(tmp_path, 4, 8, 2)
    assert outs[2]["err"] is not None and all(o["err"] is None for i, o in enumerate(outs) if i != 2)
    lost = rl.owned(8, 2, 4)[1]
    assert all(o["live"] == [c for c in range(8) if c != lost] for o in outs)
    assert len({tuple(o["best"]) for o in outs}) == 1


def test_worker_contexts_fill_the_same_slots_as_one_stream(monkeypatch):
    """relocalize(..., workers=): a rank's candidates are dealt round-robin to its contexts and evaluated by one host thread each; the
    records must land in the slots the one-stream loop fills (host logic only: the evaluation is scripted)."""
    import threading
    from lvio_fusion_amd import relocalize as rl

    class Res:
        def __init__(self, cid):
            self.score = 20 + (cid * 7) % 13
            self.relative_o_c = [0.0, 0.0, 0.0, 1.0, float(cid), 0.5 * cid, -1.0]

    seen = {}

    def fake_eval(api, ctx, cand, resolution=0.2):
        seen.setdefault(ctx, []).append((cand["id"], threading.get_ident()))
        return Res(cand["id"])
    monkeypatch.setattr(rl, "evaluate_candidate", fake_eval)
    cands = [{"id": i} for i in range(11)]
    for world in (1, 2):
        for rank in range(world):
            one = rl.empty_records(rl.slots(len(cands), world))
            for s, cid in enumerate(rl.owned(len(cands), rank, world)):
                one[s] = rl.make_record(cid, Res(cid).score, np.array(Res(cid).relative_o_c))
            seen.clear()
            monkeypatch.setattr(rl, "gather_records", lambda local, w, device=None: local.copy())
            best, many = rl.relocalize(None, "ctx0", cands, rank=rank, world=world, workers=["ctx1", "ctx2"])
            assert np.array_equal(many, one)
            assert set(seen) == {"ctx0", "ctx1", "ctx2"}                                   # every context got work ...
            assert all(len({t for _, t in v}) == 1 for v in seen.values())                # ... from one thread each
            assert sorted(c for v in seen.values() for c, _ in v) == rl.owned(len(cands), rank, world)
