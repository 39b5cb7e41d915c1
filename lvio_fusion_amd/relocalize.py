"""Loop-closure candidate evaluation sharded one candidate per GPU (BASELINE configs[4]).

Reference flow (src/lvio_fusion/src/relocator.cpp:188-206): for every keyframe of the new sub-map, Relocator::Relocate ->
Mapping::Relocate (src/mapping.cpp:251-300: 4 outer iterations x {ground, surf} scan-to-map solves, no shared mutable
state between candidates) sets loop_closure->score = score - 20 and relative_o_c; the candidate with the largest score
(`>=`, so the LAST of equal scores; only scores > 0 qualify) becomes best_frame.

Here: rank r evaluates candidates r, r + world, ... on its own GPU (lvf_scan_match, everything device-resident), the ranks
exchange ONE fixed-size record per candidate slot — (loop score, relative_o_c[7], candidate id) — with a single
all_gather (RCCL over xGMI on GPUs: pure latency; gloo in the CPU tests), and every rank runs the same arg-max.
No other collective touches the data path.
"""
import numpy as np

RELOCATE_BASE_SCORE = 20          # relocator.cpp:181  loop_closure->score += score - 20
RECORD = 9                        # score, relative_o_c[7], candidate id


def owned(n_candidates, rank, world):
    return list(range(rank, n_candidates, world))


def slots(n_candidates, world):
    return (n_candidates + world - 1) // world


def empty_records(n_slots):
    r = np.zeros((n_slots, RECORD))
    r[:, 0] = -np.inf
    r[:, 4] = 1.0                 # identity quaternion (x,y,z,w)
    r[:, 8] = -1.0                # unused slot
    return r


def make_record(cand_id, mapping_score, relative_o_c):
    rec = np.empty(RECORD)
    rec[0] = float(int(mapping_score) - RELOCATE_BASE_SCORE)
    rec[1:8] = relative_o_c
    rec[8] = float(cand_id)
    return rec


def choose_best(records):
    """records: (n, 9).  Returns (candidate id, score, relative_o_c) or None — relocator.cpp:191-204 in candidate order."""
    recs = [r for r in np.asarray(records).reshape(-1, RECORD) if r[8] >= 0]
    recs.sort(key=lambda r: r[8])
    best, max_score = None, -1.0
    for r in recs:
        if r[0] > 0 and r[0] >= max_score:          # Relocate() returned true, and `>=` keeps the later candidate on ties
            max_score, best = r[0], r
    if best is None:
        return None
    return int(best[8]), float(best[0]), best[1:8].copy()


def gather_records(local, world, device=None):
    """One all_gather of the (slots, 9) float64 table.  world == 1: no communication."""
    if world == 1:
        return local.copy()
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(np.ascontiguousarray(local))
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return np.concatenate([o.cpu().numpy() for o in out])


def evaluate_candidate(api, ctx, cand, resolution=0.2):
    """cand: dict(map, map_ground(mask), query, query_ground(mask), map_pose, last_pose, init_pose).  Mapping::Relocate."""
    mg, ms, qg, qs = split_candidate(cand)
    opt = api.scan_match_options(resolution, outer_iterations=4, prior_weight=0.0)
    h = []
    try:
        mpg = api.Map(ctx, mg, opt.thr_ground) if len(mg) else None
        scg = api.Scan(ctx, qg) if mpg is not None else None
        mps = api.Map(ctx, ms, opt.thr_surf) if len(ms) else None
        scs = api.Scan(ctx, qs) if mps is not None else None
        h = [x for x in (mpg, scg, mps, scs) if x is not None]
        return api.scan_match(mpg, scg, mps, scs, cand["map_pose"], cand["init_pose"], opt, last_pose=cand["last_pose"])
    finally:
        for x in h:
            x.close()


def split_candidate(cand):
    """(map ground, map surf, query ground, query surf) point arrays of a candidate — the reference holds them as separate clouds
    (frame->feature_lidar->points_ground / points_surf); the synthetic candidates carry masks."""
    if "split" not in cand:
        cand["split"] = (np.ascontiguousarray(cand["map"][cand["map_ground"]]), np.ascontiguousarray(cand["map"][~cand["map_ground"]]),
                         np.ascontiguousarray(cand["query"][cand["query_ground"]]), np.ascontiguousarray(cand["query"][~cand["query_ground"]]))
    return cand["split"]


def resident_candidate(api, ctx, cand):
    """the candidate's four clouds as device-resident api.Cloud objects — what a pipeline that keeps frame->feature_lidar in HBM
    (lvf_lidar_extract's outputs, lvf_cloud_concat for BuildOldMapFrame) hands to the relocator; cached on the candidate"""
    if "resident" not in cand:
        cand["resident"] = tuple(api.Cloud(ctx, np.ascontiguousarray(a[:, :4] if a.shape[1] >= 4 else np.c_[a[:, :3], np.zeros(len(a))], np.float32)) for a in split_candidate(cand))
    return cand["resident"]


def evaluate_candidates_batched(api, ctx, cands, resolution=0.2, resident=False):
    """All of this rank's candidates in ONE launch chain (lvf_scan_match_batch): map indices and scans are created per candidate, the 4 x
    {ground, surf} solves of every candidate run side by side and the records come back with one read.  Returns the result list.
    `resident`: the clouds are taken from the device (resident_candidate) instead of being uploaded."""
    opt = api.scan_match_options(resolution, outer_iterations=4, prior_weight=0.0)
    handles, jobs = [], []
    try:
        # every candidate's map indices in ONE call (lvf_map_create_batch: the host waits of an index build are shared between them)
        parts = [resident_candidate(api, ctx, cand) if resident else split_candidate(cand) for cand in cands]
        clouds, thrs, where = [], [], []
        for k, (mg, ms, qg, qs) in enumerate(parts):
            if len(mg):
                clouds.append(mg); thrs.append(opt.thr_ground); where.append((k, "map_ground"))
            if len(ms):
                clouds.append(ms); thrs.append(opt.thr_surf); where.append((k, "map_surf"))
        maps = api.Map.create_batch(ctx, clouds, thrs) if clouds else []
        handles += maps
        made = [dict(map_ground=None, map_surf=None) for _ in cands]
        for m, (k, key) in zip(maps, where):
            made[k][key] = m
        for cand, (mg, ms, qg, qs), mk in zip(cands, parts, made):
            mpg, mps = mk["map_ground"], mk["map_surf"]
            scg = api.Scan(ctx, qg) if mpg is not None else None
            scs = api.Scan(ctx, qs) if mps is not None else None
            handles += [x for x in (scg, scs) if x is not None]
            jobs.append(dict(map_ground=mpg, scan_ground=scg, map_surf=mps, scan_surf=scs, map_pose=cand["map_pose"], frame_pose=cand["init_pose"],
                             last_pose=cand["last_pose"]))
        res, _ = api.scan_match_batch(ctx, jobs, opt, RELOCATE_BASE_SCORE)
        return res
    finally:
        for x in handles:
            x.close()


def relocalize(api, ctx, candidates, rank=0, world=1, device=None, workers=None, batched=False):
    """Evaluates this rank's share, exchanges the records, returns (best or None, all records).

    `workers`: extra api.Context objects on the SAME device.  A candidate is a chain of ~100 small launches and a dozen read-backs
    (two map indices, 4 x {ground, surf} solves on a few hundred points): latency, not throughput — with more than one candidate per
    GPU the chains of different candidates overlap on separate streams, one host thread per context (the C-ABI calls release the GIL);
    what the reference's Relocator thread does one after the other (relocator.cpp:196-206), and what host/relocalize_driver.cpp does in C++.
    `batched`: this rank's candidates go through ONE launch chain instead (lvf_scan_match_batch)."""
    table = empty_records(slots(len(candidates), world))
    mine = list(enumerate(owned(len(candidates), rank, world)))
    ctxs = [ctx] + list(workers or [])
    if batched:
        res = evaluate_candidates_batched(api, ctx, [candidates[cid] for _, cid in mine])
        for (s, cid), r in zip(mine, res):
            table[s] = make_record(cid, r.score, np.array(r.relative_o_c[:]))
    elif len(ctxs) == 1 or len(mine) <= 1:
        for s, cid in mine:
            res = evaluate_candidate(api, ctx, candidates[cid])
            table[s] = make_record(cid, res.score, np.array(res.relative_o_c[:]))
    else:
        from concurrent.futures import ThreadPoolExecutor

        def run(w):
            out = []
            for s, cid in mine[w::len(ctxs)]:
                res = evaluate_candidate(api, ctxs[w], candidates[cid])
                out.append((s, make_record(cid, res.score, np.array(res.relative_o_c[:]))))
            return out
        with ThreadPoolExecutor(max_workers=len(ctxs)) as pool:
            for part in pool.map(run, range(len(ctxs))):
                for s, rec in part:
                    table[s] = rec
    allrec = gather_records(table, world, device)
    return choose_best(allrec), allrec
