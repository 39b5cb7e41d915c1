"""configs[4] timing by part: where one loop-closure candidate's time goes on one stream (host masking, map indices, scan uploads,
the 4 x {ground, surf} solves).  Mapping::Relocate = src/lvio_fusion/src/mapping.cpp:251-300."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lvio_fusion_amd import api, synthetic as syn
ctx = api.Context(0)
cands = syn.config5_candidates(8)
opt = api.scan_match_options(0.2, outer_iterations=4, prior_weight=0.0)
T = {k: 0.0 for k in ("mask", "map_ground", "map_surf", "scans", "scan_match", "close")}
for rep in range(3):
    for k in T:
        T[k] = 0.0
    for c in cands:
        t = time.perf_counter()
        mg, ms = c["map"][c["map_ground"]], c["map"][~c["map_ground"]]
        qg, qs = c["query"][c["query_ground"]], c["query"][~c["query_ground"]]
        T["mask"] += time.perf_counter() - t; t = time.perf_counter()
        mpg = api.Map(ctx, mg, opt.thr_ground); ctx.synchronize()
        T["map_ground"] += time.perf_counter() - t; t = time.perf_counter()
        mps = api.Map(ctx, ms, opt.thr_surf); ctx.synchronize()
        T["map_surf"] += time.perf_counter() - t; t = time.perf_counter()
        scg, scs = api.Scan(ctx, qg), api.Scan(ctx, qs); ctx.synchronize()
        T["scans"] += time.perf_counter() - t; t = time.perf_counter()
        res = api.scan_match(mpg, scg, mps, scs, c["map_pose"], c["init_pose"], opt, last_pose=c["last_pose"])
        T["scan_match"] += time.perf_counter() - t; t = time.perf_counter()
        for h in (mpg, scg, mps, scs):
            h.close()
        T["close"] += time.perf_counter() - t
    print("8 candidates by part (ms): " + "  ".join("%s %.2f" % (k, 1e3 * v) for k, v in T.items()) + "  total %.2f" % (1e3 * sum(T.values())))
print("sizes of candidate 0: map ground / surf %d / %d, query ground / surf %d / %d" % (
    int(cands[0]["map_ground"].sum()), int((~cands[0]["map_ground"]).sum()), int(cands[0]["query_ground"].sum()), int((~cands[0]["query_ground"]).sum())))
# ---- the same eight candidates through ONE launch chain (lvf_scan_match_batch); clouds split beforehand (the reference holds them as separate clouds)
from lvio_fusion_amd import relocalize as rl
for c in cands:
    rl.split_candidate(c)
for rep in range(4):
    t0 = time.perf_counter()
    hs, jobs = [], []
    for c in cands:
        mg, ms, qg, qs = rl.split_candidate(c)
        h = [api.Map(ctx, mg, opt.thr_ground), api.Scan(ctx, qg), api.Map(ctx, ms, opt.thr_surf), api.Scan(ctx, qs)]
        hs += h
        jobs.append(dict(map_ground=h[0], scan_ground=h[1], map_surf=h[2], scan_surf=h[3], map_pose=c["map_pose"], frame_pose=c["init_pose"], last_pose=c["last_pose"]))
    ctx.synchronize(); t1 = time.perf_counter()
    res, best = api.scan_match_batch(ctx, jobs, opt)
    t2 = time.perf_counter()
    for h in hs:
        h.close()
    t3 = time.perf_counter()
    print("batched: maps + scans %.2f ms, lvf_scan_match_batch %.2f ms, close %.2f ms, total %.2f ms; best %d, scores %s" % (
        1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t3 - t0), best, [r.score - 20 for r in res]))
t0 = time.perf_counter(); b, rec = rl.relocalize(api, ctx, cands, batched=True); print("relocalize(batched=True): %.2f ms" % (1e3 * (time.perf_counter() - t0)))
t0 = time.perf_counter(); b, rec = rl.relocalize(api, ctx, cands); print("relocalize(one at a time, clouds split beforehand): %.2f ms" % (1e3 * (time.perf_counter() - t0)))
# ---- the batched form by part (what relocalize(batched=True) does)
for rep in range(4):
    t0 = time.perf_counter()
    parts = [rl.split_candidate(c) for c in cands]
    clouds, thrs = [], []
    for mg, ms, qg, qs in parts:
        clouds += [mg, ms]; thrs += [opt.thr_ground, opt.thr_surf]
    t1 = time.perf_counter()
    maps = api.Map.create_batch(ctx, clouds, thrs)
    t2 = time.perf_counter()
    scans = []
    for mg, ms, qg, qs in parts:
        scans += [api.Scan(ctx, qg), api.Scan(ctx, qs)]
    t3 = time.perf_counter()
    jobs = [dict(map_ground=maps[2 * k], scan_ground=scans[2 * k], map_surf=maps[2 * k + 1], scan_surf=scans[2 * k + 1], map_pose=c["map_pose"], frame_pose=c["init_pose"],
                 last_pose=c["last_pose"]) for k, c in enumerate(cands)]
    res, _ = api.scan_match_batch(ctx, jobs, opt, 20.0)
    t4 = time.perf_counter()
    for h in maps + scans:
        h.close()
    t5 = time.perf_counter()
    print("batched by part (ms): split %.3f  Map.create_batch(16) %.3f  16 scans %.3f  scan_match_batch %.3f  close %.3f  total %.3f" % tuple(
        1e3 * x for x in (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t5 - t0)))
