"""configs[4] timing: eight loop-closure candidates on one stream (and, with an argument, on that many extra contexts)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lvio_fusion_amd import api, synthetic as syn, relocalize as rl
ctx = api.Context(0)
cands = syn.config5_candidates(8)
workers = [api.Context(0) for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 0)]
rl.relocalize(api, ctx, cands[:max(1, len(workers) + 1)], workers=workers)
ctx.synchronize()
for _ in range(3):
    t0 = time.perf_counter()
    best, rec = rl.relocalize(api, ctx, cands, workers=workers)
    print("8 candidates: %.2f ms (%d extra contexts)" % (1e3 * (time.perf_counter() - t0), len(workers)))
