"""configs[3] LM iteration time with random landmark ids (the bench default) and ids in creation order (what a live front-end produces)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from lvio_fusion_amd import api, synthetic as syn
ctx = api.Context(0)
for ids in (False, True):
    cfg, prob, h = bench.build_window(api, syn, ctx, ids_by_birth=ids)
    opt = bench.fixed_iterations(api, 20)
    for r in range(3):
        bench.reset_state(api, h[4], cfg); ctx.synchronize(); t0 = time.perf_counter(); s = prob.solve(opt); dt = time.perf_counter() - t0
    print("ids_by_birth", ids, "%.4f ms/iteration" % (1e3 * dt / s.num_iterations))
    if os.environ.get("LVF_STAGES"):
        bench.reset_state(api, h[4], cfg)
        print("  stages (us):", " ".join("%s=%.1f" % (n, us) for n, us, _ in prob.stage_times(opt, reps=10)))
