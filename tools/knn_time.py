"""The association alone on configs[2] (ground and surf gate): ms per launch.  tools/knn_time.py [reps]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lvio_fusion_amd import api, synthetic

def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    ctx = api.Context()
    c3 = synthetic.config3_icp()
    mp = api.Map(ctx, c3["map"], c3["thr_ground"]); sc = api.Scan(ctx, c3["query"])
    for name, thr in (("ground", c3["thr_ground"]), ("surf", c3["thr_surf"])):
        for _ in range(3):
            api.knn3(mp, sc, c3["pose0"], thr)
        ctx.synchronize()
        ctx.timer_begin()
        for _ in range(reps):
            api.knn3(mp, sc, c3["pose0"], thr)
        ctx.timer_end()
        print(name, "ms", ctx.timer_ms() / reps)

if __name__ == "__main__":
    main()
