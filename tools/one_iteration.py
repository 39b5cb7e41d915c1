import sys, os
sys.path.insert(0, os.getcwd())
from tools.run_batch import build
from lvio_fusion_amd import api
ctx = api.Context(0)
cfg, st, hs, prob = build(ctx, 0xC0FFEE)
opt = api.default_solver_options()
r, d = 1e4, 2.0
for _ in range(3):
    o = prob.lm_iteration(opt, r, d); r, d = o["radius"], o["decrease_factor"]
