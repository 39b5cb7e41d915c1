import os, sys
sys.path.insert(0, "/root/repo")
import bench
from lvio_fusion_amd import api, synthetic as syn
ctx = api.Context(0)
cfg, prob, h = bench.build_window(api, syn, ctx)
opt = api.default_solver_options()
r, d = 1e4, 2.0
for k in range(3):
    o = prob.lm_iteration(opt, r, d); r, d = o["radius"], o["decrease_factor"]
