"""adapt::Solve on the configs[3] window through the Ceres-shaped surface with the adapter's phase timing (LVF_ADAPTER_TIMING=1):
python tools/adapter_timing.py   (GPU box)"""
import os, sys, subprocess, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from lvio_fusion_amd import api, synthetic as syn
from tests.test_gpu_adapter import _dump, _cam_vec, _sorted_by_kf
ctx = api.Context(0)
cfg = _sorted_by_kf(syn.config4_window())
pre = api.preintegrate_or_none(ctx, cfg)
d = tempfile.mkdtemp()
tc, tf, po = cfg["tc"], cfg["tf"], cfg["po"]
_dump(d, "meta.i32", [cfg["n_kf"], cfg["n_lm"], 1, 0, -1], np.int32)
for name, key in (("poses", "poses"), ("vel", "vel"), ("ba", "ba"), ("bg", "bg"), ("inv_depth", "inv_depth"), ("w_kf", "w_kf")):
    _dump(d, name + ".f64", cfg[key], np.float64)
_dump(d, "cam0.f64", _cam_vec(cfg["cam0"]), np.float64); _dump(d, "cam1.f64", _cam_vec(cfg["cam1"]), np.float64)
_dump(d, "tc_left_ob.f64", tc["left_ob"], np.float64); _dump(d, "tc_right_ob.f64", tc["right_ob"], np.float64)
_dump(d, "tc_lm.i32", tc["lm_idx"], np.int32); _dump(d, "tc_kf.i32", tc["kf_idx"], np.int32)
_dump(d, "tf_first_ob.f64", tf["first_ob"], np.float64); _dump(d, "tf_ob.f64", tf["ob"], np.float64)
_dump(d, "tf_lm.i32", tf["lm_idx"], np.int32); _dump(d, "tf_kf1.i32", tf["kf1_idx"], np.int32); _dump(d, "tf_kf2.i32", tf["kf2_idx"], np.int32)
_dump(d, "po_ob.f64", po["ob"], np.float64); _dump(d, "po_pw.f64", po["pw"], np.float64)
_dump(d, "po_kf.i32", po["kf_idx"], np.int32); _dump(d, "po_pw_idx.i32", po["pw_idx"], np.int32)
_dump(d, "preint.f64", pre, np.float64)
_dump(d, "imu_i.i32", [f["kf_i"] for f in cfg["imu"]], np.int32); _dump(d, "imu_j.i32", [f["kf_j"] for f in cfg["imu"]], np.int32)
env = dict(os.environ, LVF_SELFTEST_REPEAT="1", LVF_ADAPTER_TIMING="1", LVF_SELFTEST_TICK=os.environ.get("LVF_SELFTEST_TICK", ""))
if not env["LVF_SELFTEST_TICK"]:
    del env["LVF_SELFTEST_TICK"]
p = subprocess.run([os.path.join(ROOT, "lvio_fusion_amd", "host", "adapter_selftest"), "window", d], capture_output=True, text=True, env=env)
print(p.stderr[-3000:])
