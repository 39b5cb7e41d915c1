"""A/B of two builds of the library on ONE box: tools/ab_lib.py <base.so> <new.so>  ->  headline ms/step, 8 / 64 batched windows, per-kernel stage times"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for lib in sys.argv[1:3]:
    env = dict(os.environ, LVF_LIB_PATH=os.path.abspath(lib))
    for rep in range(2):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "200", "--no-cpu-baseline", "--legs", "batched_windows_8,batched_windows_64,small_windows_100"],
                           capture_output=True, text=True, env=env, cwd=ROOT)
        d = json.loads(p.stdout.strip().splitlines()[-1])
        print(os.path.basename(lib), "ms_per_step", d["ms_per_step"], "W8", d["batched_windows_8"]["lm_iters_per_sec_aggregate"], "W64", d["batched_windows_64"]["lm_iters_per_sec_aggregate"],
              "small100", d["small_windows_100"]["lm_iters_per_sec_aggregate"], "single small", d["small_windows_100"]["single_window_ms_per_iteration"])
