"""Timeline of one lvf_cloud_voxel_filter call under rocprofv3 (kernel + memcpy trace): run as
   rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d <dir> -o v -- python tools/voxel_timeline.py ; python tools/voxel_timeline.py --report <dir>"""
import os, sys, glob, csv
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2 and sys.argv[1] == "--report":
    d = sys.argv[2]
    ev = []
    for r in csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-34:]))
    mc = glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True)
    if mc:
        for r in csv.DictReader(open(mc[0])):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", "")))
    ev.sort()
    idx = [i for i, e in enumerate(ev) if "k_voxel_emit" in e[2]]
    last = idx[-1]
    start = last
    while start > 0 and "k_voxel_key" not in ev[start][2]:
        start -= 1
    start = max(0, start - 3)
    t0 = ev[start][0]
    for e in ev[start:last + 1]:
        print(f"{(e[0] - t0) / 1e3:9.1f} {(e[1] - e[0]) / 1e3:8.1f}  {e[2]}")
    sys.exit(0)
import numpy as np
from lvio_fusion_amd import api, synthetic as syn
ctx = api.Context(0)
c3 = syn.config3_icp()
cl = api.Cloud(ctx, c3["query"])
for _ in range(5):
    v = cl.voxel_filter(0.4); ctx.synchronize(); v.close()
