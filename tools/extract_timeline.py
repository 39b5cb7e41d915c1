"""Timeline of one lvf_lidar_extract call under rocprofv3 (kernel + memcpy trace): run as
   rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d <dir> -o v -- python tools/extract_timeline.py ; python tools/extract_timeline.py --report <dir>"""
import os, sys, glob, csv
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2 and sys.argv[1] == "--report":
    d = sys.argv[2]
    ev = []
    for r in csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:]))
    for m in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
        for r in csv.DictReader(open(m)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
    ev.sort()
    # the last call: walk back from the end until a gap > 0.5 ms
    end = len(ev) - 1
    start = end
    while start > 0 and ev[start][0] - ev[start - 1][1] < 300000:
        start -= 1
    t0 = ev[start][0]
    busy = 0
    for e in ev[start:end + 1]:
        busy += e[1] - e[0]
        print(f"{(e[0] - t0) / 1e3:9.1f} {(e[1] - e[0]) / 1e3:8.1f}  {e[2]}")
    print("events %d, span %.1f us, busy %.1f us" % (end + 1 - start, (ev[end][1] - t0) / 1e3, busy / 1e3))
    sys.exit(0)
import numpy as np, time
from lvio_fusion_amd import api, synthetic as syn
ctx = api.Context(0)
raw = syn.raw_scan()
ext = syn.lidar_extrinsic()
for _ in range(6):
    g, sf = api.lidar_extract(ctx, raw, ext)
    ctx.synchronize(); g.close(); sf.close()
    time.sleep(0.002)
