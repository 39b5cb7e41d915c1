"""lvf_lidar_extract repeated on one context while second / third host threads keep other contexts of the same GPU busy (another extraction loop,
a full-size kNN loop): every scan's clouds must equal the serial result bit for bit — the one-launch scans (ticketed tiles waiting for each
other inside a launch), the last-workgroup hand-overs and the two-stream tail under oversubscription.  usage: stress_extract.py [runs]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lvio_fusion_amd import api, synthetic as syn
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 100
ctx = api.Context(0)
ext = syn.lidar_extrinsic()
scans = [syn.raw_scan(seed=40 + k) for k in range(3)]
serial = []
for sc in scans:
    g, s = api.lidar_extract(ctx, sc, ext); serial.append((g.download(), s.download())); g.close(); s.close()
stop, laps = threading.Event(), [0, 0]
other_scan = syn.raw_scan(seed=77)
c3cfg = syn.config3_icp()
ready = threading.Barrier(3)
def other_extract():
    c2 = api.Context(0)
    sc = other_scan
    ready.wait()
    tmax = 0.0
    while not stop.is_set():
        t1 = time.perf_counter()
        g, s = api.lidar_extract(c2, sc, ext); g.close(); s.close(); laps[0] += 1
        tmax = max(tmax, time.perf_counter() - t1)
    print("other extraction: slowest call %.1f ms" % (1e3 * tmax))
    c2.close()
def knn_traffic():
    c3 = api.Context(0)
    c = c3cfg
    m = api.Map(c3, c["map"], c["thr_ground"]); q = api.Scan(c3, c["query"])
    ready.wait()
    while not stop.is_set():
        api.knn3(m, q, c["pose0"], c["thr_ground"]); laps[1] += 1
    m.close(); q.close(); c3.close()
ts = [threading.Thread(target=other_extract), threading.Thread(target=knn_traffic)]
for t in ts: t.start()
ready.wait()
bad = 0
t0 = time.perf_counter()
for r in range(runs):
    k = r % len(scans)
    g, s = api.lidar_extract(ctx, scans[k], ext)
    G, S = g.download(), s.download(); g.close(); s.close()
    if G.shape != serial[k][0].shape or S.shape != serial[k][1].shape or not np.array_equal(G.view(np.uint32), serial[k][0].view(np.uint32)) or not np.array_equal(S.view(np.uint32), serial[k][1].view(np.uint32)):
        bad += 1; print("run", r, "differs", G.shape, S.shape)
dt = time.perf_counter() - t0
stop.set()
for t in ts: t.join()
print("%d of %d scans differ; %.2f ms per scan under traffic; other extraction laps %d, kNN laps %d" % (bad, runs, 1e3 * dt / runs, laps[0], laps[1]))
