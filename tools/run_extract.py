"""Feature extraction of one raw scan, N times: ms per scan on the device-counted path and (argument `host`) the host-counted one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lvio_fusion_amd import api, synthetic as syn
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
ctx = api.Context(0)
scan = syn.raw_scan(seed=0x5CA9); ext = syn.lidar_extrinsic()
for host in ([True] if "host" in sys.argv[2:] else [False] if "dev" in sys.argv[2:] else [False, True, False, True]):
    api.extract_host_counts(ctx, host)
    for _ in range(3):
        g, s = api.lidar_extract(ctx, scan, ext); g.close(); s.close()
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        g, s = api.lidar_extract(ctx, scan, ext)
        if _ < n - 1:
            g.close(); s.close()
    ctx.synchronize(); dt = time.perf_counter() - t0
    print("%s counts: %.3f ms per scan (%d points -> %d ground, %d surf)" % ("host" if host else "device", 1e3 * dt / n, len(scan), len(g), len(s)))
    g.close(); s.close()
# the same scan handed over from page-locked memory (lvf_host_alloc): no pinning by the runtime ahead of the DMA
import ctypes as C
nbytes = scan.nbytes
ptr = ctx.L.lvf_host_alloc(nbytes)
if ptr:
    pinned = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(scan.size,)).reshape(scan.shape)
    pinned[:] = scan
    api.extract_host_counts(ctx, False)
    for _ in range(3):
        g, s = api.lidar_extract(ctx, pinned, ext); g.close(); s.close()
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        g, s = api.lidar_extract(ctx, pinned, ext); g.close(); s.close()
    ctx.synchronize(); dt = time.perf_counter() - t0
    print("device counts, scan in page-locked memory: %.3f ms per scan" % (1e3 * dt / n))
    ctx.L.lvf_host_free(ptr, nbytes)
