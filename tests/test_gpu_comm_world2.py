"""lvf_comm_* (csrc/comm.hip: RCCL opened by the library) with MORE THAN ONE rank: two processes, one GPU each, the 128-byte
unique id handed over through a file (the out-of-band channel include/lvf.h describes), one all-gather of 72-byte records
(src/lvio_fusion/src/relocator.cpp:196-206).  Needs two visible GPUs: skipped on the pool's 1-GPU boxes — there the N > 1 branch is
exercised through bench.py's gloo dry run (tests/test_gpu_bench_multirank.py) and world size 1 through tests/test_gpu_relocalize.py."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, time
    import numpy as np
    sys.path.insert(0, sys.argv[1])
    rank, world, d = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    from lvio_fusion_amd import api
    ctx = api.Context(rank)
    idf = os.path.join(d, "id.bin")
    if rank == 0:
        uid = api.comm_unique_id()
        with open(idf + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(idf + ".tmp", idf)
    else:
        t0 = time.time()
        while not os.path.exists(idf):
            if time.time() - t0 > 60:
                raise SystemExit("no unique id after 60 s")
            time.sleep(0.01)
        uid = open(idf, "rb").read()
    comm = api.Comm(ctx, world, rank, uid)
    assert comm.world_size == world and comm.rank == rank
    send = np.arange(9 * 4, dtype=np.float64) + 1000.0 * rank          # four 72-byte records
    got = comm.allgather(send)
    np.save(os.path.join(d, f"got_{rank}.npy"), got)
    comm.close(); ctx.close()
""")


def test_allgather_of_records_between_two_gpus(tmp_path):
    # (the device count comes from the HIP runtime directly: importing torch here would load ITS bundled RCCL / HIP libraries into the pytest process,
    # after which the library's own dlopen'ed /opt/rocm RCCL fails to initialise in tests/test_gpu_relocalize.py — seen once, round 6)
    import ctypes
    n = ctypes.c_int(0)
    hip = ctypes.CDLL("libamdhip64.so")
    if hip.hipGetDeviceCount(ctypes.byref(n)) != 0 or n.value < 2:
        pytest.skip("needs two visible GPUs (the pool's boxes have one)")
    d = str(tmp_path)
    script = os.path.join(d, "worker.py")
    open(script, "w").write(WORKER)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, script, ROOT, str(r), "2", d], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, so + se
    want = np.stack([np.arange(36, dtype=np.float64), np.arange(36, dtype=np.float64) + 1000.0])
    for r in range(2):
        assert np.array_equal(np.load(os.path.join(d, f"got_{r}.npy")), want), f"rank {r}"
