import os

# the CPU oracle parallelises with OpenMP; on a 256-core GPU host the default team size makes its small dense loops
# pathologically slow (80 s per LM iteration at 60 keyframes instead of ~1 s) — the reference itself runs Ceres with
# num_threads = min(8, 0.75 nproc) (estimator.cpp:10)
os.environ.setdefault("OMP_NUM_THREADS", str(min(8, max(1, int(0.75 * (os.cpu_count() or 1))))))

import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure).  Built on demand with g++."""
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle
