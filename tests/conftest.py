import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the oracle's OpenMP team: the cores this process may use, not every core of the host (a GPU box exposes a 16-core share
# of a large machine; 256 threads on it made one oracle graph at batch 128 take 220 s)
os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(16, len(os.sched_getaffinity(0))))))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as ge
    return ge.import_package()


@pytest.fixture(scope="session")
def plref():
    from oracle import plref as p
    p.build()
    return p


@pytest.fixture(scope="session")
def gpu_ctx(pkg):
    """A plhip context on device 0.  GPU tests must FAIL (not skip) when the library or device is absent."""
    ctx = pkg.capi.Context(0)
    yield ctx
    ctx.close()


def golden_files(prefix):
    return sorted(glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def load_golden(path):
    z = np.load(path)
    return {k: z[k] for k in z.files}
