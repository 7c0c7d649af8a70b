import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / 'oracle', ROOT / 'tests'):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

GOLDEN = ROOT / 'tests' / 'golden'

# no weight files exist offline: the suites run the networks with seeded random parameters (explicit opt-in)
os.environ.setdefault('FASTMOT_RANDOM_WEIGHTS', '1')
# The PyTorch / NumPy references of the tests run on the host.  On a GPU box with a few hundred cores an OpenMP pool sized
# to the machine meets the affinity mask get_context() narrows to the GPU's NUMA node (runtime.bind_to_gpu_numa_node): four
# threads spinning per core made the small reference networks take 30-60 s each (GPU suite of round 5: 500 s, 380 of them in
# eight such tests).  A pool of at most 16 threads is as fast as these sizes get.
_THREADS = str(max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1))))
for _v in ('OMP_NUM_THREADS', 'MKL_NUM_THREADS', 'OPENBLAS_NUM_THREADS'):
    os.environ.setdefault(_v, _THREADS)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu through gpurun)')


@pytest.fixture(scope='session')
def ctx():
    """Process-wide HIP context; GPU tests fail (not skip) when the library cannot run."""
    from fastmot_amd.runtime import get_context
    return get_context()


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
