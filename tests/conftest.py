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
