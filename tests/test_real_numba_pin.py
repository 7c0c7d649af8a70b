"""Live pin of the goldens against the reference JIT-COMPILED BY A REAL NUMBA (build container only: needs
/root/reference and the image's /opt/conda Python 3.9 with Numba 0.54.1, see oracle/real_numba.py).  The committed
record of the full run is tests/golden/REAL_NUMBA_PIN.json; this test repeats a subset on every CPU run of the suite
where the ingredients exist, so that a change of the oracle, the goldens or the scenes cannot drift away from what the
jit-compiled reference computes."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
PY39 = Path('/opt/conda/bin/python3.9')
NUMBA = Path('/opt/conda/lib/python3.9/site-packages/numba')


@pytest.mark.skipif(not (PY39.exists() and NUMBA.exists() and Path('/root/reference/fastmot/tracker.py').exists()),
                    reason='needs the build container: /root/reference and the conda Python 3.9 with Numba')
def test_goldens_equal_the_jit_compiled_reference():
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1', NUMBA_CACHE_DIR='/tmp/fastmot_numba_cache')
    env.pop('PYTHONPATH', None)
    r = subprocess.run([str(PY39), str(ROOT / 'oracle' / 'pin_with_numba.py'), '--quick-check'], cwd='/tmp', env=env,
                       capture_output=True, text=True, timeout=900)
    sys.stdout.write(r.stdout[-2000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert 'OK' in r.stdout.splitlines()[-1]
    assert 'set order: 100 cases, 0 mismatches' in r.stdout
