"""TEST INFRASTRUCTURE ONLY -- build container only.

A REAL Numba for pinning the oracle.  The image ships an Anaconda Python 3.9 under /opt/conda with Numba 0.54.1 and
llvmlite 0.37 installed, but its NumPy was upgraded to 1.26, for which that Numba refuses to import (it supports
NumPy <= 1.20).  The JIT itself works once the import gets through; this module makes it get through, in-process and
without touching any installed file:

  * numpy.__version__ is reported as 1.20.3 while numba is imported (its version gate),
  * the names NumPy removed since (np.MachAr, np.bool, np.int, np.float, np.complex, np.object, ...) exist again, as
    numba's overload tables mention them at import time,
  * numba.np.ufunc._internal (the C helper of @vectorize, whose PyUFuncObject layout check fails on NumPy 1.26) is a stub:
    nothing the reference jit-compiles uses @vectorize.

Run with /opt/conda/bin/python3.9 (oracle/pin_with_numba.py does).  The reference pins Numba 0.48 (requirements.txt:3);
0.54.1 is what exists here, which is stated wherever a result of this module is quoted.  NUMBA_CACHE_DIR is pointed away
from the reference tree (its functions ask for cache=True) and byte-code writing is off: /root/reference stays untouched.
"""
import os
import sys
import types
import warnings


def import_numba():
    """-> (numba module, real numpy version string)"""
    if 'numba' in sys.modules and getattr(sys.modules['numba'], '__version__', None):
        import numpy as np
        return sys.modules['numba'], getattr(np, '_fastmot_real_version', np.__version__)
    os.environ.setdefault('NUMBA_CACHE_DIR', '/tmp/fastmot_numba_cache')
    sys.dont_write_bytecode = True
    warnings.filterwarnings('ignore')
    import numpy as np
    real = np.__version__
    np._fastmot_real_version = real
    major, minor = (int(x) for x in real.split('.')[:2])
    if (major, minor) > (1, 20):
        np.__version__ = '1.20.3'
        for name, val in (('MachAr', type('MachAr', (object,), {})), ('bool', bool), ('int', int), ('float', float),
                          ('complex', complex), ('object', object), ('str', str), ('long', int), ('unicode', str)):
            if name not in np.__dict__:
                setattr(np, name, val)

        class _Stub(types.ModuleType):
            def __getattr__(self, name):
                if name.startswith('__'):
                    raise AttributeError(name)
                cls = type(name, (object,), {})
                setattr(self, name, cls)
                return cls
        sys.modules['numba.np.ufunc._internal'] = _Stub('numba.np.ufunc._internal')
    import numba
    np.__version__ = real
    return numba, real
