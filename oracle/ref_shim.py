"""TEST INFRASTRUCTURE ONLY -- not part of the product path.

Imports the *unmodified* reference modules from /root/reference so that
(a) the numpy restatement in oracle/ can be validated against them and
(b) golden vectors can be generated (oracle/make_golden.py).

The reference cannot be imported as a package here: its __init__ files pull
cv2 / cupy / tensorrt / numba, none of which exist in this image (SURVEY.md
section 8c).  Recipe (proven during the survey):
  * a fake `numba` whose njit/jit decorators are the identity and prange=range;
  * empty stub modules for cv2, cupy, cupyx(.scipy.ndimage), tensorrt;
  * synthetic package objects for fastmot, fastmot.utils, fastmot.models with
    __path__ set, so the real __init__.py files are never executed.
Caveat: numba `fastmath`/`parallel` reassociation is NOT reproduced; the shim
yields the IEEE sequential answer of the reference source.  One Numba semantic
IS reproduced because it is observable in the track IDs: inside @njit functions
the builtin `set` is Numba's hash set (restated in oracle/numba_set.py), whose
iteration order differs from CPython's (matching.py:59-60, detector.py:196).
What the de-jitting leaves open was measured with a real Numba 0.54.1
(load_reference(real_numba=True), oracle/pin_with_numba.py, record in
tests/golden/REAL_NUMBA_PIN.json): every integer / boolean array of every golden
file (track ids, rounded boxes, life-cycle flags, history order, NMS keep lists,
SSD merge) is identical; float arrays (Kalman states, distances) differ by at
most 1.3e-10 (float64) / 4.3e-8 (float32 cosine terms) -- fastmath.

/root/reference only exists in the build container; nothing on the GPU box may
call load_reference().
"""
import importlib
import importlib.util
import sys
import types
from pathlib import Path

REF_ROOT = Path('/root/reference')


def reference_available():
    return (REF_ROOT / 'fastmot' / 'tracker.py').exists()


def _numba_semantics(fn):
    """What the jitted body would see where CPython differs observably: the builtin `set` inside @njit code is Numba's
    own hash set, whose iteration order is not CPython's (oracle/numba_set.py).  The function object is rebuilt over a
    copy of its module globals in which `set` is that container; the reference's source stays untouched.  Only
    functions that mention `set` are rebuilt (closures and everything else pass through)."""
    if not isinstance(fn, types.FunctionType) or 'set' not in fn.__code__.co_names or fn.__closure__:
        return fn
    import numba_set
    g = dict(fn.__globals__)
    g['set'] = numba_set.NumbaIntSet
    out = types.FunctionType(fn.__code__, g, fn.__name__, fn.__defaults__, None)
    out.__kwdefaults__ = fn.__kwdefaults__
    out.__dict__.update(fn.__dict__)
    out.__doc__ = fn.__doc__
    return out


def _fake_numba():
    nb = types.ModuleType('numba')

    def _decorator(*args, **kwargs):
        if len(args) == 1 and callable(args[0]) and not kwargs:
            return _numba_semantics(args[0])
        return _numba_semantics

    nb.njit = _decorator
    nb.jit = _decorator
    nb.prange = range
    return nb


def _stub(name, **attrs):
    mod = types.ModuleType(name)
    for key, val in attrs.items():
        setattr(mod, key, val)
    return mod


def load_reference(prefix='fastmot', real_numba=False):
    """Returns a namespace with the reference modules:
    rect, distance, matching, numba_utils, kalman_filter, track, tracker, flow, label

    real_numba: the reference's @nb.njit functions are compiled by a real Numba (oracle/real_numba.py; only where one
    is importable: the /opt/conda Python 3.9 of the build container) instead of being de-jitted -- fastmath, typed
    containers and all.  oracle/pin_with_numba.py uses this to check the committed goldens.
    """
    if not reference_available():
        raise RuntimeError('/root/reference is not present (GPU box?)')
    import numpy as np

    saved = {k: sys.modules.get(k) for k in
             ('numba', 'cv2', 'cupy', 'cupyx', 'cupyx.scipy', 'cupyx.scipy.ndimage', 'tensorrt')}
    if real_numba:
        import real_numba as _rn
        sys.modules['numba'] = _rn.import_numba()[0]
    else:
        sys.modules['numba'] = _fake_numba()

    class _Fast:
        def detect(self, *a, **k):
            return []
    sys.modules['cv2'] = _stub('cv2', FastFeatureDetector_create=lambda **k: _Fast())
    sys.modules['cupy'] = _stub('cupy')
    cupyx = _stub('cupyx', empty_pinned=lambda shape, dtype=float: np.empty(shape, dtype),
                  empty_like_pinned=lambda a: np.empty_like(a))
    cupyx.scipy = _stub('cupyx.scipy')
    cupyx.scipy.ndimage = _stub('cupyx.scipy.ndimage')
    sys.modules['cupyx'] = cupyx
    sys.modules['cupyx.scipy'] = cupyx.scipy
    sys.modules['cupyx.scipy.ndimage'] = cupyx.scipy.ndimage
    sys.modules['tensorrt'] = _stub('tensorrt')

    root = REF_ROOT / 'fastmot'
    pkgs = {}
    for name, path in ((prefix, root), (prefix + '.utils', root / 'utils'),
                       (prefix + '.models', root / 'models')):
        pkg = types.ModuleType(name)
        pkg.__path__ = [str(path)]
        pkg.__package__ = name
        sys.modules[name] = pkg
        pkgs[name] = pkg

    def _load(modname, relpath):
        full = prefix + '.' + modname
        spec = importlib.util.spec_from_file_location(full, root / relpath)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[full] = mod
        spec.loader.exec_module(mod)
        parent, _, leaf = full.rpartition('.')
        setattr(sys.modules[parent], leaf, mod)
        return mod

    ns = types.SimpleNamespace()
    ns.label = _load('models.label', 'models/label.py')
    pkgs[prefix + '.models'].get_label_name = ns.label.get_label_name
    pkgs[prefix + '.models'].set_label_map = ns.label.set_label_map
    ns.rect = _load('utils.rect', 'utils/rect.py')
    ns.numba_utils = _load('utils.numba', 'utils/numba.py')
    ns.distance = _load('utils.distance', 'utils/distance.py')
    ns.matching = _load('utils.matching', 'utils/matching.py')
    ns.kalman_filter = _load('kalman_filter', 'kalman_filter.py')
    ns.track = _load('track', 'track.py')
    ns.flow = _load('flow', 'flow.py')
    ns.tracker = _load('tracker', 'tracker.py')
    ns.saved_modules = saved
    return ns


def unload_reference(ns, prefix='fastmot'):
    """Restore sys.modules (so the product package / real numba are not shadowed)."""
    for key in [k for k in sys.modules if k == prefix or k.startswith(prefix + '.')]:
        del sys.modules[key]
    for key, val in ns.saved_modules.items():
        if val is None:
            sys.modules.pop(key, None)
        else:
            sys.modules[key] = val
