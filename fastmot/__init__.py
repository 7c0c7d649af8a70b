"""`import fastmot` drop-in: the reference's package name bound to the MI355X implementation.

The reference application (`/root/reference/app.py:10-12`: `import fastmot`, `import fastmot.models`,
`from fastmot.utils import ConfigDecoder, Profiler`) runs unmodified against this build when the repository
root is on `sys.path`: every public name and submodule of `fastmot` resolves to the object of the same name in
`fastmot_amd` (one implementation, two import names -- nothing is re-implemented here)."""
import importlib
import importlib.abc
import importlib.machinery
import importlib.util
import logging
import sys

import fastmot_amd
from fastmot_amd import (VideoIO, MultiTracker, KalmanFilter, MeasType, Flow, Track, models)  # noqa: F401

_SUBMODULES = ('models', 'models.label', 'models.yolo', 'models.reid', 'models.ssd', 'utils', 'utils.decoder',
               'utils.profiler', 'utils.visualization', 'videoio', 'tracker', 'track', 'flow', 'kalman_filter')
# modules that load the device library are aliased lazily (importing `fastmot` must work without a GPU)
_LAZY = ('mot', 'detector', 'feature_extractor')

for _name in _SUBMODULES:
    sys.modules[f'{__name__}.{_name}'] = importlib.import_module(f'fastmot_amd.{_name}')
utils = sys.modules[f'{__name__}.utils']

__all__ = list(fastmot_amd.__all__)

# app.py:49 configures logging.getLogger(fastmot.__name__): make the implementation's loggers its children
logging.getLogger('fastmot_amd').parent = logging.getLogger(__name__)


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """`import fastmot.mot` / `from fastmot.detector import YOLODetector`: the import system does not consult the
    module-level __getattr__ for submodule imports, so any `fastmot.<x>` that is not in sys.modules yet resolves here
    to the module object of `fastmot_amd.<x>` (imported at that moment: the lazy modules load the device library)."""

    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith(__name__ + '.'):
            return None
        try:
            found = importlib.util.find_spec('fastmot_amd.' + fullname[len(__name__) + 1:])
        except (ImportError, ValueError):
            return None
        return None if found is None else importlib.machinery.ModuleSpec(fullname, self)

    def create_module(self, spec):
        return importlib.import_module('fastmot_amd.' + spec.name[len(__name__) + 1:])

    def exec_module(self, module):
        pass


sys.meta_path.insert(0, _AliasFinder())


def __getattr__(name):
    if name in _LAZY:
        mod = importlib.import_module(f'fastmot_amd.{name}')
        sys.modules[f'{__name__}.{name}'] = mod
        return mod
    return getattr(fastmot_amd, name)
