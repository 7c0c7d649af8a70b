"""ctypes stubs of the diagnostic entry points (include/fastmot_hip_diag.h).  They exist only in a library built with
-DFM_DIAG:   FASTMOT_EXTRA_HIPCC_FLAGS=-DFM_DIAG python -m fastmot_amd.build   (the product library does not export
them, and fastmot_amd/_lib.py does not bind them)."""
import ctypes as C

import numpy as np

from fastmot_amd._lib import check, _ptr


def _need(ctx, name):
    if not hasattr(ctx.lib, name):
        raise RuntimeError(f'{name} is not exported: rebuild with FASTMOT_EXTRA_HIPCC_FLAGS=-DFM_DIAG')
    return getattr(ctx.lib, name)


def flow_lk_diag(ctx, n_capture=0):
    """Read-out of the LK kernel variants (option 'lk_variant'): 16 counters, and for the capture variant the
    headers [n, 4] and records [n, 80, 12, 64] of the last call."""
    counters = np.zeros(16, np.int32)
    hdr = rec = None
    if n_capture:
        hdr = np.zeros((n_capture, 4), np.int32)
        rec = np.zeros((n_capture, 80, 12, 64), np.int32)
    check(_need(ctx, 'fm_flow_lk_diag_read')(ctx.handle, _ptr(counters), C.c_int(n_capture), _ptr(hdr), _ptr(rec)))
    return counters, hdr, rec


def diag_pkhaz(ctx, variant, waves=600, iters=2000):
    out = np.zeros(8, np.int32)
    check(_need(ctx, 'fm_diag_pkhaz')(ctx.handle, C.c_int(variant), C.c_int(waves), C.c_int(iters), _ptr(out)))
    return out


def diag_pkhaz2(ctx, victim, aggressor, launches=16):
    out = np.zeros(8, np.int32)
    check(_need(ctx, 'fm_diag_pkhaz2')(ctx.handle, C.c_int(victim), C.c_int(aggressor), C.c_int(launches), _ptr(out)))
    return out
