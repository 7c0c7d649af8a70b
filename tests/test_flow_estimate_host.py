"""CPU side of Flow.predict -- fm_flow_estimate (camera-motion RANSAC + per-track RANSAC / LM, flow.py:215-263;
host C++ inside libfastmot_hip.so, no device work) against oracle/cv_oracle.flow_estimate.

The library fits all tracks speculatively in parallel (worker pool) and re-fits a track sequentially when one of its
candidate points lies under a box accepted earlier (the foreground mask of the reference): the scenes below are built
so that BOTH paths are taken, with 1, 2 and 7 threads."""
import ctypes as C
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]

WORKER = r'''
import sys, ctypes as C
import numpy as np
sys.path[:0] = [%(root)r, %(root)r + '/oracle']
import cv_oracle as cv
from fastmot_amd import _lib
lib = _lib.load()


class HostOnlyCtx(_lib.HipContext):
    """fm_flow_estimate never touches the device: a dummy non-null handle is enough."""
    def __init__(self):
        self.lib = lib
        self._ctx = C.c_void_p(1)
    def close(self):
        pass
    __del__ = close


def scene(seed, n_trk, overlap):
    rng = np.random.default_rng(seed)
    size = (1920, 1080)
    Htrue = np.array([[1.001, 0.0005, 2.5], [-0.0008, 0.9995, -1.5], [5e-7, -8e-7, 1.]])
    n_bg = 300
    bgp = np.stack([rng.uniform(0, size[0], n_bg), rng.uniform(0, size[1], n_bg)], 1)
    q = np.c_[bgp, np.ones(n_bg)] @ Htrue.T
    bgc = q[:, :2] / q[:, 2:] + rng.normal(0, 0.3, (n_bg, 2))
    bgc[::6] += rng.normal(0, 30, bgc[::6].shape)
    tracks, prev, cur, begins, ends = [], [], [], [], []
    base_x = rng.uniform(100, 1500, n_trk)
    for k in range(n_trk):
        x0 = base_x[k] if not overlap or k %% 2 == 0 else base_x[k - 1] + rng.uniform(10, 40)   # pairs overlap
        y0 = rng.uniform(50, 700)
        w, h = rng.uniform(40, 90), rng.uniform(120, 250)
        tracks.append([x0, y0, x0 + w, y0 + h])
        n = int(rng.integers(2, 120))
        p = np.stack([rng.uniform(x0, x0 + w, n), rng.uniform(y0, y0 + h, n)], 1)
        c = p * rng.uniform(0.98, 1.02) + rng.normal(0, 4, 2) + rng.normal(0, 0.4, (n, 2))
        c[::5] += rng.normal(0, 20, c[::5].shape)
        begins.append(sum(len(a) for a in prev)); prev.append(p); cur.append(c); ends.append(begins[-1] + n)
    # closest-first order as Flow.predict delivers it
    order = np.argsort([-t[3] for t in tracks], kind='stable')
    tracks = np.array(tracks)[order]
    P = np.concatenate([prev[i] for i in order] + [bgp]).astype(np.float32)
    Cc = np.concatenate([cur[i] for i in order] + [bgc]).astype(np.float32)
    lens = [len(prev[i]) for i in order]
    ends = np.cumsum(lens).astype(np.int32)
    begins = (ends - np.array(lens)).astype(np.int32)
    status = rng.random(len(P)) > 0.05
    return (P, Cc, status, begins, ends, int(ends[-1]), len(P) - 1, np.rint(tracks), size, 500, 0.99, 4)


ctx = HostOnlyCtx()
redone = 0
for seed, n_trk, overlap in ((1, 12, False), (2, 50, True), (3, 50, True), (4, 31, True), (5, 3, False)):
    args = scene(seed, n_trk, overlap)
    H, res, est, nm, inl = ctx.flow_estimate(*args)
    eH, eres, eest, enm, einl = cv.flow_estimate(*args)
    assert (H is None) == (eH is None)
    np.testing.assert_array_equal(res, eres)
    np.testing.assert_array_equal(nm, enm)
    np.testing.assert_array_equal(inl, einl)
    np.testing.assert_array_equal(est, eest)
    np.testing.assert_allclose(H, eH, rtol=1e-6, atol=1e-8)
    assert (res == 1).sum() >= n_trk // 3
t5 = (C.c_double * 5)()
import os
os.environ['FASTMOT_FLOW_TIMING_VERBOSE'] = '1'
lib.fm_flow_timing(t5, 0)
print('OK')
'''


@pytest.mark.parametrize('threads', [1, 2, 7])
def test_flow_estimate_equals_oracle(threads):
    env = dict(os.environ, FASTMOT_FLOW_THREADS=str(threads))
    res = subprocess.run([sys.executable, '-c', WORKER % {'root': str(ROOT)}], capture_output=True, text=True, env=env,
                         timeout=600)
    assert res.returncode == 0 and 'OK' in res.stdout, res.stderr[-3000:]
    # the overlapping scenes must have exercised the sequential re-fit path
    assert 'tracks re-fitted under the mask' in res.stderr
    redone = float(res.stderr.split('flow_estimate:')[1].split()[0])
    assert (redone > 0) == (threads > 1), res.stderr[-500:]      # one thread = the plain sequential algorithm
