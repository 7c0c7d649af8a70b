"""TEST INFRASTRUCTURE ONLY -- ctypes front end of oracle/c_baseline.c: the OpenCV-restated routines of cv_oracle.py as
compiled C, with the same call signatures, so that cpu_tracker.OracleFlow / OracleTracker can run on either.

    python oracle/c_baseline.py            # builds oracle/_build/libc_baseline.so (gcc -O3 -march=x86-64-v3)

`time_clip` is bench.py's `cpu_baseline_compiled` leg: the reference's TensorRT-disabled CPU path (KLT + Kalman +
association; Python orchestration as in the reference, numeric kernels compiled) on a bounded sample of the benchmark
clip, one host thread -- SURVEY.md section 8d's "Numba-class proxy".  It is NOT the reference's published performance."""
import ctypes as C
import subprocess
import time
from pathlib import Path

import numpy as np

import cv_oracle as cv

HERE = Path(__file__).resolve().parent
SRC = HERE / 'c_baseline.c'
LIB = HERE / '_build' / 'libc_baseline.so'
_lib = None


def build(force=False):
    if not force and LIB.exists() and LIB.stat().st_mtime > SRC.stat().st_mtime:
        return LIB
    LIB.parent.mkdir(exist_ok=True)
    # x86-64-v3 (AVX2) rather than -march=native: the library is built in the build container and travels to the
    # GPU box (another CPU); with -march=native's AVX-512 auto-vectorisation it also segfaulted inside pytest's
    # Python process in the build sandbox (never stand-alone or under ASan/UBSan, which report nothing)
    cmd = ['gcc', '-O3', '-march=x86-64-v3', '-ffp-contract=off', '-fno-fast-math', '-shared', '-fPIC',
           '-o', str(LIB), str(SRC), '-lm']
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError('gcc failed:\n' + res.stderr)
    return LIB


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(str(LIB))
        _lib.cb_gftt.restype = C.c_int
        _lib.cb_fast.restype = C.c_int
        _lib.cb_ransac.restype = C.c_int
        _lib.cb_homography_fit.restype = C.c_int
    return _lib


def _p(a):
    return C.c_void_p(a.ctypes.data)


class _Level(C.Structure):
    _fields_ = [('I', C.c_void_p), ('J', C.c_void_p), ('D', C.c_void_p), ('w', C.c_int), ('h', C.c_int)]


def bgr2gray(img, bits=None):
    import cv_oracle
    img = np.ascontiguousarray(img, np.uint8)
    out = np.empty(img.shape[:2], np.uint8)
    lib().cb_bgr2gray(_p(img), C.c_int(out.size), _p(out), C.c_int(cv_oracle.GRAY_COEFF_BITS if bits is None else bits))
    return out


def resize_linear_u8(img, dsize):
    assert img.ndim == 2
    img = np.ascontiguousarray(img, np.uint8)
    out = np.empty((dsize[1], dsize[0]), np.uint8)
    lib().cb_resize(_p(img), C.c_int(img.shape[1]), C.c_int(img.shape[0]), _p(out), C.c_int(dsize[0]), C.c_int(dsize[1]))
    return out


def resize_nearest(img, dsize):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.empty((dsize[1], dsize[0]), np.uint8)
    lib().cb_resize_nearest(_p(img), C.c_int(img.shape[1]), C.c_int(img.shape[0]), _p(out), C.c_int(dsize[0]),
                            C.c_int(dsize[1]))
    return out


def build_pyramid(img, win, max_level):
    """buildOpticalFlowPyramid: stops when a level is not larger than the window."""
    pyr = [np.ascontiguousarray(img, np.uint8)]
    for _ in range(max_level):
        h, w = pyr[-1].shape
        nw, nh = (w + 1) // 2, (h + 1) // 2
        if nw <= win or nh <= win:
            break
        nxt = np.empty((nh, nw), np.uint8)
        lib().cb_pyr_down(_p(pyr[-1]), C.c_int(w), C.c_int(h), _p(nxt), C.c_int(nw), C.c_int(nh))
        pyr.append(nxt)
    return pyr


def scharr_deriv(img):
    img = np.ascontiguousarray(img, np.uint8)
    d = np.empty(img.shape + (2,), np.int16)
    lib().cb_scharr(_p(img), C.c_int(img.shape[1]), C.c_int(img.shape[0]), _p(d))
    return d


_pyr_cache = {}


def _pyramids(img, win, max_level, with_deriv):
    """Pyramid (+ Scharr derivatives) of an image, cached by identity: Flow.predict keeps the previous frame's small
    image and the reference's OpenCV call rebuilds both pyramids per call, so no caching across calls is assumed --
    the cache only avoids building the SAME image's pyramid twice inside one call."""
    pyr = build_pyramid(img, win, max_level)
    der = [scharr_deriv(p) for p in pyr] if with_deriv else None
    return pyr, der


def calc_optical_flow_pyr_lk(prev_img, next_img, prev_pts, win=5, max_level=5, max_count=10, epsilon=0.03,
                             min_eig_threshold=1e-4):
    pts = np.ascontiguousarray(prev_pts, np.float32).reshape(-1, 2)
    n = len(pts)
    I, D = _pyramids(prev_img, win, max_level, True)
    J, _ = _pyramids(next_img, win, max_level, False)
    levels = len(I)
    arr = (_Level * levels)()
    for l in range(levels):
        arr[l] = _Level(I[l].ctypes.data, J[l].ctypes.data, D[l].ctypes.data, I[l].shape[1], I[l].shape[0])
    out = np.empty((n, 2), np.float32)
    st = np.empty(n, np.uint8)
    err = np.empty(n, np.float32)
    eps2 = np.float32(min(max(epsilon, 0.), 10.) ** 2)
    lib().cb_lk(arr, C.c_int(levels), C.c_int(win), C.c_int(min(max(max_count, 0), 100)), C.c_float(eps2),
                C.c_float(min_eig_threshold), C.c_int(n), _p(pts), _p(out), _p(st), _p(err))
    return out, st, err


def good_features_to_track(img, mask, max_corners, quality, min_distance, block_size=3):
    """img / mask: 2-D uint8 views (row strides honoured, unit column stride)."""
    assert img.strides[1] == 1 and mask.strides[1] == 1
    h, w = img.shape
    out = np.empty((max_corners, 2), np.float32)
    n = lib().cb_gftt(_p(img), C.c_int(img.strides[0]), _p(mask), C.c_int(mask.strides[0]), C.c_int(w), C.c_int(h),
                      C.c_int(max_corners), C.c_float(quality), C.c_int(int(min_distance)), C.c_int(block_size), _p(out))
    return out[:n].copy()


def fast_detect(img, threshold=10):
    img = np.ascontiguousarray(img, np.uint8)
    cap = img.size
    out = np.empty((cap, 2), np.float32)
    n = lib().cb_fast(_p(img), C.c_int(img.shape[1]), C.c_int(img.shape[0]), C.c_int(int(threshold)), _p(out), C.c_int(cap))
    return out[:n].copy()


def _ransac(model, a, b, thr, conf, iters):
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    H = np.zeros(9)
    mask = np.zeros(len(a), np.uint8)
    ok = lib().cb_ransac(C.c_int(model), _p(a), _p(b), C.c_int(len(a)), C.c_double(thr), C.c_double(conf),
                         C.c_int(iters), _p(H), _p(mask))
    return (H if ok else None), mask.astype(bool)


def _lm(model, a, b, M, iters=10):
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    M = np.array(M, np.float64)
    lib().cb_lm_refine(C.c_int(model), _p(a), _p(b), C.c_int(len(a)), _p(M), C.c_int(iters))
    return M


def flow_estimate(prev_pts, cur_pts, status, begins, ends, bg_begin, bg_end, track_tlbr, size,
                  ransac_max_iter, ransac_conf, inlier_thresh):
    """Second half of Flow.predict (flow.py:215-263): same control flow as cv_oracle.flow_estimate, model fits in C."""
    P = np.asarray(prev_pts, np.float32).reshape(-1, 2)
    Cc = np.asarray(cur_pts, np.float32).reshape(-1, 2)
    st = np.asarray(status, bool)
    nT = len(begins)
    inl = np.zeros(len(P), bool)
    result = np.zeros(nT, np.int32)
    est = np.zeros((nT, 4))
    n_matched = np.zeros(nT, np.int32)
    g = np.arange(bg_begin, bg_end)[st[bg_begin:bg_end]]
    if len(g) < 4:
        return None, result, est, n_matched, inl
    H, mask = _ransac(0, P[g], Cc[g], 3.0, ransac_conf, ransac_max_iter)
    n_in = int(mask.sum())
    if H is not None and len(g) > 4 and n_in > 0:
        a, b = np.ascontiguousarray(P[g][mask]), np.ascontiguousarray(Cc[g][mask])
        H2 = np.zeros(9)
        if lib().cb_homography_fit(_p(a), _p(b), C.c_int(len(a)), _p(H2)):
            H = _lm(0, a, b, H2, 10)
    if H is None or n_in < inlier_thresh:
        return None, result, est, n_matched, inl
    inl[g[mask]] = True
    boxes = np.zeros((0, 4))
    xr, yr = np.rint(Cc[:, 0]).astype(np.int64), np.rint(Cc[:, 1]).astype(np.int64)
    inside = st & (xr >= 0) & (yr >= 0) & (xr < size[0]) & (yr < size[1])
    for k in range(nT):
        sl = slice(begins[k], ends[k])
        ok = inside[sl].copy()
        if len(boxes):
            x, y = xr[sl, None], yr[sl, None]
            ok &= ~((boxes[:, 0] <= x) & (x <= boxes[:, 2]) & (boxes[:, 1] <= y) & (y <= boxes[:, 3])).any(1)
        idx = np.arange(begins[k], ends[k])[ok]
        n = len(idx)
        n_matched[k] = n
        if n < 3:
            continue
        M, mask = _ransac(1, P[idx], Cc[idx], 3.0, ransac_conf, ransac_max_iter)
        if M is None:
            continue
        if n > 2 and mask.any():
            M = _lm(1, P[idx][mask], Cc[idx][mask], M, 10)
        tb = track_tlbr[k]
        tlx = tb[0] * M[0] + tb[1] * M[1] + M[2]
        tly = tb[0] * M[3] + tb[1] * M[4] + M[5]
        scale = np.sqrt(M[0] * M[0] + M[3] * M[3])
        if scale < 0.9 or scale > 1.1:
            scale = 1.
        w, h = tb[2] - tb[0] + 1, tb[3] - tb[1] + 1
        e = np.rint([tlx, tly, tlx + w * scale - 1., tly + h * scale - 1.])
        inl[idx[mask]] = True
        est[k] = e
        outside = min(e[2], size[0] - 1) < max(e[0], 0) or min(e[3], size[1] - 1) < max(e[1], 0)
        if outside or int(mask.sum()) < inlier_thresh:
            result[k] = 2
            continue
        result[k] = 1
        boxes = np.concatenate([boxes, [[max(int(e[0]), 0), max(int(e[1]), 0), max(int(e[2]), 0), max(int(e[3]), 0)]]])
    return H.reshape(3, 3), result, est, n_matched, inl


def time_clip(cfg, video, tracker_cfg, budget_s=10.0):
    """bench.py leg: frames/s of the compiled-kernel CPU path on a bounded sample of the clip (one thread)."""
    import sys
    import cpu_tracker
    me = sys.modules[__name__]
    kw = {k: v for k, v in vars(tracker_cfg).items() if k != 'flow_cfg'}
    trk = cpu_tracker.OracleTracker(cfg['size'], 'euclidean', cv_impl=me, **kw)
    trk.reset(1 / 30.)
    rng = np.random.default_rng(5)
    ident = rng.normal(0, 1, (video.n_ids, 512))
    ident /= np.linalg.norm(ident, axis=1, keepdims=True)

    def embs():
        e = ident + rng.normal(0, 0.02, ident.shape)
        return (e / np.linalg.norm(e, axis=1, keepdims=True)).astype(np.float32)
    labels = cfg.get('labels')
    trk.init(video.frames[0], video.detections(0, labels=labels))
    t0 = time.perf_counter()
    n = 0
    f = 0
    while True:
        f += 1
        frame = video.frames[f % video.n_frames]
        if f % cfg['skip'] == 0:
            trk.compute_flow(frame)
            trk.apply_kalman()
            trk.update(f, video.detections(f % video.n_frames, labels=labels), embs())
        else:
            trk.track(frame)
        n += 1
        if time.perf_counter() - t0 > budget_s or f >= 4 * video.n_frames:
            break
    dt = time.perf_counter() - t0
    return {'value': round(n / dt, 3), 'unit': 'frames/s', 'cores': 1, 'kind': 'compiled-port',
            'sample': f'{n} frames of the same {cfg["size"][0]}x{cfg["size"][1]}/{video.n_ids}-detection clip; Python '
                      'orchestration as in the reference, its compiled parts (OpenCV KLT / RANSAC) as plain C -O3 '
                      '(oracle/c_baseline.c), Kalman / association in numpy (oracle/np_oracle.py) where the reference '
                      'uses Numba; detector + ReID networks excluded (injected); a proxy, NOT the reference itself'}


if __name__ == '__main__':
    print(build(force=True))
