"""oracle/c_baseline.c (the compiled CPU baseline of bench.py, `cpu_baseline_compiled`) pinned against the numpy
restatement oracle/cv_oracle.py function by function -- bit-exact for the integer / float32 image routines, LK and
corner detectors, 1e-9 for the double-precision model fits -- and end to end on a short clip (identical tracks)."""
import numpy as np
import pytest

import c_baseline as cb
import cpu_tracker
import cv_oracle as cv
import scenes
from synthetic import SyntheticVideo


@pytest.fixture(scope='module')
def clip():
    return SyntheticVideo((640, 360), n_ids=8, n_frames=6, seed=3)


def test_gray_conversion_both_opencv_generations(clip):
    """cv2.cvtColor(BGR2GRAY) exists in two fixed-point widths across OpenCV 4.x (cv_oracle.GRAY_COEFF_BITS): both are
    implemented everywhere (kernels: fm_flow_cfg.gray_coeff_bits), the default is the 14-bit set of the 4.1.1 the
    reference pins.  Known answers by hand, numpy port == compiled port, and how far apart the two are."""
    assert cv.GRAY_COEFF_BITS == 14
    px = np.array([[[255, 255, 255], [0, 0, 0], [255, 0, 0], [0, 255, 0], [0, 0, 255], [17, 130, 201], [200, 100, 50]]], np.uint8)
    # (B cb + G cg + R cr + half) >> bits, by hand
    want14 = [(b * 1868 + g * 9617 + r * 4899 + 8192) >> 14 for b, g, r in px[0].tolist()]
    want15 = [(b * 3735 + g * 19235 + r * 9798 + 16384) >> 15 for b, g, r in px[0].tolist()]
    assert want14 == [255, 0, 29, 150, 76, 138, 96] and want15[:5] == [255, 0, 29, 150, 76]
    np.testing.assert_array_equal(cv.bgr2gray(px, 14)[0], want14)
    np.testing.assert_array_equal(cv.bgr2gray(px, 15)[0], want15)
    np.testing.assert_array_equal(cv.bgr2gray(px)[0], want14)                       # the default
    f = clip.frames[0]
    for bits in (14, 15):
        np.testing.assert_array_equal(cb.bgr2gray(f, bits), cv.bgr2gray(f, bits))
    d = cv.bgr2gray(f, 14).astype(int) - cv.bgr2gray(f, 15).astype(int)
    assert np.abs(d).max() == 1 and 0.001 < (d != 0).mean() < 0.2                   # one grey level on a fraction of the pixels


def test_images_exact(clip):
    f = clip.frames[0]
    g = cv.bgr2gray(f)
    np.testing.assert_array_equal(cb.bgr2gray(f), g)
    for dsize in ((320, 180), (64, 36), (211, 97)):
        np.testing.assert_array_equal(cb.resize_linear_u8(g, dsize), cv.resize_linear_u8(g, dsize))
    np.testing.assert_array_equal(cb.resize_nearest(g, (64, 36)), cv.resize_nearest(g, (64, 36)))
    small = cv.resize_linear_u8(g, (320, 180))
    for a, b in zip(cb.build_pyramid(small, 5, 5), cv.build_pyramid(small, 5, 5)):
        np.testing.assert_array_equal(a, b)
        dx, dy = cv.scharr_deriv(a)
        d = cb.scharr_deriv(a)
        np.testing.assert_array_equal(d[..., 0], dx)
        np.testing.assert_array_equal(d[..., 1], dy)


def test_lk_gftt_fast_exact(clip):
    g0, g1 = cv.bgr2gray(clip.frames[0]), cv.bgr2gray(clip.frames[1])
    s0, s1 = cv.resize_linear_u8(g0, (320, 180)), cv.resize_linear_u8(g1, (320, 180))
    rng = np.random.default_rng(1)
    pts = np.stack([rng.uniform(-4, 324, 500), rng.uniform(-4, 184, 500)], 1).astype(np.float32)
    n0, st0, e0 = cv.calc_optical_flow_pyr_lk(s0, s1, pts)
    n1, st1, e1 = cb.calc_optical_flow_pyr_lk(s0, s1, pts)
    np.testing.assert_array_equal(st1, st0)
    ok = st0 > 0
    np.testing.assert_array_equal(n1[ok], n0[ok])
    np.testing.assert_array_equal(e1[ok], e0[ok])
    assert ok.mean() > 0.5
    mask = np.full(g0.shape, 255, np.uint8)
    mask[100:160, 200:260] = 0
    for (x0, y0, x1, y1), md in (((180, 60, 300, 250), 7), ((10, 10, 90, 200), 4), ((400, 100, 470, 300), 9)):
        a = cv.good_features_to_track(g0[y0:y1, x0:x1], mask[y0:y1, x0:x1], 1000, 0.06, md)
        b = cb.good_features_to_track(g0[y0:y1, x0:x1], mask[y0:y1, x0:x1], 1000, 0.06, md)
        np.testing.assert_array_equal(b, a)
        assert len(a) > 5
    bg = cv.resize_linear_u8(g0, (64, 36))
    np.testing.assert_array_equal(cb.fast_detect(bg, 10), cv.fast_detect(bg, 10))
    np.testing.assert_array_equal(cb.fast_detect(s0, 10), cv.fast_detect(s0, 10))


def test_model_fits():
    rng = np.random.default_rng(2)
    Htrue = np.array([[1.002, 0.001, 3.0], [-0.0015, 0.999, -2.0], [1e-6, -2e-6, 1.]])
    a = np.stack([rng.uniform(0, 640, 200), rng.uniform(0, 360, 200)], 1)
    q = np.c_[a, np.ones(200)] @ Htrue.T
    b = q[:, :2] / q[:, 2:] + rng.normal(0, 0.3, (200, 2))
    b[::7] += rng.normal(0, 25, b[::7].shape)
    a32, b32 = a.astype(np.float32), b.astype(np.float32)
    H0, m0 = cv.ransac_run(cv.HomographyModel(), a32, b32, 3.0, 0.99, 500)
    H1, m1 = cb._ransac(0, a32, b32, 3.0, 0.99, 500)
    np.testing.assert_array_equal(m1, m0)
    np.testing.assert_allclose(H1, H0, rtol=1e-9, atol=1e-12)
    r0 = cv.lm_refine(cv.HomographyModel(), a32[m0], b32[m0], H0, 10)
    r1 = cb._lm(0, a32[m0], b32[m0], H0, 10)
    np.testing.assert_allclose(r1, r0, rtol=1e-8, atol=1e-11)
    c = a * 1.01 + np.array([5., -3.]) + rng.normal(0, 0.4, a.shape)
    c[::6] += rng.normal(0, 15, c[::6].shape)
    c32 = c.astype(np.float32)
    M0, k0 = cv.ransac_run(cv.AffinePartialModel(), a32[:60], c32[:60], 3.0, 0.99, 500)
    M1, k1 = cb._ransac(1, a32[:60], c32[:60], 3.0, 0.99, 500)
    np.testing.assert_array_equal(k1, k0)
    np.testing.assert_allclose(M1, M0, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(cb._lm(1, a32[:60][k0], c32[:60][k0], M0, 10),
                               cv.lm_refine(cv.AffinePartialModel(), a32[:60][k0], c32[:60][k0], M0, 10), rtol=1e-9, atol=1e-11)


def test_tracker_on_compiled_kernels_equals_numpy_oracle(clip):
    """OracleTracker on c_baseline == OracleTracker on cv_oracle over a clip (ids, boxes, keypoint counts)."""
    rng = np.random.default_rng(4)
    ident = rng.normal(0, 1, (clip.n_ids, 512))
    ident /= np.linalg.norm(ident, axis=1, keepdims=True)
    embs = [(lambda e: (e / np.linalg.norm(e, axis=1, keepdims=True)).astype(np.float32))(ident + rng.normal(0, 0.02, ident.shape))
            for _ in range(clip.n_frames)]
    kw = {k: v for k, v in scenes.tracker_kwargs().items() if k != 'flow_cfg'}
    runs = []
    for impl in (None, cb):
        cpu_tracker.OTrack.count = 0
        trk = cpu_tracker.OracleTracker(clip.size, 'euclidean', cv_impl=impl, **kw)
        trk.reset(1 / 30.)
        trk.init(clip.frames[0], clip.detections(0))
        rows = []
        for f in range(1, clip.n_frames):
            trk.compute_flow(clip.frames[f])
            trk.apply_kalman()
            trk.update(f, clip.detections(f), embs[f])
            rows.append([(tid, tuple(t.tlbr), t.age, t.hits, len(t.keypoints)) for tid, t in trk.tracks.items()])
        runs.append(rows)
    assert runs[0] == runs[1] and len(runs[0][-1]) >= 6
