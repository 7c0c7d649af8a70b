"""GPU parity: batched Kalman kernels (kalman.hip) through the C ABI
  * oracle=reference: golden vectors generated from /root/reference (tests/golden/kalman_kat.npz)
  * oracle=restated : oracle/np_oracle.py on seeded random inputs (sizes 1..300 tracks)
Tolerance: fp64 round-off (rtol 1e-9 on covariances, 1e-11 on means) -- the reference multiplies
8x8 matrices through BLAS with an unspecified summation order; rounded boxes must be identical."""
import numpy as np
import pytest

import np_oracle as o

pytestmark = pytest.mark.gpu

KF_DEFAULT = dict(std_factor_acc=2.25, std_offset_acc=78.5, std_factor_det=(0.08, 0.08),
                  std_factor_klt=(0.14, 0.14), min_std_det=(4.0, 4.0), min_std_klt=(5.0, 5.0),
                  init_pos_weight=5, init_vel_weight=12, vel_coupling=0.6, vel_half_life=2)


def configure(ctx, dt):
    ctx.kf_configure(dt, **KF_DEFAULT)
    ctx.set_frame_rect([0., 0., 1919., 1079.])


@pytest.mark.parametrize('tag', ['dt30', 'dt12'])
def test_golden_chain(ctx, golden_dir, tag):
    g = np.load(golden_dir / 'kalman_kat.npz')
    configure(ctx, float(g[f'{tag}_dt']))
    boxes, H = g[f'{tag}_boxes'], g[f'{tag}_H']
    n = len(boxes)
    slots = np.arange(10, 10 + n)
    ctx.trk_create(slots, boxes)
    m, c = ctx.trk_get_state(slots)
    np.testing.assert_array_equal(m, g[f'{tag}_create_m'])
    np.testing.assert_allclose(c, g[f'{tag}_create_c'], rtol=1e-15)
    # warp only
    ctx.trk_set_state(slots, g[f'{tag}_start_m'], g[f'{tag}_start_c'])
    zeros, ones = np.zeros((n, 4)), np.ones(n)
    ctx.trk_step_ops(1, slots, H, zeros, np.zeros(n, np.uint8), ones)
    m, c = ctx.trk_get_state(slots)
    np.testing.assert_allclose(m, g[f'{tag}_warp_m'], rtol=1e-12, atol=1e-10)
    np.testing.assert_allclose(c, g[f'{tag}_warp_c'], rtol=1e-9, atol=1e-9)
    # predict only
    ctx.trk_set_state(slots, g[f'{tag}_warp_m'], g[f'{tag}_warp_c'])
    ctx.trk_step_ops(2, slots, np.eye(3), zeros, np.zeros(n, np.uint8), ones)
    m, c = ctx.trk_get_state(slots)
    np.testing.assert_allclose(m, g[f'{tag}_pred_m'], rtol=1e-12, atol=1e-10)
    np.testing.assert_allclose(c, g[f'{tag}_pred_c'], rtol=1e-10, atol=1e-9)
    # KLT update only
    ctx.trk_set_state(slots, g[f'{tag}_pred_m'], g[f'{tag}_pred_c'])
    ctx.trk_step_ops(4, slots, np.eye(3), g[f'{tag}_klt'], np.ones(n, np.uint8), g[f'{tag}_mult'])
    m, c = ctx.trk_get_state(slots)
    np.testing.assert_allclose(m, g[f'{tag}_klt_m'], rtol=1e-11, atol=1e-9)
    np.testing.assert_allclose(c, g[f'{tag}_klt_c'], rtol=1e-9, atol=1e-8)
    # detector update
    ctx.trk_set_state(slots, g[f'{tag}_klt_m'], g[f'{tag}_klt_c'])
    tlbr, lost = ctx.trk_update_det(slots, g[f'{tag}_det'])
    m, c = ctx.trk_get_state(slots)
    np.testing.assert_allclose(m, g[f'{tag}_det_m'], rtol=1e-11, atol=1e-9)
    np.testing.assert_allclose(c, g[f'{tag}_det_c'], rtol=1e-9, atol=1e-8)
    np.testing.assert_array_equal(tlbr, np.rint(g[f'{tag}_det_m'][:, :4]))


@pytest.mark.parametrize('n', [1, 3, 4, 5, 50, 300])
def test_fused_step_vs_oracle(ctx, n):
    rng = np.random.default_rng(100 + n)
    dt = 1 / 30.
    configure(ctx, dt)
    p = o.KFParams(dt)
    tl = np.stack([rng.uniform(-50, 1800, n), rng.uniform(-50, 1000, n)], 1)
    boxes = np.rint(np.concatenate([tl, tl + rng.uniform(20, 300, (n, 2))], 1))
    slots = rng.permutation(np.arange(1, 2 * n + 1))[:n]
    ctx.trk_create(slots, boxes)
    m, c = o.kf_create(p, boxes)
    H = np.eye(3) + rng.normal(0, 1e-3, (3, 3)); H[:2, 2] += rng.normal(0, 3, 2)
    H[2, :2] = rng.normal(0, 1e-6, 2); H[2, 2] = 1.
    for it in range(4):
        has = rng.random(n) < 0.7
        klt = np.rint(m[:, :4] + rng.normal(0, 3, (n, 4)))
        mult = rng.uniform(1, 6, n)
        tlbr, lost = ctx.trk_step(slots, H, klt, has, mult)
        m, c = o.kf_warp(m, c, H)
        m, c = o.kf_predict(p, m, c)
        mu, cu = o.kf_update(p, m, c, klt, 'flow', mult)
        m = np.where(has[:, None], mu, m)
        c = np.where(has[:, None, None], cu, c)
        gm, gc = ctx.trk_get_state(slots)
        np.testing.assert_allclose(gm, m, rtol=1e-11, atol=1e-9)
        np.testing.assert_allclose(gc, c, rtol=1e-9, atol=1e-8)
        np.testing.assert_array_equal(tlbr, np.rint(m[:, :4]))
        exp_lost = o.ios(np.rint(m[:, :4]), np.array([0., 0., 1919., 1079.])) < 0.5
        np.testing.assert_array_equal(lost, exp_lost)
        # continue from the device state so that round-off does not accumulate in the comparison
        m, c = gm, gc


def test_empty_batch(ctx):
    configure(ctx, 1 / 30.)
    tlbr, lost = ctx.trk_step([], np.eye(3), np.zeros((0, 4)), [], [])
    assert tlbr.shape == (0, 4) and lost.shape == (0,)
    ctx.trk_create([], np.zeros((0, 4)))
