"""GPU parity: association kernels (assoc.hip) through the C ABI.
  * oracle=reference: tests/golden/assoc_kat.npz (cdist/iou/occluded/fuse+gate/LAP/greedy from the
    reference source)
  * oracle=restated : np_oracle + scipy.optimize.linear_sum_assignment on seeded inputs incl.
    tie-heavy, rectangular, gated (1e5) and degenerate matrices.  Index results must be IDENTICAL."""
import numpy as np
import pytest

import np_oracle as o
from fastmot_amd import _lib

pytestmark = pytest.mark.gpu

KF_DEFAULT = dict(std_factor_acc=2.25, std_offset_acc=78.5, std_factor_det=(0.08, 0.08),
                  std_factor_klt=(0.14, 0.14), min_std_det=(4.0, 4.0), min_std_klt=(5.0, 5.0),
                  init_pos_weight=5, init_vel_weight=12, vel_coupling=0.6, vel_half_life=2)


def load_tracks(ctx, feats, counts=None):
    """Puts features into slots 1..n through the public feature path (first update = copy)."""
    n = len(feats)
    slots = np.arange(1, n + 1)
    ctx.feat_reset(slots)
    ctx.emb_upload(feats.astype(np.float32))
    ctx.feat_update(slots, np.arange(n))
    return slots


@pytest.mark.parametrize('tag', ['a', 'b', 'c', 'd'])
def test_golden_pairwise_and_stage(ctx, golden_dir, tag):
    g = np.load(golden_dir / 'assoc_kat.npz')
    ctx.kf_configure(1 / 30., **KF_DEFAULT)
    XA, XB = g[f'{tag}_XA'], g[f'{tag}_XB']
    ta, db = g[f'{tag}_ta'], g[f'{tag}_db']
    nt, nd = len(XA), len(XB)
    slots = load_tracks(ctx, XA)
    ctx.trk_create(slots, ta)
    occ = g[f'{tag}_occ']
    np.testing.assert_array_equal(ctx.find_occluded(db, 0.7), occ)
    np.testing.assert_array_equal(ctx.find_occluded(db, 0.3), g[f'{tag}_occ3'])
    for metric, key, tol in ((_lib.METRIC_COSINE, 'cos', 2e-7), (_lib.METRIC_EUCLIDEAN, 'euc', 1e-12)):
        ctx.emb_upload(XB)
        ctx.assoc_prepare(metric, slots, ta, g[f'{tag}_tlab'], db, g[f'{tag}_dlab'], occ)
        feat, maha, iou = ctx.assoc_get_pairwise(nt, nd)
        ref = g[f'{tag}_{key}']
        mask = g[f'{tag}_mask']
        np.testing.assert_allclose(feat[~mask], ref[~mask], rtol=0, atol=tol)
        np.testing.assert_array_equal(iou, g[f'{tag}_iou'])
        # restated oracle (Numba typing): tight tolerance
        exp = o.cdist(XA.astype(np.float64), XB, 'cosine' if key == 'cos' else 'euclidean')
        np.testing.assert_allclose(feat, exp, rtol=0, atol=1e-13)
        p = o.KFParams(1 / 30.)
        m, c = o.kf_create(p, ta)
        np.testing.assert_allclose(maha, o.kf_maha(p, m, c, db), rtol=1e-10)
    # IoU stage cost + LAP + greedy vs reference outputs
    rows, cols = np.arange(nt), np.arange(nd)
    m_rows, m_cols, gated, cost = ctx.assoc_stage(_lib.STAGE_IOU, _lib.SOLVER_GREEDY, rows, cols,
                                                  max_cost=0.8, want_cost=True)
    exp_cost = o.gate_cost(g[f'{tag}_iou'], g[f'{tag}_tlab'], g[f'{tag}_dlab'], 0.8)
    np.testing.assert_array_equal(cost, exp_cost)
    em, _, _ = o.greedy_match(exp_cost, 0.8)
    assert list(zip(m_rows.tolist(), m_cols.tolist())) == em
    # golden LAP on the reference's fused+gated cost matrix
    r, c = ctx.lap(g[f'{tag}_cost'])
    er, ec = o.lsa(g[f'{tag}_cost'])
    np.testing.assert_array_equal(r, er)
    np.testing.assert_array_equal(c, ec)
    m, ur, uc = o.assignment_matches(g[f'{tag}_cost'], r, c)
    assert [(100 + a, b) for a, b in m] == [tuple(x) for x in g[f'{tag}_lap_m'].tolist()]
    assert [100 + a for a in ur] == g[f'{tag}_lap_ut'].tolist() and uc == g[f'{tag}_lap_ud'].tolist()
    gr, gc = ctx.greedy(g[f'{tag}_iou'], 0.8)
    assert [(100 + a, b) for a, b in zip(gr.tolist(), gc.tolist())] == [tuple(x) for x in g[f'{tag}_gr_m'].tolist()]


def test_matching_stage_vs_oracle(ctx):
    rng = np.random.default_rng(42)
    dt = 1 / 30.
    ctx.kf_configure(dt, **KF_DEFAULT)
    p = o.KFParams(dt)
    for nt, nd, metric in ((50, 50, 'euclidean'), (7, 90, 'cosine'), (120, 33, 'cosine'), (300, 300, 'euclidean')):
        tl = rng.uniform(0, 1500, (nt, 2))
        tb = np.rint(np.concatenate([tl, tl + rng.uniform(30, 200, (nt, 2))], 1))
        sel = rng.integers(0, nt, nd)
        db = np.rint(tb[sel] + rng.normal(0, 8, (nd, 4)))
        XA = rng.normal(0, 1, (nt, 512)).astype(np.float32)
        XA /= np.linalg.norm(XA, axis=1, keepdims=True)
        XB = (XA[sel] + rng.normal(0, 0.03, (nd, 512))).astype(np.float32)
        XB /= np.linalg.norm(XB, axis=1, keepdims=True)
        slots = load_tracks(ctx, XA)
        nofeat = rng.random(nt) < 0.1
        ctx.feat_reset(slots[nofeat])
        ctx.trk_create(slots, tb)
        m, c = o.kf_create(p, tb)
        tlab, dlab = rng.integers(0, 2, nt), rng.integers(0, 2, nd)
        occ = o.find_occluded(db, 0.7)
        ctx.emb_upload(XB)
        mid = _lib.METRIC_COSINE if metric == 'cosine' else _lib.METRIC_EUCLIDEAN
        ctx.assoc_prepare(mid, slots, tb, tlab, db, dlab, occ)
        rows = rng.permutation(nt)[:max(1, nt // 2)]
        cols = rng.permutation(nd)[:max(1, (2 * nd) // 3)]
        m_rows, m_cols, gated, cost = ctx.assoc_stage(_lib.STAGE_MATCHING, _lib.SOLVER_LAP, rows, cols,
                                                      motion_weight=0.2, max_cost=0.8, fill_val=0.9,
                                                      want_cost=True)
        mask = nofeat[:, None] | occ[None, :]
        fd = o.cdist(XA.astype(np.float64), XB, metric, mask, 0.9)
        exp = o.matching_cost(fd, o.kf_maha(p, m, c, db), tlab, dlab, 0.2, 0.8)[np.ix_(rows, cols)]
        big = exp >= 1e5
        np.testing.assert_array_equal(cost >= 1e5, big)
        np.testing.assert_allclose(cost[~big], exp[~big], rtol=0, atol=1e-11)
        er, ec = o.lsa(cost)
        np.testing.assert_array_equal(m_rows, er)
        np.testing.assert_array_equal(m_cols, ec)
        np.testing.assert_array_equal(gated, cost[er, ec] >= 1e5)


def test_reid_stage_f32_rows(ctx):
    rng = np.random.default_rng(8)
    ctx.kf_configure(1 / 30., **KF_DEFAULT)
    nt, nd = 20, 15
    XA = rng.normal(0, 1, (nt, 512)).astype(np.float32); XA /= np.linalg.norm(XA, axis=1, keepdims=True)
    XB = rng.normal(0, 1, (nd, 512)).astype(np.float32); XB /= np.linalg.norm(XB, axis=1, keepdims=True)
    XB[:5] = XA[3:8] + rng.normal(0, 0.01, (5, 512)).astype(np.float32)
    slots = load_tracks(ctx, XA)
    boxes = np.tile(np.array([10., 10., 60., 200.]), (nt, 1))
    ctx.trk_create(slots, boxes)
    for metric, mid in (('cosine', _lib.METRIC_COSINE), ('euclidean', _lib.METRIC_EUCLIDEAN)):
        ctx.emb_upload(XB)
        ctx.assoc_prepare(mid, slots, boxes, np.ones(nt, int), np.tile(boxes[0], (nd, 1)), np.ones(nd, int),
                          np.zeros(nd, bool), trk_feat_f32=np.ones(nt, bool))
        labels = np.ones(nt, int); labels[4] = 0
        m_rows, m_cols, _, cost = ctx.assoc_stage(_lib.STAGE_REID, _lib.SOLVER_GREEDY, np.arange(nt), np.arange(nd),
                                                  max_cost=0.6, row_labels=labels, want_cost=True)
        exp = o.gate_cost(o.cdist(XA, XB, metric), labels, np.ones(nd, int))
        np.testing.assert_allclose(cost, exp, rtol=0, atol=1e-13)
        em, _, _ = o.greedy_match(exp, 0.6)
        assert list(zip(m_rows.tolist(), m_cols.tolist())) == em


def _check_lap(ctx, cost):
    r, c = ctx.lap(cost)
    er, ec = o.lsa(cost)
    np.testing.assert_array_equal(r, er)
    np.testing.assert_array_equal(c, ec)


@pytest.mark.parametrize('host_lap_elems', [262144, 0])
def test_lap_matches_scipy_exactly(ctx, host_lap_elems):
    """Host solver for small matrices (default) and the device kernels (lap64_kernel: n <= 64 in
    registers, lap_kernel: LDS / global work arrays) all reproduce SciPy's (rows, cols) exactly."""
    ctx.set_option('host_lap_elems', host_lap_elems)
    try:
        _lap_cases(ctx)
    finally:
        ctx.set_option('host_lap_elems', 262144)


def _lap_cases(ctx):
    rng = np.random.default_rng(0)
    for trial in range(200):
        nr, nc = rng.integers(1, 40, 2)
        kind = trial % 4
        if kind == 0:
            cost = rng.uniform(0, 1, (nr, nc))
        elif kind == 1:
            cost = rng.integers(0, 4, (nr, nc)).astype(float)          # massive ties
        elif kind == 2:
            cost = np.round(rng.uniform(0, 1, (nr, nc)), 1)
            cost[rng.random((nr, nc)) < 0.5] = 1e5                       # gated entries
        else:
            cost = np.full((nr, nc), 1e5)                                # everything gated
            cost[rng.random((nr, nc)) < 0.1] = 0.3
        _check_lap(ctx, cost)
    # register-resident kernel boundary (n <= 64) with massive ties
    for shape in ((64, 64), (64, 30), (30, 64), (63, 64), (64, 65), (65, 64), (50, 50), (64, 1), (1, 64)):
        for ties in (False, True):
            cost = rng.integers(0, 3, shape).astype(float) if ties else rng.uniform(0, 1, shape)
            _check_lap(ctx, cost)
    for shape in ((1, 1), (1, 300), (300, 1), (128, 128), (300, 300), (150, 420), (420, 150), (700, 700)):
        cost = rng.uniform(0, 1, shape)
        cost[rng.random(shape) < 0.3] = 1e5
        _check_lap(ctx, cost)


def test_greedy_matches_reference_semantics(ctx):
    rng = np.random.default_rng(1)
    for trial in range(100):
        nr, nc = rng.integers(1, 30, 2)
        cost = np.round(rng.uniform(0, 1, (nr, nc)), 1 if trial % 2 else 6)
        thr = float(rng.uniform(0.1, 0.9))
        r, c = ctx.greedy(cost, thr)
        em, _, _ = o.greedy_match(cost, thr)
        assert list(zip(r.tolist(), c.tolist())) == em
    cost = rng.uniform(0, 1, (300, 280))
    r, c = ctx.greedy(cost, 0.5)
    em, _, _ = o.greedy_match(cost, 0.5)
    assert list(zip(r.tolist(), c.tolist())) == em


def test_average_feature_kernel(ctx):
    rng = np.random.default_rng(3)
    slots = np.array([5, 9, 2])
    ctx.feat_reset(slots)
    state = {int(s): None for s in slots}
    for step in range(1, 6):
        embs = rng.normal(0, 1, (3, 512)).astype(np.float32)
        embs /= np.linalg.norm(embs, axis=1, keepdims=True)
        ctx.emb_upload(embs)
        ctx.feat_update(slots, [2, 0, 1])
        for s, row in zip(slots, [2, 0, 1]):
            if state[int(s)] is None:
                state[int(s)] = (embs[row].copy(), embs[row].copy())
            else:
                state[int(s)] = o.average_feature(state[int(s)][0], embs[row], step)
            fsum, avg, cnt = ctx.feat_get(int(s))
            assert cnt == step
            np.testing.assert_allclose(fsum, state[int(s)][0], rtol=0, atol=1e-6)
            np.testing.assert_allclose(avg, state[int(s)][1], rtol=0, atol=1e-6)
    ctx.feat_merge(5, 9)
    fsum, avg, cnt = ctx.feat_get(5)
    s2, a2 = o.average_feature(state[5][0], state[9][0], 10)
    assert cnt == 10
    np.testing.assert_allclose(avg, a2, atol=1e-6)
