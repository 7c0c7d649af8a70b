"""fm_cascade_host -- the host half of the one-call association cascade (csrc/assoc.hip cascade_run) -- against the
oracle's restatement of the reference's stage logic (np_oracle.matching_cost / gate_cost / lsa / assignment_matches /
greedy_match: utils/matching.py, tracker.py:198-248, pinned by the reference-generated goldens) on seeded pairwise
terms.  CPU only: the routine touches no device.  Sizes cross the table-growth boundaries of Numba's set (16, 32, 64
entries), costs carry exact ties, gated pairs, empty groups and label mismatches."""
import ctypes as C
import itertools

import numpy as np
import pytest

import np_oracle as o
from fastmot_amd import _lib


def reference_cascade(feat, maha, iou, has_feat, tlab, dlab, docc, groups, active, unconf, hist_rows, hist_labels,
                      det_conf, p):
    nD = feat.shape[1]

    def lap(cost, row_ids, col_ids):
        r, c = o.lsa(cost)
        m, ur, uc = o.assignment_matches(cost, r, c)
        return [(row_ids[a], col_ids[b]) for a, b in m], [row_ids[a] for a in ur], [col_ids[b] for b in uc]

    def matching(rows, cols):
        if not rows or not cols:
            return np.empty((len(rows), len(cols)))
        f = feat[np.ix_(rows, cols)].copy()
        empty = (~has_feat[rows])[:, None] | docc[cols][None, :]
        f[empty] = p['fill_val']
        return o.matching_cost(f, maha[np.ix_(rows, cols)], tlab[rows], dlab[cols], p['motion_weight'], p['max_assoc_cost'])

    def ioucost(rows, cols):
        if not rows or not cols:
            return np.empty((len(rows), len(cols)))
        return o.gate_cost(iou[np.ix_(rows, cols)], tlab[rows], dlab[cols], p['max_iou_cost'])

    matches1, u1 = [], []
    u_det = list(range(nD))
    for depth, rows in enumerate(groups):
        if len(u_det) == 0:
            u1.extend(itertools.chain.from_iterable(groups[depth:]))
            break
        if len(rows) == 0:
            continue
        m, ut, u_det = lap(matching(rows, u_det), rows, u_det)
        matches1 += m
        u1 += ut
    act = [r for r in u1 if active[r]]
    u1 = [r for r in u1 if not active[r]]
    matches2, u2, u_det = lap(ioucost(act, u_det), act, u_det)
    matches3, u3, u_det = lap(ioucost(unconf, u_det), unconf, u_det)
    u_det = [d for d in u_det if det_conf[d] >= p['conf_thresh']]
    valid = [d for d in u_det if not docc[d]]
    invalid = [d for d in u_det if docc[d]]
    if len(hist_rows) and len(valid):
        cost = o.gate_cost(feat[np.ix_(hist_rows, valid)], hist_labels, dlab[valid])
    else:
        cost = np.empty((len(hist_rows), len(valid)))
    gm, _, gu = o.greedy_match(cost, p['max_reid_cost'])
    return (matches1 + matches2 + matches3, u1 + u2 + u3, [(a, valid[b]) for a, b in gm], invalid + [valid[b] for b in gu])


def run_host(feat, maha, iou, has_feat, tlab, dlab, docc, groups, active, unconf, hist_rows, hist_labels, det_conf, p):
    lib = _lib.load()
    nT, nD = feat.shape
    flat = list(itertools.chain.from_iterable(groups))
    off = np.zeros(len(groups) + 1, np.int32)
    np.cumsum([len(g) for g in groups], out=off[1:])
    arrs = (off, np.asarray(flat, np.int32), np.asarray([active[r] for r in flat], np.uint8), np.asarray(unconf, np.int32),
            np.asarray(hist_rows, np.int32), np.asarray(hist_labels, np.int64), np.ascontiguousarray(det_conf, np.float64))
    cin = _lib.CascadeIn(len(groups), len(unconf), len(hist_rows), 0, *(a.__array_interface__['data'][0] for a in arrs),
                         p['motion_weight'], p['max_assoc_cost'], p['fill_val'], p['max_iou_cost'], p['conf_thresh'],
                         p['max_reid_cost'])
    out = np.full(_lib.CASCADE_HEADER + 3 * (len(flat) + len(unconf)) + 3 * nD, -7, np.int32)
    ins = [np.ascontiguousarray(a, t) for a, t in ((feat, np.float64), (maha, np.float64), (iou, np.float64),
                                                    (has_feat, np.uint8), (tlab, np.int64), (dlab, np.int64), (docc, np.uint8))]
    rc = lib.fm_cascade_host(C.c_int(nT), C.c_int(nD), *(_lib._ptr(a) for a in ins), C.byref(cin), _lib._ptr(out),
                             C.c_int(len(out)))
    _lib.check(rc)
    w = out.tolist()
    n1, n2, n3, a1, a2, a3, n_reid, n_inv, n_rest, used = w[:10]
    pos = _lib.CASCADE_HEADER
    q = pos + 2 * (n1 + n2 + n3)
    matches = list(zip(w[pos:q:2], w[pos + 1:q:2]))
    pos, q = q, q + a1 + a2 + a3
    u_rows = w[pos:q]
    pos, q = q, q + 2 * n_reid
    reid = list(zip(w[pos:q:2], w[pos + 1:q:2]))
    new = w[q:q + n_inv + n_rest]
    assert used == q + n_inv + n_rest
    return matches, u_rows, reid, new


def make_case(rng, nT, nD, n_hist, quantise):
    feat = rng.uniform(0, 1.2, (nT, nD))
    maha = rng.uniform(0, 14, (nT, nD))
    iou = rng.uniform(0, 1, (nT, nD))
    if quantise:                                # exact ties: the solver's scan order and the greedy's first minimum decide
        feat, maha, iou = np.round(feat, 1), np.round(maha), np.round(iou, 1)
    has_feat = rng.random(nT) < 0.9
    n_lab = int(rng.integers(1, 4))
    tlab = rng.integers(0, n_lab, nT)
    dlab = rng.integers(0, n_lab, nD)
    docc = rng.random(nD) < 0.2
    det_conf = rng.uniform(0.3, 1.0, nD)
    rows = rng.permutation(nT)
    hist_rows = rows[:n_hist].tolist()
    rest = rows[n_hist:]
    n_unconf = int(rng.integers(0, max(1, len(rest) // 4 + 1)))
    unconf = rest[:n_unconf].tolist()
    conf = rest[n_unconf:]
    depth = rng.integers(0, 4, len(conf))
    if rng.random() < 0.3:
        depth[depth == 1] = 2                   # an empty depth group in the middle
    groups = [conf[depth == g].tolist() for g in range(4)]
    active = rng.random(nT) < 0.7
    hist_labels = rng.integers(0, n_lab, n_hist).tolist()
    p = dict(motion_weight=0.2, max_assoc_cost=float(rng.choice([0.8, 0.9, 0.5])), max_iou_cost=float(rng.choice([0.6, 0.3])),
             conf_thresh=0.5, max_reid_cost=float(rng.choice([0.6, 0.45, 0.2])))
    p['fill_val'] = min(p['max_assoc_cost'] + 0.1, 1.)
    return (feat, maha, iou, has_feat, tlab, dlab, docc, groups, active, unconf, hist_rows, hist_labels, det_conf, p)


@pytest.mark.parametrize('quantise', [False, True])
def test_cascade_host_equals_reference_logic(quantise):
    rng = np.random.default_rng(11 + quantise)
    sizes = [(1, 1), (3, 7), (7, 3), (8, 9), (9, 8), (16, 17), (17, 16), (20, 50), (50, 20), (33, 31), (50, 50), (64, 70),
             (70, 64), (90, 40), (40, 90), (5, 0 + 1), (130, 130)]
    n = 0
    for nT, nD in sizes:
        for rep in range(12 if nT * nD < 3000 else 4):
            case = make_case(rng, nT, nD, int(rng.integers(0, max(1, nT // 3) + 1)), quantise)
            ref = reference_cascade(*case)
            got = run_host(*case)
            assert got[0] == ref[0], (nT, nD, rep, 'matches')
            assert got[1] == ref[1], (nT, nD, rep, 'unmatched track order')
            assert got[2] == ref[2], (nT, nD, rep, 're-identification')
            assert got[3] == ref[3], (nT, nD, rep, 'new-track detections order')
            n += 1
    assert n > 150


def test_cascade_host_no_detections_left_after_first_group():
    """tracker.py:207-209: once no detection is left the remaining depth groups go to the unmatched list unsolved."""
    rng = np.random.default_rng(5)
    nT, nD = 12, 3
    case = list(make_case(rng, nT, nD, 0, False))
    case[0][:] = 0.1                             # everything matches in the first group
    case[1][:] = 1.0
    case[3][:] = True
    case[4][:] = 0
    case[5][:] = 0
    case[6][:] = False
    case[7] = [[0, 1, 2, 3], [4, 5], [6, 7, 8], [9, 10, 11]]
    case[9] = []
    case[10], case[11] = [], []
    ref = reference_cascade(*case)
    got = run_host(*case)
    assert got == ref and len(ref[0]) == 3 and len(ref[1]) == 9
