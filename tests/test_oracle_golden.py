"""CPU: the numpy restatement (oracle/np_oracle.py) against the golden vectors generated from
the reference itself (oracle/make_golden.py) -- oracle=reference pins the oracle."""
import numpy as np
import pytest

import np_oracle as o


@pytest.fixture(scope='module')
def kat(golden_dir):
    return np.load(golden_dir / 'kalman_kat.npz')


@pytest.mark.parametrize('tag', ['dt30', 'dt12'])
def test_kalman_chain(kat, tag):
    p = o.KFParams(float(kat[f'{tag}_dt']))
    boxes, H = kat[f'{tag}_boxes'], kat[f'{tag}_H']
    m, c = o.kf_create(p, boxes)
    np.testing.assert_allclose(m, kat[f'{tag}_create_m'], rtol=0, atol=0)
    np.testing.assert_allclose(c, kat[f'{tag}_create_c'], rtol=1e-15)
    m, c = kat[f'{tag}_start_m'], kat[f'{tag}_start_c']
    m, c = o.kf_warp(m, c, H)
    np.testing.assert_allclose(m, kat[f'{tag}_warp_m'], rtol=1e-12, atol=1e-10)
    np.testing.assert_allclose(c, kat[f'{tag}_warp_c'], rtol=1e-9, atol=1e-9)
    m, c = o.kf_predict(p, kat[f'{tag}_warp_m'], kat[f'{tag}_warp_c'])
    np.testing.assert_allclose(m, kat[f'{tag}_pred_m'], rtol=1e-12, atol=1e-10)
    np.testing.assert_allclose(c, kat[f'{tag}_pred_c'], rtol=1e-10, atol=1e-9)
    m, c = o.kf_update(p, kat[f'{tag}_pred_m'], kat[f'{tag}_pred_c'], kat[f'{tag}_klt'], 'flow', kat[f'{tag}_mult'])
    np.testing.assert_allclose(m, kat[f'{tag}_klt_m'], rtol=1e-11, atol=1e-9)
    np.testing.assert_allclose(c, kat[f'{tag}_klt_c'], rtol=1e-9, atol=1e-8)
    d = o.kf_maha(p, kat[f'{tag}_klt_m'], kat[f'{tag}_klt_c'], kat[f'{tag}_det'])
    np.testing.assert_allclose(d, kat[f'{tag}_maha'], rtol=1e-9)
    m, c = o.kf_update(p, kat[f'{tag}_klt_m'], kat[f'{tag}_klt_c'], kat[f'{tag}_det'], 'detector')
    np.testing.assert_allclose(m, kat[f'{tag}_det_m'], rtol=1e-11, atol=1e-9)
    np.testing.assert_allclose(c, kat[f'{tag}_det_c'], rtol=1e-9, atol=1e-8)


@pytest.mark.parametrize('tag', ['a', 'b', 'c', 'd'])
def test_assoc(golden_dir, tag):
    g = np.load(golden_dir / 'assoc_kat.npz')
    XA, XB, mask = g[f'{tag}_XA'], g[f'{tag}_XB'], g[f'{tag}_mask']
    # XA as the f64 copy of f32 averages (tracker.py:321-326).  atol 2e-7: the shim accumulates the
    # f32 b_norm terms in f32 under NumPy 2 (see np_oracle.cdist docstring); euclidean is all-f64.
    XA = XA.astype(np.float64)
    np.testing.assert_allclose(o.cdist(XA, XB, 'cosine', mask, 0.9), g[f'{tag}_cos'], rtol=0, atol=2e-7)
    np.testing.assert_allclose(o.cdist(XA, XB, 'euclidean', mask, 0.9), g[f'{tag}_euc'], rtol=0, atol=1e-12)
    np.testing.assert_array_equal(o.iou_dist(g[f'{tag}_ta'], g[f'{tag}_db']), g[f'{tag}_iou'])
    np.testing.assert_array_equal(o.find_occluded(g[f'{tag}_db'], 0.7), g[f'{tag}_occ'])
    np.testing.assert_array_equal(o.find_occluded(g[f'{tag}_db'], 0.3), g[f'{tag}_occ3'])
    cost = o.matching_cost(g[f'{tag}_cos'], g[f'{tag}_maha'], g[f'{tag}_tlab'], g[f'{tag}_dlab'], 0.2, 0.8)
    np.testing.assert_array_equal(cost, g[f'{tag}_cost'])
    nt = cost.shape[0]
    r, c = o.lsa(cost)
    m, ur, uc = o.assignment_matches(cost, r, c)
    assert [(100 + a, b) for a, b in m] == [tuple(x) for x in g[f'{tag}_lap_m'].tolist()]
    assert [100 + a for a in ur] == g[f'{tag}_lap_ut'].tolist()
    assert uc == g[f'{tag}_lap_ud'].tolist()
    m, ur, uc = o.greedy_match(g[f'{tag}_iou'], 0.8)
    assert [(100 + a, b) for a, b in m] == [tuple(x) for x in g[f'{tag}_gr_m'].tolist()]
    assert [100 + a for a in ur] == g[f'{tag}_gr_ut'].tolist()
    assert uc == g[f'{tag}_gr_ud'].tolist()


@pytest.mark.parametrize('name', ['s20_skip5_euclid', 's50_skip1_cosine', 's50_skip2_euclid', 's8_flowfail',
                                  's16_blackout_confirm3', 's40_multiclass_reid'])
def test_tracker_scenes(golden_dir, name):
    """The restated CPU tracker (oracle/cpu_tracker.py) reproduces the reference MultiTracker's
    golden runs: identical track ids / order / rounded boxes / lifecycle on every frame."""
    import scenes
    import cpu_tracker
    g = np.load(golden_dir / f'tracker_{name}.npz')
    scene = scenes.Scene(name)
    tracker = cpu_tracker.OracleTracker(scene.size, scene.metric, **scenes.tracker_kwargs(name))
    records, final = scenes.run_scene(tracker, scene)
    out = scenes.pack_records(records, final)
    np.testing.assert_array_equal(out['tracks'], g['tracks'])
    np.testing.assert_array_equal(out['hist'], g['hist'])
    np.testing.assert_allclose(out['final_mean'], g['final_mean'], rtol=1e-9, atol=1e-7)


def test_restatement_against_the_jit_compiled_reference(golden_dir):
    """tests/golden/real_numba_floats.npz (oracle/pin_with_numba.py): the float arrays the reference produces when its
    @njit functions are compiled by a REAL Numba (0.54.1), where they differ from the de-jitted goldens.  The oracle's
    cosine distance follows Numba's typing (float64 accumulation of the float32 terms), not the shim's NumPy-2 weak-scalar
    artifact: it equals the real thing to the last bit or two, and the gated cost built on it exactly; everything Kalman
    is within the reassociation noise of fastmath (1e-10 absolute on states of order 1e3)."""
    rf = np.load(golden_dir / 'real_numba_floats.npz')
    g = np.load(golden_dir / 'assoc_kat.npz')
    for tag in 'abcd':
        XA, XB, mask = g[f'{tag}_XA'].astype(np.float64), g[f'{tag}_XB'], g[f'{tag}_mask']
        np.testing.assert_allclose(o.cdist(XA, XB, 'cosine', mask, 0.9), rf[f'assoc_kat:{tag}_cos'], rtol=0, atol=2e-15)
        np.testing.assert_allclose(o.cdist(XA, XB, 'euclidean', mask, 0.9), rf[f'assoc_kat:{tag}_euc'], rtol=0, atol=1e-12)
        cost = o.matching_cost(rf[f'assoc_kat:{tag}_cos'], g[f'{tag}_maha'], g[f'{tag}_tlab'], g[f'{tag}_dlab'], 0.2, 0.8)
        np.testing.assert_array_equal(cost, rf[f'assoc_kat:{tag}_cost'])
    for name in rf.files:
        stem, key = name.split(':')
        if stem == 'assoc_kat':
            continue
        committed = np.load(golden_dir / f'{stem}.npz')[key]
        np.testing.assert_allclose(committed, rf[name], rtol=1e-9, atol=1e-9)
    rec = __import__('json').loads((golden_dir / 'REAL_NUMBA_PIN.json').read_text())
    assert rec['set_order']['restatement_mismatches'] == 0
    assert all(v['status'] in ('identical', 'float differences only') for v in rec['goldens'].values())
    assert sum(v['status'] == 'identical' for v in rec['goldens'].values()) >= 3


def test_yolo_decode_against_reference_kernel_outputs(golden_dir):
    """tests/golden/yolo_decode_ref.npz: outputs of the reference's CalDetection / CalDetection_NewCoords compiled for the
    host from plugins/yolo_layer.cu (oracle/yolo_layer_ref.py --golden; __expf -> expf): same rows in the same order,
    class ids identical, NEW_COORDS bit for bit, the sigmoid / exp variant to the last bits of exp()."""
    g = np.load(golden_dir / 'yolo_decode_ref.npz')
    for k in range(int(g['n'])):
        nc, new, iw, ih, sxy = g[f'k{k}_params']
        got = o.yolo_decode(g[f'k{k}_head'], g[f'k{k}_anchors'], int(nc), (int(iw), int(ih)), float(sxy), bool(new))
        ref = g[f'k{k}_rows']
        np.testing.assert_array_equal(got[:, 5], ref[:, 5])
        if new:
            np.testing.assert_array_equal(got, ref)
        else:
            np.testing.assert_allclose(got, ref, rtol=1e-6, atol=2e-6)
