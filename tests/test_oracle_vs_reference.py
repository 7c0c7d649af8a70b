"""CPU, build container only: oracle restatement vs the live reference (skipped on the GPU box,
where /root/reference does not exist)."""
import numpy as np
import pytest

import np_oracle as o
import ref_shim

pytestmark = pytest.mark.skipif(not ref_shim.reference_available(), reason='/root/reference absent')


@pytest.fixture(scope='module')
def ref():
    ns = ref_shim.load_reference()
    yield ns
    ref_shim.unload_reference(ns)


def test_kalman_random(ref):
    rng = np.random.default_rng(3)
    kf = ref.kalman_filter.KalmanFilter()
    kf.reset_dt(1 / 25.)
    p = o.KFParams(1 / 25.)
    MeasType = ref.kalman_filter.MeasType
    n = 24
    tl = rng.uniform(0, 1000, (n, 2))
    boxes = np.rint(np.concatenate([tl, tl + rng.uniform(20, 300, (n, 2))], 1))
    H = np.eye(3) + rng.normal(0, 1e-3, (3, 3)); H[2, 2] = 1; H[2, :2] *= 1e-3
    m, c = o.kf_create(p, boxes)
    m[:, 4:] = rng.normal(0, 20, (n, 4))
    m2, c2 = o.kf_warp(m, c, H)
    m2, c2 = o.kf_predict(p, m2, c2)
    m2, c2 = o.kf_update(p, m2, c2, boxes + 2, 'flow', 3.0)
    for i in range(n):
        a, b = kf.warp(m[i], c[i], H)
        a, b = kf.predict(a, b)
        a, b = kf.update(a, b, boxes[i] + 2, MeasType.FLOW, 3.0)
        np.testing.assert_allclose(m2[i], a, rtol=1e-11, atol=1e-9)
        np.testing.assert_allclose(c2[i], b, rtol=1e-9, atol=1e-8)


def test_greedy_and_lap_random(ref):
    rng = np.random.default_rng(5)
    for _ in range(30):
        nr, nc = rng.integers(1, 12, 2)
        cost = np.round(rng.uniform(0, 1, (nr, nc)), 1)    # many ties
        cost[rng.random((nr, nc)) < 0.3] = 1e5
        rid, cid = list(range(10, 10 + nr)), list(range(nc))
        m, ur, uc = ref.matching.linear_assignment(cost, rid, cid)
        r, c = o.lsa(cost)
        m2, ur2, uc2 = o.assignment_matches(cost, r, c)
        assert [(10 + a, b) for a, b in m2] == [(int(a), int(b)) for a, b in m]
        assert [10 + a for a in ur2] == [int(x) for x in ur] and uc2 == [int(x) for x in uc]
        m, ur, uc = ref.matching.greedy_match(cost.copy(), rid, cid, 0.5)
        m2, ur2, uc2 = o.greedy_match(cost, 0.5)
        assert [(10 + a, b) for a, b in m2] == [(int(a), int(b)) for a, b in m]
        assert [10 + a for a in ur2] == [int(x) for x in ur] and uc2 == [int(x) for x in uc]


def test_average_feature(ref):
    rng = np.random.default_rng(9)
    af = ref.track.AverageFeature()
    s = a = None
    for k in range(1, 6):
        v = rng.normal(0, 1, 512).astype(np.float32)
        v /= np.linalg.norm(v)
        af.update(v)
        if s is None:
            s, a = v.copy(), v.copy()
        else:
            s, a = o.average_feature(s, v, k)
        np.testing.assert_allclose(a, af.avg, rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize('num_classes,new_coords', [(80, False), (2, False), (1, False), (80, True), (3, True)])
def test_yolo_decode_against_the_reference_kernels(num_classes, new_coords):
    """np_oracle.yolo_decode against the reference's own CalDetection / CalDetection_NewCoords (plugins/yolo_layer.cu:
    127-230), compiled for the host from the reference file by oracle/yolo_layer_ref.py (CUDA's __expf becomes expf):
    row order, output layout and class arg-max identical, the NEW_COORDS variant (no exponential in it) bit for bit,
    the sigmoid variant to the last bits of exp()."""
    import yolo_layer_ref
    if not yolo_layer_ref.available():
        pytest.skip('needs /root/reference')
    rng = np.random.default_rng(num_classes + 7 * new_coords)
    anchors = [12, 16, 19, 36, 40, 28]
    for (H, W), in_wh in (((19, 19), (608, 608)), ((20, 36), (1152, 640)), ((1, 1), (32, 32))):
        head = (rng.uniform(0, 1, ((5 + num_classes) * 3, H, W)) if new_coords
                else rng.normal(0, 2.5, ((5 + num_classes) * 3, H, W))).astype(np.float32)
        if num_classes > 1:
            head[5 + 1, 0, 0] = head[5, 0, 0]          # a tie between two class logits: the first one wins
        ref = yolo_layer_ref.decode(head, anchors, num_classes, in_wh, 1.05, new_coords)
        got = o.yolo_decode(head, anchors, num_classes, in_wh, 1.05, new_coords)
        assert ref.shape == got.shape == (3 * H * W, 7)
        np.testing.assert_array_equal(got[:, 5], ref[:, 5])
        if new_coords:
            np.testing.assert_array_equal(got, ref)
        else:
            np.testing.assert_allclose(got, ref, rtol=1e-6, atol=2e-6)
