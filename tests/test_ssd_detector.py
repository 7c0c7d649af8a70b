"""SSDDetector host stages (fastmot_amd/detector.py; next row n4 of SURVEY section 8f) against golden vectors
produced by the reference's own functions (oracle/make_golden_ssd.py: detector.py:122-217 exec'd unmodified):
tile layout, tile normalisation, engine-output filtering, cross-tile merging.  The SSD networks themselves
cannot be built here (models/ssd.py), so inference is a callable replaying the golden engine output."""
from pathlib import Path

import numpy as np
import pytest

from fastmot_amd.detector import SSDDetector, DET_DTYPE

G = np.load(Path(__file__).parent / 'golden' / 'ssd_kat.npz')


def make(tag, backend=None):
    gx, gy, overlap, w, h, topk, thresh, merge, max_area = G[f'{tag}_params']
    det = SSDDetector((int(w), int(h)), tuple(int(m) for m in G[f'{tag}_mask']), model='SSDInceptionV2',
                      tile_overlap=float(overlap), tiling_grid=(int(gx), int(gy)), conf_thresh=float(thresh),
                      merge_thresh=float(merge), max_area=int(max_area),
                      backend=backend or (lambda batch: G[f'{tag}_det_out']))
    det.model = type('M', (det.model,), {'TOPK': int(topk)})
    return det


@pytest.mark.parametrize('tag', ['a', 'b', 'c'])
def test_tiles_filter_merge_match_reference(tag):
    det = make(tag)
    np.testing.assert_array_equal(det.tiles, G[f'{tag}_tiles'])
    assert tuple(det.tiling_region_sz) == tuple(G[f'{tag}_region'])
    dets, tile_ids = det.filter_dets(G[f'{tag}_det_out'], det.tiles, det.model.TOPK, det.label_mask, det.max_area,
                                     det.conf_thresh, det.scale_factor)
    np.testing.assert_array_equal(dets.tlbr.reshape(-1, 4), G[f'{tag}_flt_tlbr'].reshape(-1, 4))
    np.testing.assert_array_equal(dets.label, G[f'{tag}_flt_label'])
    np.testing.assert_array_equal(dets.conf, G[f'{tag}_flt_conf'])
    np.testing.assert_array_equal(tile_ids, G[f'{tag}_flt_tile'])
    merged = det.merge_dets(dets, tile_ids, det.batch_size, det.merge_thresh)
    assert merged.dtype == DET_DTYPE
    np.testing.assert_array_equal(merged.tlbr.reshape(-1, 4), G[f'{tag}_mrg_tlbr'])
    np.testing.assert_array_equal(merged.label, G[f'{tag}_mrg_label'])
    np.testing.assert_array_equal(merged.conf, G[f'{tag}_mrg_conf'])
    assert (np.diff(merged.label) >= 0).all()


def test_detect_async_postprocess_protocol():
    """frame -> resize to the tiling region -> tiles -> callable -> detections, through the two-phase API."""
    seen = {}

    def backend(batch):
        seen['shape'], seen['range'] = batch.shape, (float(batch.min()), float(batch.max()))
        return G['a_det_out']
    det = make('a', backend)
    frame = np.random.default_rng(1).integers(0, 256, (1080, 1920, 3), dtype=np.uint8)
    out = det(frame)
    assert seen['shape'] == (8, 3, 300, 300) and -1.0 <= seen['range'][0] < -0.9 and 0.9 < seen['range'][1] <= 1.0
    np.testing.assert_array_equal(out.tlbr.reshape(-1, 4), G['a_mrg_tlbr'])
    with pytest.raises(AssertionError):
        det.postprocess()                       # nothing in flight
    with pytest.raises(NotImplementedError):
        SSDDetector((1920, 1080), (1,))         # no network source here: explains what is missing
    with pytest.raises(ValueError):
        SSDDetector((1920, 1080), (91,), backend=backend)


def test_tile_normalisation_matches_reference():
    out = np.empty((2, 3, 300, 300), np.float32)
    SSDDetector.normalize(G['n_frame'], G['n_tiles'], out)
    np.testing.assert_array_equal(out[:, :, :4, :4], G['n_out_corner'])
    np.testing.assert_array_equal(out[:, :, -1, -3:], G['n_out_last'])
    np.testing.assert_allclose(out.sum(axis=(2, 3)).astype(np.float64), G['n_out_sum'], rtol=1e-6)
