"""The detector's post-network chain at the BASELINE sizes, through the C ABI, against the oracle
(oracle/detector_check.py): preprocess -> [HIP network] -> decode -> candidate sort -> DIoU-NMS -> box filters on
~1500 scripted candidates in the dense-suppression regime the benchmark times.

  config[1]  1080p -> YOLOv4 608x608, 80 classes, 3 heads x 3 anchors, classic (sigmoid / exp) decode
  config[2]  1080p -> YOLOv4-CSP 640x640, 80 classes, NEW_COORDS decode, letterbox on (the descriptor) and off
  config[4]  4K    -> YOLOv4-P6 1280x1280, 80 classes, 4 heads x 4 anchors, candidates of three classes

Reference: plugins/yolo_layer.cu:127-230, detector.py:289-365, utils/rect.py:199-244.  Bars (written in
detector_check.check): uint8 input pixels identical; candidate rows rel 5e-6 (fast-exp intrinsics; NEW_COORDS rows
carry no transcendental and must meet the same bar); NMS / final filter on the device's candidates bit-identical;
the whole chain against the oracle's own decode: same detections, boxes +-1 px, confidences rel 5e-6."""
import numpy as np
import pytest

import detector_check
from fastmot_amd.detector import YOLODetector
from fastmot_amd.models import YOLO
from synthetic import SyntheticVideo, scripted_head_weights

pytestmark = pytest.mark.gpu


class YOLOv4CSP_640_Stretch(YOLO.get_model('YOLOv4CSP_640')):
    """config[2]'s network with the letterbox switched off (the frame is stretched to 640x640)."""
    LETTERBOX = False


CASES = {
    'config1_yolov4_608': dict(size=(1920, 1080), model='YOLOv4_608', labels=(1,), n_ids=50),
    'config2_csp640_letterbox': dict(size=(1920, 1080), model='YOLOv4CSP_640', labels=(1,), n_ids=50),
    'config2_csp640_stretch': dict(size=(1920, 1080), model='YOLOv4CSP_640_Stretch', labels=(1,), n_ids=50),
    'config4_p6_1280_4k_3cls': dict(size=(3840, 2160), model='YOLOv4P6_1280', labels=(0, 1, 2), n_ids=60),
}


@pytest.mark.parametrize('case', sorted(CASES))
def test_detector_chain_equals_oracle_at_benchmark_size(ctx, case):
    c = CASES[case]
    video = SyntheticVideo(c['size'], n_ids=c['n_ids'], n_frames=2, seed=100)
    ctx.frame_configure(*c['size'])
    label = c['labels'] if len(c['labels']) > 1 else c['labels'][0]
    weights = scripted_head_weights(c['size'], c['model'], label, video.frames[0], 1500)
    det = YOLODetector(c['size'], c['labels'], model=c['model'], conf_thresh=0.25, nms_thresh=0.5,
                       max_area=800000, min_aspect_ratio=1.2, max_candidates=8192, weights=weights)
    try:
        m = det.model
        if case.startswith('config4'):
            assert len(det.heads) == 4 and all(len(a) == 8 for a in m.ANCHORS)
        if case.startswith('config2'):
            assert m.NEW_COORDS and m.LETTERBOX == case.endswith('letterbox')
        for f in range(2):
            res, dets = detector_check.check(det, video.frames[f])
            print(case, f, res)
            assert 600 <= res['candidates'] <= 4000, res          # the regime the bench times (~1500)
            # (classic heads: ~1500 -> a handful, the dense-suppression regime; NEW_COORDS heads with random box logits
            # spread anchor-sized boxes over the grid: hundreds survive -- the scan kernel's other regime)
            assert 1 <= res['detections'] < res['candidates'], res
            if case.startswith('config1'):
                assert res['detections'] < res['candidates'] // 50, res
            if len(c['labels']) > 1:
                assert len(np.unique(dets.label)) > 1, 'candidates of several classes expected'
                assert (np.diff(dets.label) >= 0).all()
            assert res['preprocess_identical'], res
            assert res['candidate_set_ok'], res
            assert res['decode_ok'], res
            assert res['sorted_ok'], res
            assert res['nms_identical'], res
            assert res['chain_vs_oracle_decode_ok'], res
            assert res['detector_chain_identical']
            # DESIGN section 7 "known deviations": candidate ties are resolved deterministically where the reference's
            # quicksort leaves them undefined -- on this clip the detections do not depend on that order
            assert res['tie_order_invariant'], res
    finally:
        det.backend.close()
