"""a1 end to end: fastmot_amd.MOT.step (detector network -> injected detections -> real KLT on the frames ->
OSNet -> Kalman -> association; fastmot/mot.py:125-168, tracker.py:139-293, flow.py:135-264) against the CPU
oracle (oracle/cpu_tracker.OracleTracker + cv_oracle) on the SAME synthetic clip at the BASELINE sizes.

The oracle consumes the embeddings the HIP OSNet produced (the network itself is checked against PyTorch in
test_conv_gpu.py / test_fullsize_gpu.py), so every difference seen here comes from KLT / Kalman / association.
Bar: track IDs, dict order, rounded boxes, life-cycle counters, history, per-track keypoint counts and KLT
boxes IDENTICAL on every frame (the LK kernel is bit-identical to the restatement); homography to 2e-5 absolute."""
from types import SimpleNamespace

import pytest

import e2e_check
import scenes

pytestmark = pytest.mark.gpu

KLT_BOX_TOL_PX = 0.0       # KLT boxes are rounded (flow.py:273-280): identical
H_TOL = 2e-5               # homography entries (translations are tens of pixels): double-precision host code
                           # (Jacobi eigen-solver, LM) vs numpy's LAPACK; measured <= 1.2e-6


def build_mot(size, video, yolo='YOLOv4_608', reid='OSNet025', batch=64, labels=(1,)):
    import fastmot_amd.mot as mot_mod
    from fastmot_amd.detector import YOLODetector
    from synthetic import InjectedYOLODetector
    kw = scenes.tracker_kwargs()
    mot_mod.YOLODetector = InjectedYOLODetector
    try:
        mot = mot_mod.MOT(size, detector_type='YOLO', detector_frame_skip=1, class_ids=tuple(labels),
                          yolo_detector_cfg=SimpleNamespace(model=yolo, conf_thresh=0.25, nms_thresh=0.5,
                                                            max_area=800000, min_aspect_ratio=1.2,
                                                            max_candidates=8192),
                          feature_extractor_cfgs=tuple(SimpleNamespace(model=reid, batch_size=batch) for _ in labels),
                          tracker_cfg=SimpleNamespace(**kw))
    finally:
        mot_mod.YOLODetector = YOLODetector
    mot.detector.bind_video(video, labels=tuple(labels))
    return mot, kw


def check(summary):
    print(summary)
    assert summary['ids_identical'], summary['first_mismatch']
    assert summary['all_identical'], summary['first_mismatch']
    assert summary['klt_box_max_px'] <= KLT_BOX_TOL_PX, summary
    assert summary['H_max_abs'] <= H_TOL, summary


@pytest.mark.parametrize('skip,n_frames,prefetch', [(1, 32, True), (5, 41, False)])
def test_mot_step_equals_oracle_1080p_50(ctx, skip, n_frames, prefetch):
    """BASELINE config[1] (skip 1, with the next-frame prefetch bench.py uses) and a config[0]/[2]-style
    detector_frame_skip=5 run: 1920x1080, 50 objects, YOLOv4@608 + OSNet-x0.25."""
    from synthetic import SyntheticVideo
    size = (1920, 1080)
    video = SyntheticVideo(size, n_ids=50, n_frames=n_frames, seed=100)
    mot, kw = build_mot(size, video)
    hip, emb = e2e_check.hip_pass(mot, video, n_frames, skip, prefetch=prefetch)
    ora, _, _ = e2e_check.oracle_pass(size, mot.extractors[0].metric.lower(), kw, video, n_frames, skip, emb)
    summary = e2e_check.compare(hip, ora)
    mot.tracker._clear_tracks()
    assert summary['frames'] == n_frames and summary['max_tracks'] >= 45
    check(summary)


def test_mot_step_equals_oracle_small_long(ctx):
    """A longer clip at 960x540 (objects leave / re-enter, tracks get lost and re-identified)."""
    from synthetic import SyntheticVideo
    size = (960, 540)
    video = SyntheticVideo(size, n_ids=14, n_frames=90, seed=21)
    mot, kw = build_mot(size, video, batch=16)
    hip, emb = e2e_check.hip_pass(mot, video, 90, 2, prefetch=True)
    ora, _, _ = e2e_check.oracle_pass(size, mot.extractors[0].metric.lower(), kw, video, 90, 2, emb)
    summary = e2e_check.compare(hip, ora)
    mot.tracker._clear_tracks()
    check(summary)


# ---- the BASELINE configurations beyond config[1], checked against the COMPILED restatement (oracle/c_baseline.c:
# the cv_oracle routines in C, pinned to cv_oracle function by function by tests/test_c_baseline.py -- the numpy
# oracle runs the 4K / 300-object clip at 0.15 frames/s, the compiled one fits the test budget)
def _config_parity(size, n_ids, n_frames, skip, yolo, reid, labels, prefetch):
    import c_baseline
    from synthetic import SyntheticVideo
    video = SyntheticVideo(size, n_ids=n_ids, n_frames=n_frames, seed=100)
    mot, kw = build_mot(size, video, yolo=yolo, reid=reid, labels=labels)
    try:
        hip, emb = e2e_check.hip_pass(mot, video, n_frames, skip, prefetch=prefetch)
        lab = labels if len(labels) > 1 else None
        ora, _, done = e2e_check.oracle_pass(size, mot.extractors[0].metric.lower(), kw, video, n_frames, skip, emb,
                                             labels=lab, cv_impl=c_baseline)
        summary = e2e_check.compare(hip, ora)
    finally:
        mot.tracker._clear_tracks()
    assert done == n_frames and summary['frames'] == n_frames
    return summary


def test_mot_step_equals_oracle_config2_csp640_osnet10_skip5(ctx):
    """BASELINE config[2]: 1080p, 50 objects, YOLOv4-CSP@640 + OSNet-x1.0 (cosine metric), detector_frame_skip=5
    (KLT-heavy: four of five frames are MultiTracker.track, mot.py:160-163), 41 frames = 8 detector frames."""
    summary = _config_parity((1920, 1080), 50, 41, 5, 'YOLOv4CSP_640', 'OSNet10', (1,), True)
    assert summary['max_tracks'] >= 45
    check(summary)


def test_mot_step_equals_oracle_config4_4k_p6_300_multiclass(ctx):
    """BASELINE config[4]: 3840x2160, 300 objects of 3 classes, YOLOv4-P6@1280, one OSNet-x0.25 per class
    (_split_bboxes_by_cls, mot.py:180-189, with the reference's bisect quirk Q3), detector_frame_skip=1, 16 frames."""
    summary = _config_parity((3840, 2160), 300, 16, 1, 'YOLOv4P6_1280', 'OSNet025', (0, 1, 2), True)
    assert summary['max_tracks'] >= 280
    check(summary)
