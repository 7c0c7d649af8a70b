"""GPU end to end: MOT.step on a synthetic video (frames -> detector net -> injected detections ->
KLT -> OSNet -> Kalman -> association).  Checks the API contract of mot.py:103-168 and that the
identities of the synthetic objects are held (one track id per object over the clip)."""
from types import SimpleNamespace

import numpy as np
import pytest

import scenes

pytestmark = pytest.mark.gpu


class TinyDet:
    pass


def build_mot(size, video, skip):
    import fastmot_amd.mot as mot_mod
    from fastmot_amd.models import YOLO
    from synthetic import InjectedYOLODetector

    class BenchTiny(YOLO):
        NUM_CLASSES = 2
        INPUT_SHAPE = (3, 160, 288)
        LAYER_FACTORS = [8, 16, 32]
        SCALES = [1.2, 1.1, 1.05]
        ANCHORS = [[4, 7, 8, 15, 12, 30], [18, 40, 25, 60, 30, 80], [40, 90, 60, 70, 80, 95]]
    kw = scenes.tracker_kwargs()
    mot_mod.YOLODetector = InjectedYOLODetector
    try:
        mot = mot_mod.MOT(size, detector_type='YOLO', detector_frame_skip=skip, class_ids=(1,),
                          yolo_detector_cfg=SimpleNamespace(model='BenchTiny', conf_thresh=0.25, nms_thresh=0.5,
                                                            max_area=800000, min_aspect_ratio=1.2),
                          feature_extractor_cfgs=(SimpleNamespace(model='OSNet025', batch_size=16),),
                          tracker_cfg=SimpleNamespace(**kw))
    finally:
        from fastmot_amd.detector import YOLODetector
        mot_mod.YOLODetector = YOLODetector
    mot.detector.bind_video(video)
    return mot


@pytest.mark.parametrize('skip', [1, 3])
def test_mot_step_holds_identities(ctx, skip):
    from synthetic import SyntheticVideo
    from fastmot_amd import Track
    size = (960, 540)
    video = SyntheticVideo(size, n_ids=12, n_frames=24, seed=3)
    mot = build_mot(size, video, 1)
    mot.detector_frame_skip = skip
    # injected detections are consumed once per detector frame -> index by frame
    mot.detector._video = video
    Track._count = 0
    mot.reset(1 / 30.)
    ids_per_obj = [set() for _ in range(video.n_ids)]
    for f in range(video.n_frames):
        mot.detector._frame_idx = f
        mot.step(video.frames[f])
        assert mot.frame_count == f + 1
        vis = list(mot.visible_tracks())
        for t in vis:
            c = (t.tlbr[:2] + t.tlbr[2:]) / 2
            gt = video.gt[f]
            inside = (gt[:, 0] <= c[0]) & (c[0] <= gt[:, 2]) & (gt[:, 1] <= c[1]) & (c[1] <= gt[:, 3])
            if inside.sum() == 1:
                ids_per_obj[int(np.flatnonzero(inside)[0])].add(t.trk_id)
        if f >= skip:
            assert len(vis) >= video.n_ids - 3
    # every object is followed, and almost all by a single identity
    assert all(len(s) >= 1 for s in ids_per_obj)
    assert sum(len(s) == 1 for s in ids_per_obj) >= video.n_ids - 2
    assert mot.tracker.homography is not None


@pytest.mark.parametrize('skip,resident', [(1, False), (2, False), (1, True)])
def test_next_frame_prefetch_changes_nothing(ctx, skip, resident):
    """MOT.step(frame, next_frame): the detector network of frame t+1 runs during frame t's ReID /
    association stages.  Tracks (ids, boxes, lifecycle) are bit-identical to strictly sequential steps, for
    uploaded host frames (second upload slot, promoted without re-upload) and for resident ring frames."""
    from synthetic import SyntheticVideo
    from fastmot_amd.detector import DeviceFrame
    from fastmot_amd import Track
    size = (960, 540)
    video = SyntheticVideo(size, n_ids=10, n_frames=16, seed=11)
    if resident:
        ctx.frame_configure(size[0], size[1], video.n_frames)
        for i, fr in enumerate(video.frames):
            ctx.frame_ring_store(i, fr)
        frames = [DeviceFrame(i) for i in range(video.n_frames)]
    else:
        frames = video.frames
    runs = []
    for prefetch in (False, True):
        mot = build_mot(size, video, skip)
        Track._count = 0
        mot.reset(1 / 30.)
        det_frame = 0
        rows = []
        for f in range(video.n_frames):
            mot.detector._frame_idx = f
            nxt = frames[f + 1] if prefetch and f + 1 < video.n_frames else None
            mot.step(frames[f], next_frame=nxt)
            rows.append([(t.trk_id, tuple(t.tlbr), t.confirmed, t.active, t.age, t.hits)
                         for t in mot.tracker.tracks.values()])
        runs.append(rows)
        mot.tracker._clear_tracks()
    assert runs[0] == runs[1]
    assert len(runs[0][-1]) >= 8


def test_native_prediction_worker_equals_python_thread(ctx, monkeypatch):
    """The KLT + Kalman chain on the library's worker thread (fm_track_predict_async: marshal on the main thread,
    fm_flow_predict + fm_trk_step in C, scatter after the join) gives the same tracks as Flow.predict +
    MultiTracker.apply_kalman on a second Python thread, frame by frame."""
    import fastmot_amd.mot as mot_mod
    from synthetic import SyntheticVideo
    from fastmot_amd import Track
    size = (960, 540)
    video = SyntheticVideo(size, n_ids=10, n_frames=16, seed=11)
    runs = []
    for native in (True, False):
        monkeypatch.setattr(mot_mod, '_NATIVE_FLOW', native)
        mot = build_mot(size, video, 2)                 # detector frames and tracker-only frames alternate
        Track._count = 0
        mot.reset(1 / 30.)
        rows = []
        for f in range(video.n_frames):
            mot.detector._frame_idx = f
            mot.step(video.frames[f])
            rows.append([(t.trk_id, tuple(t.tlbr), t.confirmed, t.active, t.age, t.hits, len(t.keypoints),
                          float(t.inlier_ratio)) for t in mot.tracker.tracks.values()])
        runs.append(rows)
        mot.tracker._clear_tracks()
    assert runs[0] == runs[1]
    assert len(runs[0][-1]) >= 8


def test_pipeline_is_deterministic(ctx):
    """The two-thread, four-stream pipeline gives the same tracks (ids, boxes, keypoint counts) on every run
    of the same clip (scripts/stress_determinism.py is the long version of this check)."""
    from synthetic import SyntheticVideo
    from fastmot_amd import Track
    size = (960, 540)
    video = SyntheticVideo(size, n_ids=10, n_frames=12, seed=11)
    runs = []
    for _ in range(4):
        mot = build_mot(size, video, 1)
        Track._count = 0
        mot.reset(1 / 30.)
        rows = []
        for f in range(video.n_frames):
            mot.detector._frame_idx = f
            mot.step(video.frames[f], next_frame=video.frames[f + 1] if f + 1 < video.n_frames else None)
            rows.append([(t.trk_id, tuple(t.tlbr), t.age, t.hits, len(t.keypoints)) for t in mot.tracker.tracks.values()])
        runs.append(rows)
        mot.tracker._clear_tracks()
    assert all(r == runs[0] for r in runs[1:])


def test_draw_overlays_do_not_change_tracking(ctx):
    """MOT(draw=True) renders the overlays of visualizer_cfg onto the caller's frame in place after each step
    (mot.py:166-167); the tracks are the same as without drawing, the frames differ from the originals exactly
    where something was drawn, and device-resident frames are rejected for drawing."""
    from synthetic import SyntheticVideo
    from fastmot_amd.utils.visualization import Visualizer, get_color
    from fastmot_amd.detector import DeviceFrame
    from fastmot_amd import Track
    size = (960, 540)
    video = SyntheticVideo(size, n_ids=8, n_frames=10, seed=5)
    runs = []
    for draw in (False, True):
        mot = build_mot(size, video, 2)
        mot.draw = draw
        # (no detection / KLT boxes: they are drawn after, and mostly on top of, the track boxes checked below)
        mot.visualizer = Visualizer(draw_obj_flow=True, draw_bg_flow=True, draw_trajectory=True, draw_covariance=True)
        Track._count = 0
        mot.reset(1 / 30.)
        rows, seen_visible = [], False
        for f in range(video.n_frames):
            mot.detector._frame_idx = f
            frame = video.frames[f].copy()
            mot.step(frame, next_frame=video.frames[f + 1] if f + 1 < video.n_frames else None)
            rows.append([(t.trk_id, tuple(t.tlbr), t.age, t.hits) for t in mot.tracker.tracks.values()])
            changed = (frame != video.frames[f]).any(axis=2)
            if not draw:
                assert not changed.any()
            else:
                vis = list(mot.visible_tracks())
                seen_visible = seen_visible or bool(vis)
                assert changed.sum() > 50 * len(vis)
                for t in vis:                                   # bottom edge of every visible track's box
                    x0, y0, x1, y1 = np.clip(t.tlbr.astype(int), 0, [size[0] - 1, size[1] - 1] * 2)
                    if y1 < size[1] - 1 and x1 - x0 > 8:
                        assert (frame[y1, x0 + 2:x1 - 1] == get_color(t.trk_id)).all(axis=1).mean() > 0.7
        runs.append(rows)
        if draw:
            assert seen_visible
            with pytest.raises(TypeError):
                ctx.frame_configure(size[0], size[1], 2)
                ctx.frame_ring_store(0, video.frames[0])
                mot.step(DeviceFrame(0))
        mot.tracker._clear_tracks()
    assert runs[0] == runs[1]


def test_stage_trace_orders_a_pipelined_step(ctx):
    """fm_trace_start / fm_trace_read (include/fastmot_hip.h): one timed event per stage boundary on the stage's own
    stream.  In a pipelined run every detector pass shows inputs-ready <= network-done <= decode-done, its
    post-processing begins after the decode, the ReID network of a frame begins after that frame's detections left the
    post-processing, and reading the trace disarms it (tracks are unaffected: same ids as an untraced run)."""
    from synthetic import SyntheticVideo
    from fastmot_amd import Track
    size = (960, 540)
    video = SyntheticVideo(size, n_ids=10, n_frames=16, seed=11)
    ids = []
    for traced in (False, True):
        mot = build_mot(size, video, 1)
        Track._count = 0
        mot.reset(1 / 30.)
        if traced:
            zero_ns = ctx.trace_start(4096)
            assert zero_ns > 0
        for f in range(video.n_frames):
            mot.detector._frame_idx = f
            nxt = video.frames[f + 1] if f + 1 < video.n_frames else None
            mot.step(video.frames[f], next_frame=nxt)
        ids.append(sorted(t.trk_id for t in mot.tracker.tracks.values()))
        if traced:
            tags, ms = ctx.trace_read()
            by = {t: ms[tags == t] for t in np.unique(tags)}
            n = video.n_frames
            assert all(len(by[t]) == n for t in (11, 12, 13)), {t: len(v) for t, v in by.items()}
            assert len(by[30]) == len(by[31]) == n - 1                  # one copy per prefetched frame
            assert len(by[20]) == len(by[21]) == n                      # every pass post-processed exactly once
            assert np.all(by[11] <= by[12]) and np.all(by[12] <= by[13])
            assert np.all(np.diff(by[11]) > 0)                           # passes run in order on their stream
            assert np.all(by[20] >= by[13]) and np.all(by[21] >= by[20])
            assert len(by[32]) == len(by[33]) and np.all(by[33] >= by[32])
            m = min(len(by[32]), n - 1)                                  # (frame 0 initialises the tracker: no ReID)
            assert m == n - 1 and np.all(by[32][-m:] >= by[21][-m:])
            tags2, _ = ctx.trace_read()                                  # disarmed: nothing recorded any more
            assert len(tags2) == 0
        mot.tracker._clear_tracks()
    assert ids[0] == ids[1] and len(ids[0]) >= 8
