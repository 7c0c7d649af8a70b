"""TEST INFRASTRUCTURE ONLY -- end-to-end checker of MOT.step (fastmot/mot.py:125-168) with the real
KLT in the loop: the HIP pipeline (fastmot_amd.MOT) and the CPU restatement (cpu_tracker.OracleTracker +
cv_oracle) are driven over the SAME frames with the SAME injected detections and the embeddings the HIP
OSNet produced, and compared frame by frame.

Used by tests/test_e2e_parity_gpu.py and by the `cpu_baseline` leg of bench.py (which times the oracle
pass and reports what the comparison found next to it).  Never imported by the product package."""
import time

import numpy as np

import cpu_tracker


def frame_record(tracker):
    """What defines the tracker's observable state after a step (reference MultiTracker attributes)."""
    tracks = tracker.tracks
    return dict(
        ids=list(tracks.keys()),
        tlbr=np.array([t.tlbr for t in tracks.values()], float).reshape(-1, 4),
        life=[(t.age, t.hits, bool(t.confirmed), bool(t.active), int(t.label), t.start_frame, t.end_frame)
              for t in tracks.values()],
        n_kp=[len(t.keypoints) for t in tracks.values()],
        hist=list(tracker.hist_tracks.keys()),
        klt={k: np.asarray(v, float).copy() for k, v in tracker.klt_bboxes.items()},
        H=None if tracker.homography is None else np.asarray(tracker.homography, float).copy())


def hip_pass(mot, video, n_frames, skip, prefetch=False, frames=None):
    """Runs fastmot_amd.MOT.step over the clip; returns per-frame records and, per detector frame, the
    injected detections and the embeddings read back from the HIP OSNet."""
    from fastmot_amd import Track
    frames = video.frames if frames is None else frames
    ext = mot.extractors[0]
    captured = {}
    cur = {}
    orig_post = ext.postprocess

    def post():
        emb = orig_post()
        captured[cur['f']] = np.array(emb, np.float32, copy=True)
        return emb              # the same object: MultiTracker.update recognises the device-resident copy
    ext.postprocess = post
    mot.detector_frame_skip = skip
    Track._count = 0
    mot.reset(1 / 30.)
    mot.tracker.klt_bboxes, mot.tracker.homography = {}, None      # (a reused MOT keeps the last clip's values)
    recs = []
    try:
        for f in range(n_frames):
            cur['f'] = f
            mot.detector._frame_idx = f
            nxt = frames[(f + 1) % len(frames)] if prefetch and f + 1 < n_frames else None
            mot.step(frames[f % len(frames)], next_frame=nxt)
            recs.append(frame_record(mot.tracker))
    finally:
        ext.postprocess = orig_post
    return recs, captured


def oracle_pass(size, metric, tracker_kw, video, n_frames, skip, embeddings, budget_s=None, labels=None, cv_impl=None):
    """The same clip through the CPU restatement (reference schedule mot.py:125-168: frame 0 = init, detector
    frames = flow + kalman + update, other frames = track).  Returns (records, seconds, frames done).
    cv_impl: None = cv_oracle (numpy), or the c_baseline module (the same routines compiled, pinned function by
    function to cv_oracle by tests/test_c_baseline.py; ~12x faster, which is what makes the 4K / 300-object and the
    KLT-heavy configurations fit a test budget)."""
    kw = {k: v for k, v in tracker_kw.items() if k != 'flow_cfg'}
    trk = cpu_tracker.OracleTracker(size, metric, cv_impl=cv_impl, **kw)
    trk.reset(1 / 30.)
    recs = []
    t0 = time.perf_counter()
    for f in range(n_frames):
        frame = video.frames[f % len(video.frames)]
        if f == 0:
            trk.init(frame, video.detections(0, labels=labels))
        elif f % skip == 0:
            trk.compute_flow(frame)
            trk.apply_kalman()
            trk.update(f, video.detections(f, labels=labels), embeddings[f])
        else:
            trk.track(frame)
        recs.append(frame_record(trk))
        if budget_s is not None and time.perf_counter() - t0 > budget_s:
            break
    return recs, time.perf_counter() - t0, len(recs)


def compare(hip, ora):
    """Frame-by-frame comparison; returns a summary dict (nothing is asserted here)."""
    n = min(len(hip), len(ora))
    out = dict(frames=n, ids_identical=True, boxes_identical=True, lifecycle_identical=True, history_identical=True,
               keypoint_counts_identical=True, klt_keys_identical=True, klt_box_max_px=0.0, H_max_abs=0.0,
               first_mismatch=None, max_tracks=0)

    def flag(key, f, detail):
        out[key] = False
        if out['first_mismatch'] is None:
            out['first_mismatch'] = f'{key} at frame {f}: {detail}'
    for f in range(n):
        a, b = hip[f], ora[f]
        out['max_tracks'] = max(out['max_tracks'], len(b['ids']))
        if a['ids'] != b['ids']:
            flag('ids_identical', f, f"{a['ids'][:12]}... vs {b['ids'][:12]}...")
            break                                   # everything downstream is meaningless after an ID split
        if a['tlbr'].shape != b['tlbr'].shape or not np.array_equal(a['tlbr'], b['tlbr']):
            d = np.abs(a['tlbr'] - b['tlbr'])
            flag('boxes_identical', f, f'{int((d > 0).any(axis=1).sum())} boxes differ, max {d.max()} px')
        if a['life'] != b['life']:
            flag('lifecycle_identical', f, 'age/hits/flags differ')
        if a['hist'] != b['hist']:
            flag('history_identical', f, f"{a['hist']} vs {b['hist']}")
        if a['n_kp'] != b['n_kp']:
            k = next(i for i, (x, y) in enumerate(zip(a['n_kp'], b['n_kp'])) if x != y)
            flag('keypoint_counts_identical', f, f"track {a['ids'][k]}: {a['n_kp'][k]} vs {b['n_kp'][k]}")
        if list(a['klt'].keys()) != list(b['klt'].keys()):
            flag('klt_keys_identical', f, f"{len(a['klt'])} vs {len(b['klt'])} KLT boxes")
        else:
            for k in a['klt']:
                out['klt_box_max_px'] = max(out['klt_box_max_px'], float(np.abs(a['klt'][k] - b['klt'][k]).max()))
        if (a['H'] is None) != (b['H'] is None):
            flag('klt_keys_identical', f, 'homography present on one side only')
        elif a['H'] is not None:
            out['H_max_abs'] = max(out['H_max_abs'], float(np.abs(a['H'] - b['H']).max()))
    out['all_identical'] = all(out[k] for k in ('ids_identical', 'boxes_identical', 'lifecycle_identical',
                                                'history_identical', 'keypoint_counts_identical',
                                                'klt_keys_identical'))
    return out
