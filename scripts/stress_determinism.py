"""Runs the same synthetic clip repeatedly through MOT.step (sequential and next-frame-prefetch modes) and
reports any run whose tracks differ from the first one -- a race detector for the two-thread pipeline."""
import os as _os
_os.environ.setdefault('FASTMOT_RANDOM_WEIGHTS', '1')   # no weight files offline
import sys
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import numpy as np
from test_mot_gpu import build_mot
from synthetic import SyntheticVideo
from fastmot_amd import Track

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
size = (960, 540)
video = SyntheticVideo(size, n_ids=10, n_frames=16, seed=11)
frames = video.frames

def run(prefetch):
    mot = build_mot(size, video, 1)
    Track._count = 0
    mot.reset(1 / 30.)
    rows = []
    for f in range(video.n_frames):
        mot.detector._frame_idx = f
        nxt = frames[f + 1] if prefetch and f + 1 < video.n_frames else None
        mot.step(frames[f], next_frame=nxt)
        rows.append([(t.trk_id, tuple(float(v) for v in t.tlbr), t.confirmed, t.active, t.age, t.hits, len(t.keypoints))
                     for t in mot.tracker.tracks.values()])
    mot.tracker._clear_tracks()
    return rows

base = run(False)
for mode in (False, True):
    bad = 0
    for r in range(reps):
        rows = run(mode)
        if rows != base:
            bad += 1
            for f, (a, b) in enumerate(zip(base, rows)):
                if a != b:
                    d = [(x, y) for x, y in zip(a, b) if x != y][:2]
                    print(f'mode prefetch={mode} rep {r}: first difference at frame {f}: {d}')
                    break
    print(f'prefetch={mode}: {bad} of {reps} runs differ from the reference run')
