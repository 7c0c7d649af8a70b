"""Multi-stream deployment = one process per stream.  Two MOT processes running CONCURRENTLY (here
on the one GPU of the test box; on a node every rank has its own GPU) with the gallery exchange off must each produce
exactly the records of the same clip tracked alone: streams share nothing (the reference's only shared state is the
class-level Track._count, track.py:130 -- per process here as there)."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]

WORKER = r'''
import json, os, sys, time
sys.path[:0] = [%(root)r, %(root)r + '/tests', %(root)r + '/oracle']
os.environ['FASTMOT_RANDOM_WEIGHTS'] = '1'
rank = int(sys.argv[1])
from types import SimpleNamespace
import scenes
import fastmot_amd.mot as mot_mod
from fastmot_amd import Track
from fastmot_amd.detector import YOLODetector
from synthetic import InjectedYOLODetector, SyntheticVideo

size = (960, 540)
video = SyntheticVideo(size, n_ids=12, n_frames=48, seed=40 + rank)
mot_mod.YOLODetector = InjectedYOLODetector
mot = mot_mod.MOT(size, detector_type='YOLO', detector_frame_skip=2, class_ids=(1,),
                  yolo_detector_cfg=SimpleNamespace(model='YOLOv4', conf_thresh=0.25, nms_thresh=0.5, max_area=800000,
                                                    min_aspect_ratio=1.2),
                  feature_extractor_cfgs=(SimpleNamespace(model='OSNet025', batch_size=16),),
                  tracker_cfg=SimpleNamespace(**scenes.tracker_kwargs()))
mot.detector.bind_video(video)
Track._count = 0
mot.reset(1 / 30.)
# start line: both processes begin stepping together (the parent creates the file once both are ready)
open(sys.argv[2] + f'.ready{rank}', 'w').close()
t0 = time.time()
while not os.path.exists(sys.argv[2] + '.go') and time.time() - t0 < 120:
    time.sleep(0.01)
rows = []
for f in range(video.n_frames):
    mot.detector._frame_idx = f
    mot.step(video.frames[f], next_frame=video.frames[f + 1] if f + 1 < video.n_frames else None)
    rows.append([(t.trk_id, [float(v) for v in t.tlbr], t.age, t.hits, bool(t.confirmed), len(t.keypoints))
                 for t in mot.tracker.tracks.values()])
mot.tracker._clear_tracks()
print('RESULT ' + json.dumps(rows))
'''


def _launch(rank, tag):
    return subprocess.Popen([sys.executable, '-c', WORKER % {'root': str(ROOT)}, str(rank), tag],
                            stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


def _collect(p):
    out, err = p.communicate(timeout=900)
    assert p.returncode == 0, err[-3000:]
    return json.loads([l for l in out.splitlines() if l.startswith('RESULT ')][-1][7:])


@pytest.mark.gpu
def test_two_concurrent_streams_equal_their_solo_runs(tmp_path):
    import os
    import time
    solo = []
    for rank in range(2):
        tag = str(tmp_path / f'solo{rank}')
        open(tag + '.go', 'w').close()
        solo.append(_collect(_launch(rank, tag)))
    tag = str(tmp_path / 'duo')
    procs = [_launch(rank, tag) for rank in range(2)]
    t0 = time.time()
    while not all(os.path.exists(f'{tag}.ready{r}') for r in range(2)) and time.time() - t0 < 600:
        time.sleep(0.05)
    open(tag + '.go', 'w').close()
    duo = [_collect(p) for p in procs]
    for rank in range(2):
        assert len(duo[rank]) == 48 and max(len(r) for r in duo[rank]) >= 10
        assert duo[rank] == solo[rank], f'stream {rank} differs when another stream runs beside it'
    assert duo[0] != duo[1]                  # (different clips: the comparison is not vacuous)
