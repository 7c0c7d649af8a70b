"""The ReID-gallery exchange on the real backend: one rank on RCCL through the library's C ABI (fm_gallery_*:
csrc/gallery.hip, librccl.so bound at run time -- NO torch in the process), in a subprocess.  With a single rank
there are no foreign entries, so the tracks must be identical to a run without the exchange; what is exercised is the
side-stream path itself (pinned staging, H2D, ncclAllGather on the communicator, D2H, completion event, end-of-stream
protocol) and its statistics."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]

WORKER = r'''
import json, os, sys
sys.path[:0] = [%(root)r, %(root)r + '/tests']
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=%(port)r, HSA_ENABLE_IPC_MODE_LEGACY='0', FASTMOT_RANDOM_WEIGHTS='1')
import numpy as np
from types import SimpleNamespace
import scenes
import fastmot_amd.mot as mot_mod
from fastmot_amd import Track
from fastmot_amd.detector import YOLODetector
from fastmot_amd.gallery import GallerySync
from synthetic import InjectedYOLODetector, SyntheticVideo

size = (960, 540)
video = SyntheticVideo(size, n_ids=10, n_frames=40, seed=13)


def run(sync):
    kw = scenes.tracker_kwargs()
    kw['max_age'] = 2                       # tracks get lost quickly -> the local gallery is not empty
    if sync is not None:
        kw['gallery_sync'] = sync
    mot_mod.YOLODetector = InjectedYOLODetector
    try:
        mot = mot_mod.MOT(size, detector_type='YOLO', detector_frame_skip=1, class_ids=(1,),
                          yolo_detector_cfg=SimpleNamespace(model='YOLOv4', conf_thresh=0.25, nms_thresh=0.5,
                                                            max_area=800000, min_aspect_ratio=1.2),
                          feature_extractor_cfgs=(SimpleNamespace(model='OSNet025', batch_size=16),),
                          tracker_cfg=SimpleNamespace(**kw))
    finally:
        mot_mod.YOLODetector = YOLODetector
    mot.detector.bind_video(video)
    Track._count = 0
    mot.reset(1 / 30.)
    rows = []
    for f in range(video.n_frames):
        mot.detector._frame_idx = f
        # every third object is not detected in frames 12..24: its track is lost and enters the history
        if 12 <= f < 24:
            orig = video.detections
            video.detections = lambda i, label=1, labels=None, o=orig: o(i, label, labels)[np.arange(video.n_ids) %% 3 != 0]
        mot.step(video.frames[f])
        if 12 <= f < 24:
            video.detections = orig
        rows.append([(t.trk_id, tuple(t.tlbr), t.age, t.hits) for t in mot.tracker.tracks.values()])
    hist = len(mot.tracker.hist_tracks)
    mot.tracker._clear_tracks()
    return rows, hist


plain, _ = run(None)
sync = GallerySync(history_size=50, feat_dim=512)
with_sync, hist = run(sync)
rounds = sync.close()
stats = sync.stats()
assert 'torch' not in sys.modules, 'the RCCL path must not need torch'
print('RESULT ' + json.dumps(dict(identical=plain == with_sync, stats=stats, rounds=rounds, foreign=len(sync.foreign),
                                  frames=len(plain), max_hist=hist)))
'''


@pytest.mark.gpu
def test_gallery_exchange_on_rccl_single_rank():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = str(s.getsockname()[1])
    s.close()
    res = subprocess.run([sys.executable, '-c', WORKER % {'root': str(ROOT), 'port': port}], capture_output=True,
                         text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith('RESULT ')][-1][7:])
    assert out['identical'] and out['frames'] == 40
    st = out['stats']
    assert st['backend'] == 'rccl' and st['world'] == 1 and st['collectives'] >= 39 and st['asynchronous']
    assert st['bytes_per_rank'] == 32 + 50 * 24 + 50 * 512 * 4
    assert 'allgather_stream_us' in st and out['foreign'] == 0 and out['rounds'] >= 1
