"""Wall-clock breakdown of Flow.predict on the synthetic 1080p/50-track clip (no detector running)."""
import os as _os
_os.environ.setdefault('FASTMOT_RANDOM_WEIGHTS', '1')   # no weight files offline
import sys, time
import numpy as np
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import bench
from fastmot_amd import Track
from fastmot_amd.runtime import get_context
from fastmot_amd.utils.synthetic import SyntheticVideo
from fastmot_amd.detector import DeviceFrame

video = SyntheticVideo(bench.SIZE, n_ids=50, n_frames=16, seed=100)
ctx = get_context()
ctx.frame_configure(1920, 1080, 16)
for i, fr in enumerate(video.frames):
    ctx.frame_ring_store(i, fr)
mot = bench.build_mot(bench.CONFIGS[1], video)
Track._count = 0
mot.reset(1 / 30.)
for s in range(8):
    mot.detector._frame_idx = s % 16
    mot.step(DeviceFrame(s % 16))
flow = mot.tracker.flow
import fastmot_amd.flow as fmod
# instrument ctx calls
acc = {}
def wrap(name):
    fn = getattr(ctx, name)
    def w(*a, **k):
        t = time.perf_counter(); r = fn(*a, **k); acc[name] = acc.get(name, 0) + time.perf_counter() - t; return r
    setattr(ctx, name, w)
for n in ('flow_begin', 'flow_prepare', 'flow_lk', 'flow_estimate', 'frame_ring_select'):
    wrap(n)
N = 40
tot = 0
for s in range(8, 8 + N):
    ctx.synchronize()
    tracks = [t for t in mot.tracker.tracks.values() if t.active]
    from fastmot_amd.detector import bind_frame
    t0 = time.perf_counter()
    boxes, H = flow.predict(DeviceFrame(s % 16), tracks)
    tot += time.perf_counter() - t0
    mot.tracker.klt_bboxes, mot.tracker.homography = boxes, H
    mot.tracker.apply_kalman()
    dets = video.detections(s % 16)
    emb = mot.extractors[0](DeviceFrame(s % 16), dets.tlbr)
    mot.tracker.update(s, dets, emb)
print('predict total ms', tot / N * 1e3, 'n_tracks', len(tracks))
for k, v in acc.items():
    print(f'  {k:<20} {v / N * 1e3:.3f} ms')
print('  python remainder    ', (tot - sum(acc.values())) / N * 1e3)
pts = sum(len(t.keypoints) for t in mot.tracker.tracks.values())
print('keypoints alive', pts, 'bg', len(flow.bg_keypoints))
