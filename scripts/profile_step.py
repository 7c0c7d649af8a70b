"""Host-side wall-clock breakdown of MOT.step on the bench workload (no profiler attached): wraps the
context's C-ABI calls and the pipeline methods with perf_counter accumulators."""
import os as _os
_os.environ.setdefault('FASTMOT_RANDOM_WEIGHTS', '1')   # no weight files offline
import sys, time, collections
import numpy as np
sys.path.insert(0, '.')
import bench
from fastmot_amd import Track
from fastmot_amd.detector import DeviceFrame
from fastmot_amd.runtime import get_context
from fastmot_amd.utils.synthetic import SyntheticVideo

video = SyntheticVideo(bench.SIZE, n_ids=bench.N_DETS, n_frames=bench.RING, seed=100)
ctx = get_context()
ctx.frame_configure(bench.SIZE[0], bench.SIZE[1], bench.RING)
for i, fr in enumerate(video.frames):
    ctx.frame_ring_store(i, fr)
mot = bench.build_mot(bench.CONFIGS[1], video)
Track._count = 0
mot.reset(1 / 30.)

acc = collections.defaultdict(float)
cnt = collections.defaultdict(int)
tl0 = collections.defaultdict(float)     # sum of (start - step start), (end - step start): the timeline of a step
tl1 = collections.defaultdict(float)
step_t0 = [0.0]
PREFETCH = _os.environ.get('PROFILE_PREFETCH', '0') == '1'

def wrap(obj, name, label=None):
    fn = getattr(obj, name)
    label = label or name
    def w(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            e = time.perf_counter()
            acc[label] += e - t
            cnt[label] += 1
            tl0[label] += t - step_t0[0]
            tl1[label] += e - step_t0[0]
    setattr(obj, name, w)

for n in ('flow_predict', 'detect_async', 'detect_sync', 'extract_async',
          'extract_sync', 'assoc_prepare', 'assoc_stage', 'find_occluded', 'feat_update', 'trk_update_det',
          'trk_step', 'trk_step_ops', 'detect_async_next', 'frame_ring_select_next', 'frame_upload_next', 'frame_upload', 'frame_promote_next', 'emb_upload', 'synchronize', 'trk_create', 'feat_merge', 'frame_ring_select'):
    if hasattr(ctx, n):
        wrap(ctx, n, 'ctx.' + n)
wrap(mot.detector, 'detect_async', 'det.detect_async')
wrap(mot.detector, 'postprocess', 'det.postprocess')
wrap(mot.extractors[0], 'extract_async', 'ext.extract_async')
wrap(mot.extractors[0], 'postprocess', 'ext.postprocess')
wrap(mot.tracker, 'compute_flow', 'trk.compute_flow')
wrap(mot.tracker, 'apply_kalman', 'trk.apply_kalman')
wrap(mot.tracker, 'update', 'trk.update')
wrap(mot, '_step', 'mot._step')

DFRAMES = [DeviceFrame(i) for i in range(bench.RING)]    # the detector recognises a prefetched frame by identity
if _os.environ.get('PROFILE_H2D', '0') == '1':            # frames in pinned host memory, uploaded every step (bench.py)
    _host = ctx.pinned_frames(bench.RING)
    for _i, _fr in enumerate(video.frames):
        _host[_i] = _fr
    DFRAMES = [_host[_i] for _i in range(bench.RING)]


def run(n, start):
    for s in range(start, start + n):
        mot.detector._frame_idx = s % bench.RING
        step_t0[0] = time.perf_counter()
        if PREFETCH:
            mot.step(DFRAMES[s % bench.RING], next_frame=DFRAMES[(s + 1) % bench.RING])
        else:
            mot.step(DFRAMES[s % bench.RING])

run(20, 0)
ctx.synchronize()
acc.clear(); cnt.clear(); tl0.clear(); tl1.clear()
import ctypes as C0
ctx.lib.fm_flow_timing((C0.c_double * 5)(), 1)
N = 200
t0 = time.perf_counter()
run(N, 20)
ctx.synchronize()
el = time.perf_counter() - t0
print(f'ms/step {el / N * 1e3:.3f}')
import ctypes as C
t5 = (C.c_double * 5)()
ctx.lib.fm_flow_timing(t5, 0)
print('flow_predict stages (ms/call): begin %.3f prepare %.3f lk %.3f estimate %.3f' % tuple(t5[i] / max(t5[4], 1) for i in range(4)))
for k in sorted(acc, key=lambda k: -acc[k]):
    c = max(cnt[k], 1)
    print(f'{k:<24} {acc[k] / N * 1e3:8.3f} ms/step  calls/step {cnt[k] / N:5.2f}   '
          f'mean start +{tl0[k] / c * 1e3:6.3f}  end +{tl1[k] / c * 1e3:6.3f} ms after the step began')
