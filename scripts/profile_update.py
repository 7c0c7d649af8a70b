"""cProfile of MultiTracker.update on the bench workload (host-side Python cost of the association)."""
import os as _os
_os.environ.setdefault('FASTMOT_RANDOM_WEIGHTS', '1')   # no weight files offline
import cProfile, pstats, sys, io
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
def run(n, start):
    for s in range(start, start + n):
        mot.detector._frame_idx = s % bench.RING
        mot.step(DeviceFrame(s % bench.RING))
run(20, 0)
pr = cProfile.Profile()
orig = mot.tracker.update
def wrapped(*a, **k):
    pr.enable()
    try:
        return orig(*a, **k)
    finally:
        pr.disable()
mot.tracker.update = wrapped
run(200, 20)
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(22)
print(s.getvalue()[:4500])
