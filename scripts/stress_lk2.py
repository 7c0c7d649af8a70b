"""LK alone: images are built once, then fm_flow_lk runs repeatedly (every second call has the same roles)."""
import os as _os
_os.environ.setdefault('FASTMOT_RANDOM_WEIGHTS', '1')   # no weight files offline
import sys, threading
sys.path.insert(0, '.')
import numpy as np
from fastmot_amd.utils.synthetic import SyntheticVideo
from fastmot_amd.flow import Flow
from fastmot_amd.detector import bind_frame, DeviceFrame
from fastmot_amd.runtime import get_context
from fastmot_amd.engine import HipNet, NET_DETECTOR
from fastmot_amd.models.graph import Graph, RandomWeights

size = (960, 540)
video = SyntheticVideo(size, n_ids=10, n_frames=3, seed=11)
ctx = get_context()
ctx.frame_configure(size[0], size[1], 3)
for i in range(3):
    ctx.frame_ring_store(i, video.frames[i])
F = [DeviceFrame(i) for i in range(3)]
from types import SimpleNamespace
ML = int(sys.argv[5]) if len(sys.argv) > 5 else 5
MC = int(sys.argv[6]) if len(sys.argv) > 6 else 10
flow = Flow(size, opt_flow_params=SimpleNamespace(winSize=(5, 5), maxLevel=ML, criteria=(3, MC, 0.03)))
g = Graph(RandomWeights(seed=1), (64, 32), 16)
params = [g.lightconv_params(f'p{i}', 16) for i in range(4)]
G = int(sys.argv[2]) if len(sys.argv) > 2 else 4
NB = int(sys.argv[3]) if len(sys.argv) > 3 else 50
g.lightconv_group('l', [g.input] * G, params[:G], gap_slot=False)
hnet = HipNet(ctx, NET_DETECTOR, g, NB, reuse_buffers=True)
hnet.run(NB); ctx.synchronize()
stop = False
def hammer():
    ctx.bind_thread()
    while not stop:
        hnet.run(NB)
        ctx.synchronize()
rng = np.random.default_rng(0)
pts = np.stack([rng.uniform(20, size[0] / 2 - 20, 800), rng.uniform(20, size[1] / 2 - 20, 800)], 1).astype(np.float32)
flow.init(F[0])
bind_frame(ctx, F[1], size)
ctx.flow_begin()
ctx.synchronize()
import hashlib
IDS = (0, 1) + tuple(range(2, 3 + ML)) + tuple(range(10, 11 + ML))
hashes = lambda: tuple(hashlib.md5(ctx.flow_read_image(w).tobytes()).hexdigest()[:6] for w in IDS)
h0 = hashes()
a0 = ctx.flow_lk(pts); b0 = ctx.flow_lk(pts)        # roles: (0->1), then (1->0)
th = threading.Thread(target=hammer); th.start()
if len(sys.argv) > 4 and sys.argv[4] == 'pyr':
    bad_p = 0
    for r in range(int(sys.argv[1])):
        ctx.flow_begin()                 # gray + resize + pyrdown + scharr of the same frame again
        ctx.synchronize()
        bad_p += hashes() != h0
    print('pyramid rebuilds that differ:', bad_p)
bad = 0
reps = int(sys.argv[1])
for r in range(reps):
    a = ctx.flow_lk(pts); b = ctx.flow_lk(pts)
    for x, x0 in ((a, a0), (b, b0)):
        if not (np.array_equal(x[1], x0[1]) and np.array_equal(x[0][x0[1] > 0], x0[0][x0[1] > 0])):
            bad += 1
stop = True; th.join()
h1 = hashes()
print('images unchanged after the stress:', h0 == h1)
ra = ctx.flow_lk(pts); rb = ctx.flow_lk(pts)
print('LK after the hammer stopped equals the base:', np.array_equal(ra[0][ra[1] > 0], a0[0][a0[1] > 0]), np.array_equal(rb[0][rb[1] > 0], b0[0][b0[1] > 0]))
print(f'G={G} batch={NB} maxLevel={ML} maxCount={MC}', end=' '); print(f'LK only: {bad} of {2 * reps} calls differ')
