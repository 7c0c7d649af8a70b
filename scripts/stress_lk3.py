"""Determinism of fm_flow_lk under load for one I/O mode (FASTMOT_LK_IO in the environment): N calls while another
thread keeps the CUs busy with grouped LightConv launches; prints how many calls differ from the idle result."""
import os as _os
_os.environ.setdefault('FASTMOT_RANDOM_WEIGHTS', '1')
import sys, threading
sys.path.insert(0, '.')
import numpy as np
from fastmot_amd.utils.synthetic import SyntheticVideo
from fastmot_amd.flow import Flow
from fastmot_amd.detector import DeviceFrame, bind_frame
from fastmot_amd.runtime import get_context
from fastmot_amd.engine import HipNet, NET_DETECTOR
from fastmot_amd.models.graph import Graph, RandomWeights

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
size = (960, 540)
video = SyntheticVideo(size, n_ids=6, n_frames=2, seed=4)
ctx = get_context()
ctx.frame_configure(size[0], size[1], 2)
for i in range(2):
    ctx.frame_ring_store(i, video.frames[i])
flow = Flow(size)
flow.init(DeviceFrame(0))
bind_frame(ctx, DeviceFrame(1), size)
ctx.flow_begin()
ctx.synchronize()
g = Graph(RandomWeights(seed=1), (64, 32), 16)
params = [g.lightconv_params(f'p{i}', 16) for i in range(4)]
g.lightconv_group('l', [g.input] * 4, params)
net = HipNet(ctx, NET_DETECTOR, g, 50, reuse_buffers=True)
net.run(50)
ctx.synchronize()
rng = np.random.default_rng(0)
pts = np.stack([rng.uniform(20, size[0] / 2 - 20, 600), rng.uniform(20, size[1] / 2 - 20, 600)], 1).astype(np.float32)
base = [ctx.flow_lk(pts), ctx.flow_lk(pts)]
idle_bad = 0
for r in range(20):
    for k in range(2):
        nxt, st, er = ctx.flow_lk(pts)
        idle_bad += not (np.array_equal(st, base[k][1]) and np.array_equal(nxt[st > 0], base[k][0][st > 0]))
stop = []
def hammer():
    ctx.bind_thread()
    while not stop:
        net.run(50)
        ctx.synchronize()
th = threading.Thread(target=hammer)
th.start()
bad = 0
detail = []
try:
    for r in range(N // 2):
        for k in range(2):
            nxt, st, er = ctx.flow_lk(pts)
            ok = np.array_equal(st, base[k][1]) and np.array_equal(nxt[st > 0], base[k][0][st > 0])
            if not ok:
                bad += 1
                d = np.flatnonzero((st != base[k][1]) | (nxt != base[k][0]).any(1))
                detail.append((r, k, len(d), d[:4].tolist(), nxt[d[:2]].tolist(), base[k][0][d[:2]].tolist()))
finally:
    stop.append(1)
    th.join()
print(f"LK_IO={_os.environ.get('FASTMOT_LK_IO', 'default')}: idle {idle_bad}/40 differ, under load {bad}/{N} differ", detail[:4])
