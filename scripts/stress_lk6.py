"""Which part of the OSNet-x0.25 graph still disturbs the (SGPR-capped) LK kernel?  Hammers = prefixes of the layer
table, and single layers of it (their inputs hold whatever the arena holds: only the launches matter)."""
import os as _os
_os.environ.setdefault('FASTMOT_RANDOM_WEIGHTS', '1')
import sys, threading
sys.path.insert(0, '.')
import numpy as np
from fastmot_amd.utils.synthetic import SyntheticVideo
from fastmot_amd.flow import Flow
from fastmot_amd.detector import DeviceFrame, bind_frame
from fastmot_amd.runtime import get_context
from fastmot_amd.engine import HipNet, NET_EXTRACTOR
from fastmot_amd.models import ReID

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
size = (960, 540)
video = SyntheticVideo(size, n_ids=6, n_frames=2, seed=4)
ctx = get_context()
ctx.feat_configure(512)
ctx.frame_configure(size[0], size[1], 2)
for i in range(2):
    ctx.frame_ring_store(i, video.frames[i])
flow = Flow(size)
flow.init(DeviceFrame(0))
bind_frame(ctx, DeviceFrame(1), size)
ctx.flow_begin()
ctx.synchronize()
rng = np.random.default_rng(0)
pts = np.stack([rng.uniform(20, size[0] / 2 - 20, 600), rng.uniform(20, size[1] / 2 - 20, 600)], 1).astype(np.float32)
base = [ctx.flow_lk(pts), ctx.flow_lk(pts)]
OPN = {0: 'conv', 2: 'maxpool', 3: 'avgpool', 8: 'head', 9: 'liteconv', 11: 'gated_sum', 12: 'stemconv', 16: 'litechain'}
g0, _ = ReID.get_model('OSNet025').build_graph()
full = list(g0.layers)
print('layers:', ' '.join(f"{i}:{OPN.get(d['op'], d['op'])}" for i, d in enumerate(full)), flush=True)


def trial(label, layers):
    g, _ = ReID.get_model('OSNet025').build_graph()
    g.layers[:] = layers(g.layers)
    net = HipNet(ctx, NET_EXTRACTOR, g, 50, reuse_buffers=False)
    net.run(50)
    ctx.synchronize()
    stop = []

    def hammer():
        ctx.bind_thread()
        while not stop:
            net.run(50)
            ctx.synchronize()
    th = threading.Thread(target=hammer)
    th.start()
    bad = pts_bad = 0
    worst = 0.
    try:
        for r in range(N // 2):
            for k in range(2):
                nxt, st, er = ctx.flow_lk(pts)
                ok = st > 0
                d = np.abs(nxt[ok] - base[k][0][ok]).max(axis=1) if np.array_equal(st, base[k][1]) else np.ones(1)
                if (d > 0).any():
                    bad += 1
                    pts_bad += int((d > 0).sum())
                    worst = max(worst, float(d.max()))
    finally:
        stop.append(1)
        th.join()
    ctx.synchronize()
    net.close()
    print(f'hammer={label:<22} calls differing {bad}/{N}, points {pts_bad}, worst {worst:.3g} px', flush=True)


n = len(full)
for k in (n, 3, 8, 14, 20, 26):
    trial(f'prefix[:{k}]', lambda L, k=k: L[:k])
kinds = {}
for i, d in enumerate(full):
    kinds.setdefault(d['op'], []).append(i)
for op, idx in kinds.items():
    trial(f'only {OPN.get(op, op)} x{len(idx)}', lambda L, idx=idx: [L[i] for i in idx])
