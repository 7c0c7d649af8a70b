"""Narrowing down what disturbs fm_flow_lk (see stress_lk4.py): single-operator hammers of the OSNet graph, with the
LK launch optionally isolated on its CUs (FASTMOT_LK_LDS = bytes of dynamic LDS requested per LK workgroup)."""
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
kinds = sys.argv[2].split(',') if len(sys.argv) > 2 else ['liteconv', 'liteconv_nogap', 'litechain', 'gated', 'pool', 'conv1x1', 'stem7']
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
rng = np.random.default_rng(0)
pts = np.stack([rng.uniform(20, size[0] / 2 - 20, 600), rng.uniform(20, size[1] / 2 - 20, 600)], 1).astype(np.float32)
base = [ctx.flow_lk(pts), ctx.flow_lk(pts)]


def make(kind):
    g = Graph(RandomWeights(seed=1), (64, 32), 16)
    if kind in ('liteconv', 'liteconv_nogap'):
        params = [g.lightconv_params(f'p{i}', 16) for i in range(4)]
        g.lightconv_group('l', [g.input] * 4, params, gap_slot=(kind == 'liteconv'))
    elif kind == 'litechain':
        params = [g.lightconv_params(f'p{t}.{i}', 16) for t in range(1, 5) for i in range(t)]
        g.lightchain('c', g.input, params, 'relu')
    elif kind == 'gated':
        g.gated_sum('g', [g.input, g.input, g.input, g.input], 1)
    elif kind == 'pool':
        g.pool(g.input, 2, 2, 0, avg=True)
    elif kind == 'conv1x1':
        x = g.input
        for i in range(4):
            x = g.conv(f'c{i}', x, 64 if i % 2 == 0 else 16, 1, 1, 'relu')
    elif kind == 'stem7':
        g = Graph(RandomWeights(seed=1), (256, 128), 3)
        g.conv('conv1', g.input, 16, 7, 2, 'relu', pad=3)
    return HipNet(ctx, NET_DETECTOR, g, 50, reuse_buffers=True)


for kind in kinds:
    try:
        net = make(kind)
    except Exception as exc:                      # an operator that cannot be built stand-alone
        print(f'hammer={kind:<15} skipped: {exc!r}'[:160], flush=True)
        continue
    stop = []

    def hammer():
        ctx.bind_thread()
        while not stop:
            net.run(50)
            ctx.synchronize()
    th = threading.Thread(target=hammer)
    th.start()
    bad_calls = bad_pts = 0
    worst = 0.
    try:
        for r in range(N // 2):
            for k in range(2):
                nxt, st, er = ctx.flow_lk(pts)
                ok = st > 0
                d = np.abs(nxt[ok] - base[k][0][ok]).max(axis=1) if np.array_equal(st, base[k][1]) else np.ones(1)
                if (d > 0).any():
                    bad_calls += 1
                    bad_pts += int((d > 0).sum())
                    worst = max(worst, float(d.max()))
    finally:
        stop.append(1)
        th.join()
    ctx.synchronize()
    print(f"LK_LDS={_os.environ.get('FASTMOT_LK_LDS', '0'):<7} hammer={kind:<15} calls differing {bad_calls}/{N}, points {bad_pts}, "
          f"worst {worst:.4g} px", flush=True)
    net.close()
