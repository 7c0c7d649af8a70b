"""Which concurrent workload disturbs fm_flow_lk?  For each hammer (grouped LightConv launches / a plain 3x3 conv
layer / the whole OSNet / the whole YOLOv4 / none) 400 calls of fm_flow_lk on constant inputs are compared with the
idle result, and the KLT images (gray, both pyramids) are read back afterwards and compared with their idle copies."""
import os as _os
_os.environ.setdefault('FASTMOT_RANDOM_WEIGHTS', '1')
import sys, threading
sys.path.insert(0, '.')
import numpy as np
from fastmot_amd.utils.synthetic import SyntheticVideo
from fastmot_amd.flow import Flow
from fastmot_amd.detector import DeviceFrame, bind_frame
from fastmot_amd.runtime import get_context
from fastmot_amd.engine import HipNet, NET_DETECTOR, NET_EXTRACTOR
from fastmot_amd.models import YOLO, ReID
from fastmot_amd.models.graph import Graph, RandomWeights

N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
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


def images():
    return [ctx.flow_read_image(w).copy() for w in (0, 1, 2, 3, 4, 5, 6, 7, 10, 11, 12, 13, 14, 15)]


rng = np.random.default_rng(0)
pts = np.stack([rng.uniform(20, size[0] / 2 - 20, 600), rng.uniform(20, size[1] / 2 - 20, 600)], 1).astype(np.float32)
base = [ctx.flow_lk(pts), ctx.flow_lk(pts)]
img0 = images()


def make(kind):
    if kind == 'liteconv':
        g = Graph(RandomWeights(seed=1), (64, 32), 16)
        params = [g.lightconv_params(f'p{i}', 16) for i in range(4)]
        g.lightconv_group('l', [g.input] * 4, params)
        return HipNet(ctx, NET_DETECTOR, g, 50, reuse_buffers=True), 50
    if kind == 'conv3x3':
        g = Graph(RandomWeights(seed=1), (38, 38), 256)
        x = g.input
        for i in range(6):
            x = g.conv(f'c{i}', x, 256, 3, 1, 'leaky')
        return HipNet(ctx, NET_DETECTOR, g, 1, reuse_buffers=True), 1
    if kind == 'osnet':
        ctx.feat_configure(512)
        g, _ = ReID.get_model('OSNet025').build_graph()
        return HipNet(ctx, NET_EXTRACTOR, g, 50, reuse_buffers=True), 50
    if kind == 'yolov4':
        g, _ = YOLO.get_model('YOLOv4_608').build_graph()
        return HipNet(ctx, NET_DETECTOR, g, 1, reuse_buffers=True), 1
    return None, 0


for kind in ('none', 'liteconv', 'conv3x3', 'osnet', 'yolov4', 'liteconv'):
    net, batch = make(kind)
    stop = []

    def hammer():
        ctx.bind_thread()
        while not stop:
            if net is not None:
                net.run(batch)
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
    img1 = images()
    changed = [i for i, (a, b) in enumerate(zip(img0, img1)) if not np.array_equal(a, b)]
    idle_again = sum(not (np.array_equal(ctx.flow_lk(pts)[0], base[k][0])) for _ in range(10) for k in range(2))
    print(f'hammer={kind:<9} calls differing {bad_calls}/{N}, points {bad_pts}, worst {worst:.4g} px; KLT images changed: '
          f'{changed}; idle afterwards: {idle_again}/20 differ', flush=True)
    if net is not None:
        net.close()
