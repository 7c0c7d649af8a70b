import os, sys, time, threading
os.environ.setdefault('FASTMOT_RANDOM_WEIGHTS', '1')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fastmot_amd.utils.synthetic import SyntheticVideo
from fastmot_amd.flow import Flow
from fastmot_amd.detector import DeviceFrame, bind_frame
from fastmot_amd.runtime import get_context
from fastmot_amd.engine import HipNet, NET_EXTRACTOR
from fastmot_amd.models import ReID
size = (1920, 1080)
video = SyntheticVideo(size, n_ids=50, n_frames=2, seed=4)
ctx = get_context(); ctx.feat_configure(512)
ctx.frame_configure(size[0], size[1], 2)
for i in range(2): ctx.frame_ring_store(i, video.frames[i])
flow = Flow(size); flow.init(DeviceFrame(0)); bind_frame(ctx, DeviceFrame(1), size); ctx.flow_begin(); ctx.synchronize()
rng = np.random.default_rng(0)
for npts in (601, 6900):
    pts = np.stack([rng.uniform(20, size[0] / 2 - 20, npts), rng.uniform(20, size[1] / 2 - 20, npts)], 1).astype(np.float32)
    res = {}
    for name, v in (('one point / wave', 128), ('two points / wave', 0)):
        ctx.set_option('lk_variant', v)
        r = [ctx.flow_lk(pts), ctx.flow_lk(pts)]
        t0 = time.perf_counter()
        for _ in range(100): ctx.flow_lk(pts)
        dt = (time.perf_counter() - t0) / 100
        res[name] = r
        print(f'{npts} points, {name}: {dt * 1e3:.3f} ms per fm_flow_lk call; tracked {int(r[0][1].sum())}')
    a, b = res['one point / wave'], res['two points / wave']
    print('   identical:', all(np.array_equal(a[k][1], b[k][1]) and np.array_equal(a[k][0][a[k][1] > 0], b[k][0][b[k][1] > 0]) and np.array_equal(a[k][2][a[k][1] > 0], b[k][2][b[k][1] > 0]) for k in range(2)))
ctx.set_option('lk_variant', 0)
