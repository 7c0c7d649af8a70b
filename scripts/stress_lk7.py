"""What do the disturbed LK results look like?  Un-isolated LK (lk_isolation = 0) beside the detector network with the
LDS-halo streamed conv enabled (FASTMOT_CONVS_HALO=1, a known disturber): for every differing point print the
reference and the returned values, the status / error fields, and whether the returned position equals ANOTHER point's
reference result (a mix-up between wavefronts) or the point's own input (no update at all)."""
import os as _os
_os.environ.setdefault('FASTMOT_RANDOM_WEIGHTS', '1')
_os.environ.setdefault('FASTMOT_CONVS_HALO', '1')
import sys, threading
sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import numpy as np
from fastmot_amd.utils.synthetic import SyntheticVideo
from fastmot_amd.flow import Flow
from fastmot_amd.detector import DeviceFrame, bind_frame
from fastmot_amd.runtime import get_context
from fastmot_amd.engine import HipNet, NET_DETECTOR
from fastmot_amd.models import YOLO

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
g, _ = YOLO.get_model('YOLOv4_608').build_graph()
net = HipNet(ctx, NET_DETECTOR, g, 1, reuse_buffers=True)
net.run(1)
ctx.synchronize()
rng = np.random.default_rng(0)
pts = np.stack([rng.uniform(20, size[0] / 2 - 20, 600), rng.uniform(20, size[1] / 2 - 20, 600)], 1).astype(np.float32)
ctx.set_option('lk_isolation', 0)
bases = [ctx.flow_lk(pts), ctx.flow_lk(pts)]        # consecutive calls track in opposite directions (the call swaps the image sets)
for k in range(2):
    b2 = ctx.flow_lk(pts)
    print(f'idle repeat {k}: status equal', np.array_equal(bases[k][1], b2[1]), ' positions of tracked points equal',
          np.array_equal(bases[k][0][bases[k][1] > 0], b2[0][b2[1] > 0]), ' errors of tracked points equal',
          np.array_equal(bases[k][2][bases[k][1] > 0], b2[2][b2[1] > 0]))
stop = []


def hammer():
    ctx.bind_thread()
    while not stop:
        net.run(1)
        ctx.synchronize()


th = threading.Thread(target=hammer)
th.start()
seen = 0
try:
    for r in range(N):
        nxt, st, err = ctx.flow_lk(pts)
        base = bases[r % 2]
        ok = base[1] > 0
        bad = np.flatnonzero((st != base[1]) | (ok & ((nxt != base[0]).any(1) | (err != base[2]))))
        for i in bad:
            seen += 1
            if seen > 25:
                continue
            d_other = np.abs(base[0] - nxt[i]).max(1)
            j = int(np.argmin(d_other))
            print(f'call {r:3d} point {i:3d}: ref ({base[0][i][0]:.5f}, {base[0][i][1]:.5f}) st {base[1][i]} err {base[2][i]:.5f}'
                  f' | got ({nxt[i][0]:.5f}, {nxt[i][1]:.5f}) st {st[i]} err {err[i]:.5f}'
                  f' | delta ({nxt[i][0] - base[0][i][0]:+.6f}, {nxt[i][1] - base[0][i][1]:+.6f})'
                  f' | input ({pts[i][0]:.3f}, {pts[i][1]:.3f})'
                  f' | nearest other reference: point {j} at {d_other[j]:.6f}')
finally:
    stop.append(1)
    th.join()
    ctx.set_option('lk_isolation', 1)
print(f'{seen} differing point results in {N} calls')
