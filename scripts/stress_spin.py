"""Does a long deterministic kernel on the flow stream stay deterministic while grouped LightConv launches
saturate the GPU from another host thread?  (platform vs. library bug)"""
import os as _os
_os.environ.setdefault('FASTMOT_RANDOM_WEIGHTS', '1')   # no weight files offline
import sys, threading, ctypes as C
sys.path.insert(0, '.')
import numpy as np
from fastmot_amd.utils.synthetic import SyntheticVideo
from fastmot_amd.flow import Flow
from fastmot_amd.detector import DeviceFrame
from fastmot_amd.runtime import get_context
from fastmot_amd.engine import HipNet, NET_DETECTOR
from fastmot_amd.models.graph import Graph, RandomWeights

size = (960, 540)
video = SyntheticVideo(size, n_ids=4, n_frames=2, seed=1)
ctx = get_context()
ctx.frame_configure(size[0], size[1], 2)
for i in range(2):
    ctx.frame_ring_store(i, video.frames[i])
flow = Flow(size)
flow.init(DeviceFrame(0))
g = Graph(RandomWeights(seed=1), (64, 32), 16)
params = [g.lightconv_params(f'p{i}', 16) for i in range(4)]
g.lightconv_group('l', [g.input] * 4, params, gap_slot=False)
hnet = HipNet(ctx, NET_DETECTOR, g, 50, reuse_buffers=True)
hnet.run(50); ctx.synchronize()
stop = False
def hammer():
    ctx.bind_thread()
    while not stop:
        hnet.run(50); ctx.synchronize()
blocks, iters = 100, int(sys.argv[2]) if len(sys.argv) > 2 else 3000
def spin(mode):
    out = np.empty(256 * blocks, np.uint32)
    rc = ctx.lib.fm_debug_spin(ctx.handle, C.c_int(blocks), C.c_int(iters), C.c_int(mode), out.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return out
base = {m: spin(m) for m in (0, 1, 2, 3)}
th = threading.Thread(target=hammer); th.start()
for m in (0, 1, 2, 3):
    bad = 0
    for r in range(int(sys.argv[1])):
        o = spin(m)
        if not np.array_equal(o, base[m]):
            bad += 1
            if bad == 1:
                idx = np.flatnonzero(o != base[m])
                print(f'  mode {m}: first mismatch: {len(idx)} threads, e.g. gid {idx[:8]} (lanes {idx[:8] % 64})')
    print(f'mode {m} (shuffles={m & 1}, byte loads={(m >> 1) & 1}): {bad} of {sys.argv[1]} runs differ')
stop = True; th.join()
