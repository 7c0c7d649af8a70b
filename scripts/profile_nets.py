"""Per-network timing on the GPU: whole-run wall time (stream-synchronised) and per-layer HIP-event
times of the conv (MFMA) launches -> achieved TFLOP/s against the 2.5 PFLOP/s dense fp16 peak."""
import os as _os
_os.environ.setdefault('FASTMOT_RANDOM_WEIGHTS', '1')   # no weight files offline
import sys, time, json
import numpy as np
sys.path.insert(0, '.')
from fastmot_amd.runtime import get_context
from fastmot_amd.engine import HipNet, NET_DETECTOR, NET_EXTRACTOR
from fastmot_amd.models import YOLO, ReID

ctx = get_context()
print(ctx.device_info())
res = {}
for which, (name, cls, batch) in enumerate((('YOLOv4_608', YOLO.get_model('YOLOv4_608'), 1), ('OSNet025', ReID.get_model('OSNet025'), 50))):
    g, _ = cls.build_graph()
    if which == 1:
        ctx.feat_configure(512)
    net = HipNet(ctx, which, g, batch, reuse_buffers=True)
    for _ in range(3):
        net.run(batch)
    ctx.synchronize()
    t = time.perf_counter()
    iters = 20
    for _ in range(iters):
        net.run(batch)
    ctx.synchronize()
    wall = (time.perf_counter() - t) / iters * 1e3
    prof = net.profile(batch, 3)
    flops, byts = net.cost(batch)
    res[name] = dict(wall_ms=wall, **prof, conv_gflop=flops / 1e9, conv_mbytes=byts / 1e6,
                     conv_tflops_wall=flops / wall / 1e9, conv_tflops_kernel=flops / prof['conv_ms'] / 1e9)
    print(name, json.dumps(res[name]))
