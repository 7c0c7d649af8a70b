"""Micro-benchmark of one association stage (stage cost + LAP + readback) at the bench size."""
import os as _os
_os.environ.setdefault('FASTMOT_RANDOM_WEIGHTS', '1')   # no weight files offline
import sys, time
import numpy as np
sys.path.insert(0, '.')
from fastmot_amd import _lib
from fastmot_amd.runtime import get_context

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
ctx = get_context()
ctx.kf_configure(1 / 30., 2.25, 78.5, (0.08, 0.08), (0.14, 0.14), (4., 4.), (5., 5.), 5, 12, 0.6, 2)
ctx.set_frame_rect(np.array([0., 0., 1919., 1079.]))
ctx.feat_configure(512)
rng = np.random.default_rng(0)
tl = rng.uniform(0, 900, (n, 2)); boxes = np.concatenate([tl, tl + rng.uniform(30, 120, (n, 2))], 1)
slots = np.arange(1, n + 1, dtype=np.int32)
ctx.trk_create(slots, boxes)
emb = rng.normal(0, 1, (n, 512)).astype(np.float32); emb /= np.linalg.norm(emb, axis=1, keepdims=True)
ctx.emb_upload(emb)
ctx.feat_update(slots, np.arange(n, dtype=np.int32))
det = boxes[rng.permutation(n)] + rng.normal(0, 2, (n, 4))
lab = np.ones(n, np.int64)
ctx.assoc_prepare(_lib.METRIC_EUCLIDEAN, slots, boxes, lab, det, lab, np.zeros(n, np.uint8))
rows = np.arange(n, dtype=np.int32); cols = np.arange(n, dtype=np.int32)
for name, fn in (('assoc_stage(matching, LAP)', lambda: ctx.assoc_stage(_lib.STAGE_MATCHING, _lib.SOLVER_LAP, rows, cols, 0.02, 0.7, 0.8)),
                 ('assoc_stage(iou, LAP)', lambda: ctx.assoc_stage(_lib.STAGE_IOU, _lib.SOLVER_LAP, rows, cols, 0., 0.6, 1.)),
                 ('find_occluded', lambda: ctx.find_occluded(det, 0.4)),
                 ('assoc_prepare', lambda: ctx.assoc_prepare(_lib.METRIC_EUCLIDEAN, slots, boxes, lab, det, lab, np.zeros(n, np.uint8))),
                 ('synchronize (idle)', lambda: ctx.synchronize())):
    for _ in range(20):
        fn()
    ctx.synchronize()
    t = time.perf_counter()
    for _ in range(300):
        fn()
    ctx.synchronize()
    print(f'{name:<30} {(time.perf_counter() - t) / 300 * 1e6:8.1f} us/call')
