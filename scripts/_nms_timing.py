import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['FASTMOT_RANDOM_WEIGHTS'] = '1'
import numpy as np
from fastmot_amd.detector import YOLODetector
from fastmot_amd.runtime import get_context
ctx = get_context()
rng = np.random.default_rng(5)
det = YOLODetector((1920, 1080), (0,), model='YOLOv4', conf_thresh=0.3, nms_thresh=0.45, max_area=800000, min_aspect_ratio=0.5, max_candidates=8192)
for n in (100, 500, 1500, 3000, 6000):
    rows = np.stack([rng.uniform(0, 0.9, n), rng.uniform(0, 0.8, n), rng.uniform(0.01, 0.15, n), rng.uniform(0.02, 0.3, n),
                     rng.uniform(0.3, 1, n), np.zeros(n), rng.uniform(0.9, 1, n)], 1).astype(np.float32)
    out = ctx.filter_dets(rows)
    t0 = time.perf_counter()
    for _ in range(20):
        out = ctx.filter_dets(rows)
    print(f'K={n}: {len(out)} detections, {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per filter_dets call (upload + sort + NMS + download)')
