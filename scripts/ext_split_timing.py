"""ReID extraction of 50 crops with 1..4 concurrent network instances, stand-alone (no pipeline): ms per call."""
import os as _os
_os.environ.setdefault('FASTMOT_RANDOM_WEIGHTS', '1')
import sys, time
sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import numpy as np
from fastmot_amd.feature_extractor import FeatureExtractor
from fastmot_amd.utils.synthetic import SyntheticVideo
from fastmot_amd.runtime import get_context

size = (1920, 1080)
video = SyntheticVideo(size, n_ids=50, n_frames=2, seed=100)
ctx = get_context()
boxes = video.gt[0].astype(np.float64)
for split in (1, 2, 3, 4):
    ext = FeatureExtractor('OSNet025', batch_size=64, size=size, split_batches=split)
    for _ in range(5):
        ext(video.frames[0], boxes)
    t = time.perf_counter()
    for _ in range(50):
        ext(video.frames[0], boxes)
    print(f'split {split}: {(time.perf_counter() - t) / 50 * 1e3:.3f} ms per extraction (incl. frame upload)')
