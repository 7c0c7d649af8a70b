"""FeatureExtractor (API of fastmot/feature_extractor.py:11-98) on the HIP conv engine.

extract_async crops / resizes / normalises on the GPU straight from the resident frame and runs
OSNet for all boxes (batches of `batch_size`); postprocess returns the L2-normalised embeddings as a
host array and leaves a copy on the device for MultiTracker.update (no re-upload)."""
import os

import numpy as np

from . import _lib, models
from .detector import bind_frame
from .engine import HipNet, NET_EXTRACTOR, NET_EXTRACTOR_B
from .runtime import get_context


class FeatureExtractor:
    def __init__(self, model='OSNet025', batch_size=16, weights=None, size=None, reuse_buffers=True,
                 split_batches=int(os.environ.get('FASTMOT_EXT_SPLIT', '2'))):
        """model : name of a class that inherits `models.ReID`; batch_size : samples per network
        launch (fastmot/feature_extractor.py:12-25).  `size` (frame width, height) is only needed
        when the extractor is used without a detector having bound the frame first."""
        self.model = models.ReID.get_model(model)
        assert batch_size >= 1
        self.batch_size = batch_size
        self.size = size

        self.feature_dim = self.model.OUTPUT_LAYOUT
        self.ctx = get_context()
        self.ctx.feat_configure(self.feature_dim)
        self.graph, _ = self.model.build_graph(weights)
        self.backend = HipNet(self.ctx, NET_EXTRACTOR, self.graph, self.batch_size, reuse_buffers=reuse_buffers)
        # further instances of the network (own buffers, own streams): a batch runs as up to `split_batches`
        # concurrent parts of >= 4 crops (extract.hip); results are the same rows of the same embedding matrix
        for which in range(NET_EXTRACTOR_B, NET_EXTRACTOR_B + 3):              # a previous extractor's instances
            _lib.check(self.ctx.lib.fm_net_destroy(self.ctx.handle, which))
        self.extra_backends = [HipNet(self.ctx, NET_EXTRACTOR_B + i, self.graph, self.batch_size, reuse_buffers=reuse_buffers)
                               for i in range(max(0, min(int(split_batches), 4) - 1))]
        self.ctx.extract_configure(self.graph.input.tid, self.model.INPUT_SHAPE[2], self.model.INPUT_SHAPE[1])
        self.last_num_features = 0

    def __call__(self, frame, tlbrs):
        """Extract feature embeddings from bounding boxes synchronously."""
        self.extract_async(frame, tlbrs)
        return self.postprocess()

    @property
    def metric(self):
        return self.model.METRIC

    def extract_async(self, frame, tlbrs):
        """Extract feature embeddings from bounding boxes asynchronously."""
        size = self.size if self.size is not None else getattr(self.ctx, 'frame_size', None)
        if size is None:
            size = (frame.shape[1], frame.shape[0])
        bind_frame(self.ctx, frame, size)
        self.last_num_features = self.ctx.extract_async(tlbrs)

    def postprocess(self):
        """Synchronizes and returns a NxM matrix of N extracted embeddings with dimension M."""
        if self.last_num_features == 0:
            self.ctx.device_emb_host = None
            return np.empty((0, self.feature_dim))
        embeddings = self.ctx.extract_sync(self.last_num_features)
        self.ctx.device_emb_host = embeddings
        return embeddings

    def null_embeddings(self, detections):
        """Returns a NxM matrix of N identical embeddings (disables feature extraction)."""
        embeddings = np.ones((len(detections), self.feature_dim))
        embeddings /= np.linalg.norm(embeddings, axis=1, keepdims=True)
        return embeddings
