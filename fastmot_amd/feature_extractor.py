"""FeatureExtractor (API of fastmot/feature_extractor.py:11-98) on the HIP conv engine.

extract_async crops / resizes / normalises on the GPU straight from the resident frame and runs
OSNet for all boxes (batches of `batch_size`); postprocess returns the L2-normalised embeddings as a
host array and leaves a copy on the device for MultiTracker.update (no re-upload)."""

import numpy as np

from . import _lib, models
from .detector import bind_frame
from .engine import HipNet, NET_EXTRACTOR, NET_EXTRACTOR_B
from .runtime import get_context


class FeatureExtractor:
    def __init__(self, model='OSNet025', batch_size=16, weights=None, size=None, reuse_buffers=True,
                 split_batches=1, resident=True):
        """model : name of a class that inherits `models.ReID`; batch_size : samples per network
        launch (fastmot/feature_extractor.py:12-25).  `size` (frame width, height) is only needed
        when the extractor is used without a detector having bound the frame first.  `resident=False`
        defers putting the network on the device until the extractor first receives boxes."""
        self.model = models.ReID.get_model(model)
        assert batch_size >= 1
        self.batch_size = batch_size
        self.size = size

        self.feature_dim = self.model.OUTPUT_LAYOUT
        self.ctx = get_context()
        self.ctx.feat_configure(self.feature_dim)
        self.graph, _ = self.model.build_graph(weights)
        self._reuse_buffers = reuse_buffers
        self._split_batches = max(1, min(int(split_batches), 4))     # (default 1: more parts measured slower)
        self.backend = None
        self.extra_backends = []
        self.last_num_features = 0
        self._pending = False
        # the first extractor of a context is made resident right away; further ones (one per class id in
        # MOT) only when they actually receive boxes
        if resident:
            self._activate()

    def _activate(self):
        """Instantiates this extractor's network(s) on the device.  A context holds the networks of ONE
        extractor at a time (MOT builds one FeatureExtractor per class id, mot.py:99, but the reference's
        _split_bboxes_by_cls hands every box to the first one): an extractor that receives boxes while another
        one is resident swaps itself in -- correct for any split, and free in the reference's actual usage."""
        ctx = self.ctx
        prev = getattr(ctx, 'active_extractor', None)
        if prev is not None and prev is not self and prev._pending:
            raise RuntimeError('another FeatureExtractor of this context has a batch in flight: call its '
                               'postprocess() before extracting with a different model')
        # further instances of the network (own buffers, own streams): a batch runs as up to `split_batches`
        # concurrent parts of >= 4 crops (extract.hip); results are the same rows of the same embedding matrix
        for which in range(NET_EXTRACTOR_B, NET_EXTRACTOR_B + 3):              # a previous extractor's instances
            _lib.check(ctx.lib.fm_net_destroy(ctx.handle, which))
        self.backend = HipNet(ctx, NET_EXTRACTOR, self.graph, self.batch_size, reuse_buffers=self._reuse_buffers)
        self.extra_backends = [HipNet(ctx, NET_EXTRACTOR_B + i, self.graph, self.batch_size,
                                      reuse_buffers=self._reuse_buffers) for i in range(self._split_batches - 1)]
        ctx.extract_configure(self.graph.input.tid, self.model.INPUT_SHAPE[2], self.model.INPUT_SHAPE[1])
        ctx.active_extractor = self

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
        if len(tlbrs) == 0:
            self.last_num_features = 0          # nothing to do on the device (and nothing of another
            return                              # extractor's pending batch is disturbed)
        if getattr(self.ctx, 'active_extractor', None) is not self:
            self._activate()
        bind_frame(self.ctx, frame, size)
        self.last_num_features = self.ctx.extract_async(tlbrs)
        self._pending = True

    def postprocess(self):
        """Synchronizes and returns a NxM matrix of N extracted embeddings with dimension M."""
        if self.last_num_features == 0:
            return np.empty((0, self.feature_dim))
        embeddings = self.ctx.extract_sync(self.last_num_features)
        self._pending = False
        self.ctx.device_emb_host = embeddings
        return embeddings

    def null_embeddings(self, detections):
        """Returns a NxM matrix of N identical embeddings (disables feature extraction)."""
        embeddings = np.ones((len(detections), self.feature_dim))
        embeddings /= np.linalg.norm(embeddings, axis=1, keepdims=True)
        return embeddings
