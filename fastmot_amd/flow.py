"""Flow (API of fastmot/flow.py:16-264): KLT tracking of per-track keypoints + camera motion.
The numeric pipeline (gray / resize / GFTT / FAST / pyramidal LK / RANSAC) runs in flow.hip."""
import logging

import numpy as np

from .runtime import get_context

LOGGER = logging.getLogger(__name__)


class Flow:
    def __init__(self, size,
                 bg_feat_scale_factor=(0.1, 0.1),
                 opt_flow_scale_factor=(0.5, 0.5),
                 feat_density=0.005,
                 feat_dist_factor=0.06,
                 ransac_max_iter=500,
                 ransac_conf=0.99,
                 max_error=100,
                 inlier_thresh=4,
                 bg_feat_thresh=10,
                 obj_feat_params=None,
                 opt_flow_params=None):
        """Parameters / range checks: fastmot/flow.py:17-79.  Deviation from the reference:
        flow.py:92-93 tests `opt_flow_params is None` with the condition inverted (user LK
        parameters are silently ignored and Flow(size) crashes); here given parameters are
        applied and None keeps the defaults -- identical results for cfg/mot.json, whose LK
        parameters equal the defaults."""
        self.size = size
        assert 0 < bg_feat_scale_factor[0] <= 1 and 0 < bg_feat_scale_factor[1] <= 1
        self.bg_feat_scale_factor = bg_feat_scale_factor
        assert 0 < opt_flow_scale_factor[0] <= 1 and 0 < opt_flow_scale_factor[1] <= 1
        self.opt_flow_scale_factor = opt_flow_scale_factor
        assert 0 <= feat_density <= 1
        self.feat_density = feat_density
        assert feat_dist_factor >= 0
        self.feat_dist_factor = feat_dist_factor
        assert ransac_max_iter >= 0
        self.ransac_max_iter = ransac_max_iter
        assert 0 <= ransac_conf <= 1
        self.ransac_conf = ransac_conf
        assert 0 <= max_error <= 255
        self.max_error = max_error
        assert inlier_thresh >= 1
        self.inlier_thresh = inlier_thresh
        assert bg_feat_thresh >= 0
        self.bg_feat_thresh = bg_feat_thresh

        self.obj_feat_params = {"maxCorners": 1000, "qualityLevel": 0.06, "blockSize": 3}
        self.opt_flow_params = {"winSize": (5, 5), "maxLevel": 5, "criteria": (3, 10, 0.03)}
        if obj_feat_params is not None:
            self.obj_feat_params.update(vars(obj_feat_params))
        if opt_flow_params is not None:
            self.opt_flow_params.update(vars(opt_flow_params))

        self.ctx = get_context()
        # background feature points for visualization
        self.bg_keypoints = None
        self.prev_bg_keypoints = None
        self.frame_rect = np.array([0., 0., round(float(size[0]) - 1.), round(float(size[1]) - 1.)])

    def init(self, frame):
        raise NotImplementedError('flow.hip lands in a later milestone of this round')

    def predict(self, frame, tracks):
        raise NotImplementedError('flow.hip lands in a later milestone of this round')
