"""Flow (API of fastmot/flow.py:16-264): KLT tracking of per-track keypoints + camera motion.
The numeric pipeline (gray / resize / GFTT / FAST / pyramidal LK / RANSAC) runs in flow.hip."""
import ctypes as C
import logging

import numpy as np

from . import _lib
from .detector import bind_frame
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
                 opt_flow_params=None,
                 gray_coeff_bits=14):
        """Parameters / range checks: fastmot/flow.py:17-79.  Deviation from the reference:
        flow.py:92-93 tests `opt_flow_params is None` with the condition inverted (user LK
        parameters are silently ignored and Flow(size) crashes); here given parameters are
        applied and None keeps the defaults -- identical results for cfg/mot.json, whose LK
        parameters equal the defaults.

        gray_coeff_bits (not in the reference: there it is whatever the installed OpenCV does): fixed-point width of
        cv2.cvtColor(BGR2GRAY) -- 14 (B 1868, G 9617, R 4899: OpenCV's RGB2Gray<uchar> up to the 4.2 era, i.e. the
        4.1.1 the reference's Dockerfile pins) or 15 (3735 / 19235 / 9798: later 4.x).  The two differ by one grey
        level on a fraction of the pixels; DESIGN.md section 7 says what is known about which is right."""
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
        assert gray_coeff_bits in (14, 15)
        self.gray_coeff_bits = gray_coeff_bits

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
        self._opt_scale = np.array(self.opt_flow_scale_factor, np.float32)
        self._bg_scale = np.array(self.bg_feat_scale_factor, np.float32)
        self._configured = False
        self._params = _lib.FlowPredictParams(
            feat_density, feat_dist_factor, (C.c_float * 2)(*self._opt_scale), (C.c_float * 2)(*self._bg_scale),
            float(max_error), int(ransac_max_iter), float(ransac_conf), int(inlier_thresh), int(size[0]), int(size[1]))

    def _configure(self):
        """Allocates the device images / pyramids (flow.py:100-118 preallocates pinned buffers)."""
        ctx = self.ctx
        if getattr(ctx, 'frame_size', None) != tuple(self.size):
            ctx.frame_configure(self.size[0], self.size[1], getattr(ctx, 'ring_size', 0))
        cfg = _lib.FlowCfg()
        cfg.small_w = round(self.opt_flow_scale_factor[0] * self.size[0])
        cfg.small_h = round(self.opt_flow_scale_factor[1] * self.size[1])
        cfg.bg_w = round(self.bg_feat_scale_factor[0] * self.size[0])
        cfg.bg_h = round(self.bg_feat_scale_factor[1] * self.size[1])
        win = self.opt_flow_params['winSize']
        assert win[0] == win[1], 'square LK windows only'
        cfg.win_size = win[0]
        cfg.max_level = self.opt_flow_params['maxLevel']
        crit = self.opt_flow_params['criteria']
        cfg.max_count = crit[1] if crit[0] & 1 else 30        # TermCriteria.COUNT
        cfg.epsilon = crit[2] if crit[0] & 2 else 0.01        # TermCriteria.EPS
        cfg.fast_thresh = self.bg_feat_thresh
        cfg.max_corners = self.obj_feat_params['maxCorners']
        cfg.block_size = self.obj_feat_params['blockSize']
        cfg.quality_level = self.obj_feat_params['qualityLevel']
        cfg.gray_coeff_bits = self.gray_coeff_bits
        ctx.flow_configure(cfg)
        self._configured = True

    def init(self, frame):
        """Preprocesses the first frame to prepare for subsequent `predict` (flow.py:121-133)."""
        if not self._configured:
            self._configure()
        bind_frame(self.ctx, frame, self.size)
        self.ctx.flow_init()
        self.bg_keypoints = np.empty((0, 2), np.float32)
        self.prev_bg_keypoints = np.empty((0, 2), np.float32)

    def predict(self, frame, tracks):
        """Predicts tracklet positions in the next frame and estimates camera motion
        (flow.py:135-264).  Returns ({trk_id: tlbr}, 3x3 homography) or ({}, None) on failure;
        keypoints / inlier ratios of `tracks` are updated in place.

        One library call (fm_flow_predict): gray / pyramid of the new frame, keypoint bookkeeping and
        detection, background keypoints, pyramidal LK, RANSAC camera motion and per-track boxes; this
        method only orders the tracks and scatters the results."""
        ctx = self.ctx
        bind_frame(ctx, frame, self.size)
        tracks, inside, tlbrs, kps, kp_off = self.marshal(tracks)
        out = ctx.flow_predict(inside, tlbrs, kps, kp_off, self._params)
        return self.scatter(tracks, *out)

    def marshal(self, tracks):
        """Orders `tracks` (in place) from closest to farthest and packs what fm_flow_predict reads:
        -> (tracks, inside_tlbr f64[n,4], tlbr f64[n,4], keypoints f32[m,2], keypoint offsets i32[n+1])."""
        empty = np.empty((0, 2), np.float32)
        # same order as `tracks.sort(reverse=True)` with Track.__lt__ (track.py:160-162 compares exactly this tuple),
        # without ~6 Python-level comparisons per track
        tracks.sort(key=lambda t: (t.tlbr[3], -t.age), reverse=True)
        n_trk = len(tracks)
        fr = self.frame_rect
        if n_trk:
            tlbrs = np.array([t.tlbr for t in tracks], np.float64).reshape(n_trk, 4)
            inside = np.concatenate([np.maximum(tlbrs[:, :2], fr[:2]), np.minimum(tlbrs[:, 2:], fr[2:])], axis=1)
            assert (inside[:, 2:] >= inside[:, :2]).all()
            kp_list = [t.keypoints for t in tracks]
            kp_off = np.zeros(n_trk + 1, np.int32)
            np.cumsum([len(k) for k in kp_list], out=kp_off[1:])
            kps = np.concatenate(kp_list).astype(np.float32, copy=False) if kp_off[-1] else empty
        else:
            tlbrs, inside = np.zeros((0, 4)), np.zeros((0, 4))
            kp_off, kps = np.zeros(1, np.int32), empty
        return tracks, inside, tlbrs, kps, kp_off

    def scatter(self, tracks, status, homography, result, est, n_matched, prev, cur, off, bg):
        """Writes the outputs of fm_flow_predict back to the (ordered) tracks; -> ({trk_id: tlbr}, homography)
        or ({}, None) when the camera motion could not be estimated."""
        empty = np.empty((0, 2), np.float32)
        if status != _lib.FLOW_OK:
            self.bg_keypoints = empty
            LOGGER.warning('Camera motion estimation failed')
            return {}, None
        self.prev_bg_keypoints = prev[bg[0]:bg[1]]
        self.bg_keypoints = cur[bg[0]:bg[1]]

        # estimate target bounding boxes: the per-track keypoint arrays are views cut in one call each
        next_bboxes = {}
        if len(tracks):
            cuts = off[1:-1]
            prev_parts, cur_parts = np.split(prev[:off[-1]], cuts), np.split(cur[:off[-1]], cuts)
            counts = (off[1:] - off[:-1]).tolist()
            result, n_matched, est_rows = result.tolist(), n_matched.tolist(), list(est)
            for k, track in enumerate(tracks):
                code = result[k]
                if code == 0:
                    track.keypoints = empty
                    continue
                track.prev_keypoints = prev_parts[k]
                if code == 2:
                    track.keypoints = empty
                    continue
                track.keypoints = cur_parts[k]
                next_bboxes[track.trk_id] = est_rows[k]
                track.inlier_ratio = counts[k] / n_matched[k]
        return next_bboxes, homography
