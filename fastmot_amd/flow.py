"""Flow (API of fastmot/flow.py:16-264): KLT tracking of per-track keypoints + camera motion.
The numeric pipeline (gray / resize / GFTT / FAST / pyramidal LK / RANSAC) runs in flow.hip."""
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
        self._opt_scale = np.array(self.opt_flow_scale_factor, np.float32)
        self._bg_scale = np.array(self.bg_feat_scale_factor, np.float32)
        self._configured = False

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
        keypoints / inlier ratios of `tracks` are updated in place."""
        self.predict_begin(frame, tracks)
        return self.predict_finish()

    def predict_begin(self, frame, tracks):
        """First half of `predict` (flow.py:153-200): images, keypoint bookkeeping / detection and
        background keypoints -- one device round trip.  MOT.step calls the two halves separately so
        that the ReID network can be enqueued in between."""
        ctx = self.ctx
        bind_frame(ctx, frame, self.size)
        ctx.flow_begin()                       # gray + small + pyramid of the new frame (async)

        # order tracks from closest to farthest
        tracks.sort(reverse=True)
        n_trk = len(tracks)
        fr = self.frame_rect
        empty = np.empty((0, 2), np.float32)

        # detect target feature points + background feature points (one device round trip)
        all_prev_pts = []
        if n_trk:
            tlbrs = np.array([t.tlbr for t in tracks], np.float64).reshape(n_trk, 4)
            inside = np.concatenate([np.maximum(tlbrs[:, :2], fr[:2]), np.minimum(tlbrs[:, 2:], fr[2:])], axis=1)
            assert (inside[:, 2] >= inside[:, 0]).all() and (inside[:, 3] >= inside[:, 1]).all()
            kp_off = np.zeros(n_trk + 1, np.int32)
            np.cumsum([len(t.keypoints) for t in tracks], out=kp_off[1:])
            kps = np.concatenate([t.keypoints for t in tracks]).astype(np.float32) if kp_off[-1] else empty
        else:
            tlbrs, inside = np.zeros((0, 4)), np.zeros((0, 4))
            kp_off, kps = np.zeros(1, np.int32), empty
        areas, keep, needy, new_pts, new_off, new_cnt, keypoints = ctx.flow_prepare(
            inside, tlbrs, kps, kp_off, self.feat_density, self.feat_dist_factor)
        for k in range(n_trk):
            if needy[k]:     # only detect new keypoints when too few are propagated
                all_prev_pts.append(new_pts[new_off[k]:new_off[k] + new_cnt[k]].copy())
            else:
                all_prev_pts.append(kps[kp_off[k]:kp_off[k + 1]][keep[kp_off[k]:kp_off[k + 1]]])
        target_ends = np.cumsum([len(p) for p in all_prev_pts]).astype(np.int32) if n_trk else np.zeros(0, np.int32)
        target_begins = np.concatenate([[0], target_ends[:-1]]).astype(np.int32) if n_trk else np.zeros(0, np.int32)

        self._pending = (tracks, tlbrs, all_prev_pts, target_begins, target_ends, keypoints)

    def predict_finish(self):
        """Second half of `predict` (flow.py:201-264): LK matching, camera motion, target boxes."""
        ctx = self.ctx
        tracks, tlbrs, all_prev_pts, target_begins, target_ends, keypoints = self._pending
        self._pending = None
        n_trk = len(tracks)
        empty = np.empty((0, 2), np.float32)
        if len(keypoints) == 0:
            self.bg_keypoints = empty
            ctx.flow_swap()
            LOGGER.warning('Camera motion estimation failed')
            return {}, None
        keypoints = keypoints * (1 / self._bg_scale)
        bg_begin = int(target_ends[-1]) if n_trk else 0
        all_prev_pts.append(keypoints)

        # match features using optical flow (frame buffers are swapped inside)
        all_prev_pts = np.concatenate(all_prev_pts).astype(np.float32)
        scaled_prev_pts = all_prev_pts * self._opt_scale
        all_cur_pts, status, err = ctx.flow_lk(scaled_prev_pts)
        status = status.astype(np.bool_) & (err < self.max_error)
        all_cur_pts[status] = all_cur_pts[status] * (1 / self._opt_scale)

        # camera motion + per-track boxes (RANSAC; host side of the library)
        n_pts = len(all_prev_pts)
        homography, result, est, n_matched, inl = ctx.flow_estimate(
            all_prev_pts, all_cur_pts, status, target_begins, target_ends, bg_begin, max(n_pts - 1, bg_begin),
            tlbrs, self.size, self.ransac_max_iter, self.ransac_conf, self.inlier_thresh)
        if homography is None:
            self.bg_keypoints = empty
            LOGGER.warning('Camera motion estimation failed')
            return {}, None
        bg = slice(bg_begin, n_pts)
        self.prev_bg_keypoints = all_prev_pts[bg][inl[bg]]
        self.bg_keypoints = all_cur_pts[bg][inl[bg]]

        # estimate target bounding boxes
        next_bboxes = {}
        for k, track in enumerate(tracks):
            code = result[k]
            if code == 0:
                track.keypoints = empty
                continue
            sl = slice(target_begins[k], target_ends[k])
            track.prev_keypoints = all_prev_pts[sl][inl[sl]]
            track.keypoints = all_cur_pts[sl][inl[sl]]
            if code == 2:
                track.keypoints = empty
                continue
            next_bboxes[track.trk_id] = est[k].copy()
            track.inlier_ratio = len(track.keypoints) / n_matched[k]
        return next_bboxes, homography
