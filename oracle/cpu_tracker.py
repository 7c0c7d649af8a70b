"""TEST INFRASTRUCTURE ONLY -- CPU (numpy) restatement of the tracker side of the hot path:
MultiTracker (fastmot/tracker.py:18-401), Track/AverageFeature (fastmot/track.py:91-225) and Flow
(fastmot/flow.py:121-264, with the OpenCV calls restated in cv_oracle.py).

Used (a) as an end-to-end checker that is itself pinned against the reference goldens
(tests/test_oracle_golden.py::test_tracker_scenes) and (b) as bench.py's `cpu_baseline` leg
(kind "port").  Never imported by the product package.

Set / dict / list ordering follows the reference expression by expression, because it decides the
track IDs."""
from collections import OrderedDict, deque
import itertools

import numpy as np

import cv_oracle as cv
import np_oracle as o


class OTrack:
    count = 0

    def __init__(self, frame_id, tlbr, state, label, confirm_hits=1, buffer_size=30):
        OTrack.count += 1
        self.trk_id = OTrack.count
        self.start_frame = frame_id
        self.frame_ids = deque([frame_id], maxlen=buffer_size)
        self.bboxes = deque([tlbr], maxlen=buffer_size)
        self.confirm_hits, self.state, self.label = confirm_hits, state, label
        self.age = self.hits = 0
        self.f_sum = self.f_avg = None
        self.f_count = 0
        self.inlier_ratio = 1.
        self.keypoints = np.empty((0, 2), np.float32)
        self.prev_keypoints = np.empty((0, 2), np.float32)

    tlbr = property(lambda self: self.bboxes[-1])
    end_frame = property(lambda self: self.frame_ids[-1])
    active = property(lambda self: self.age < 2)
    confirmed = property(lambda self: self.hits >= self.confirm_hits)

    def __lt__(self, other):
        return (self.tlbr[-1], -self.age) < (other.tlbr[-1], -other.age)

    def feat_update(self, emb):
        self.f_count += 1
        if self.f_sum is None:
            self.f_sum, self.f_avg = emb.copy(), emb.copy()
        else:
            self.f_sum, self.f_avg = o.average_feature(self.f_sum, emb, self.f_count)

    def feat_merge(self, other):
        self.f_count += other.f_count
        if self.f_sum is None:
            self.f_sum, self.f_avg = other.f_sum, other.f_avg
        elif other.f_sum is not None:
            self.f_sum, self.f_avg = o.average_feature(self.f_sum, other.f_sum, self.f_count)


# ---- the @njit glue of Flow.predict (flow.py:266-364), as OracleFlow.predict uses it; pinned against the jit-compiled
# reference by tests/test_flow_helpers_kat.py (fixture from oracle/pin_with_numba.py)
def rect_filter(kp, ins, fg):
    """flow.py:283-295: keypoints whose rounded position lies inside the rectangle and on foreground"""
    if len(kp) == 0:
        return np.empty((0, 2), np.float32)
    p2 = np.rint(kp).astype(np.int32)
    ok = (p2[:, 0] >= ins[0]) & (p2[:, 0] <= ins[2]) & (p2[:, 1] >= ins[1]) & (p2[:, 1] <= ins[3])
    kp, p2 = kp[ok], p2[ok]
    return kp[fg[p2[:, 1], p2[:, 0]] == 255] if len(kp) else kp


def feature_dist(area, factor):
    """flow.py:268-271"""
    return max(round(np.sqrt(area) * factor), 1)


def ellipse_filter(kp_local, tlbr, offset):
    """flow.py:298-307: crop-local corners -> frame coordinates (float32), those inside the box's inscribed ellipse"""
    kp = kp_local.reshape(-1, 2) + np.asarray(offset, np.float32)
    c = (tlbr[:2] + tlbr[2:]) / 2
    ax = (tlbr[2:] - tlbr[:2] + 1) * 0.5
    return kp[(((kp - c) / ax) ** 2).sum(1) <= 1.]


def lk_status(st, err, max_error):
    """flow.py:348-350"""
    return st.ravel().astype(bool) & (err.ravel() < max_error)


def unscale_pts(pts, scale, mask=None):
    """flow.py:335-345 (float32 throughout: 1 / float32 array stays float32 under Numba as under NumPy)"""
    un = 1 / np.array(scale, np.float32)
    pts = pts.reshape(-1, 2)
    if mask is None:
        return pts * un
    pts[mask] = pts[mask] * un
    return pts


class OracleFlow:
    """Flow.init / Flow.predict on real frames with cv_oracle (flow.py:121-264)."""

    def __init__(self, size, bg_scale=(0.1, 0.1), opt_scale=(0.5, 0.5), feat_density=0.005, feat_dist_factor=0.06,
                 ransac_max_iter=500, ransac_conf=0.99, max_error=100, inlier_thresh=4, bg_feat_thresh=10, cv_impl=None,
                 gray_coeff_bits=None):
        self.cv = cv_impl if cv_impl is not None else cv      # cv_oracle (numpy) or c_baseline (compiled C)
        self.gray_bits = gray_coeff_bits                      # None = cv_oracle.GRAY_COEFF_BITS (14)
        self.size = size
        self.bg_scale, self.opt_scale = bg_scale, opt_scale
        self.feat_density, self.feat_dist_factor = feat_density, feat_dist_factor
        self.ransac_max_iter, self.ransac_conf = ransac_max_iter, ransac_conf
        self.max_error, self.inlier_thresh, self.bg_feat_thresh = max_error, inlier_thresh, bg_feat_thresh
        self.small_sz = (round(opt_scale[0] * size[0]), round(opt_scale[1] * size[1]))
        self.bg_sz = (round(bg_scale[0] * size[0]), round(bg_scale[1] * size[1]))
        self.frame_rect = np.array([0., 0., size[0] - 1., size[1] - 1.])
        self.bg_keypoints = self.prev_bg_keypoints = np.empty((0, 2), np.float32)

    def init(self, frame):
        self.prev_gray = self.cv.bgr2gray(frame, self.gray_bits)
        self.prev_small = self.cv.resize_linear_u8(self.prev_gray, self.small_sz)

    def predict(self, frame, tracks):
        gray = self.cv.bgr2gray(frame, self.gray_bits)
        small = self.cv.resize_linear_u8(gray, self.small_sz)
        tracks.sort(reverse=True)
        empty = np.empty((0, 2), np.float32)
        fg = np.full(gray.shape, 255, np.uint8)
        all_prev = []
        fr = self.frame_rect
        for t in tracks:
            ins = np.concatenate([np.maximum(t.tlbr[:2], fr[:2]), np.minimum(t.tlbr[2:], fr[2:])]).astype(int)
            tm = fg[ins[1]:ins[3] + 1, ins[0]:ins[2] + 1]
            area = int((tm != 0).sum())
            kp = rect_filter(t.keypoints, ins, fg) if len(t.keypoints) else t.keypoints
            if len(kp) < self.feat_density * area:
                md = feature_dist(area, self.feat_dist_factor)
                kp = self.cv.good_features_to_track(self.prev_gray[ins[1]:ins[3] + 1, ins[0]:ins[2] + 1], tm, 1000, 0.06, md)
                if len(kp):
                    kp = ellipse_filter(kp, t.tlbr, ins[:2])
            all_prev.append(kp.astype(np.float32).reshape(-1, 2))
            tm[:] = 0
        ends = np.cumsum([len(p) for p in all_prev]).astype(np.int32) if tracks else np.zeros(0, np.int32)
        begins = np.concatenate([[0], ends[:-1]]).astype(np.int32) if tracks else np.zeros(0, np.int32)
        bg_img = self.cv.resize_linear_u8(self.prev_gray, self.bg_sz)
        mask_small = self.cv.resize_nearest(fg, self.bg_sz)
        kp = self.cv.fast_detect(bg_img, self.bg_feat_thresh)
        kp = kp[[mask_small[int(p[1] + 0.5), int(p[0] + 0.5)] != 0 for p in kp]] if len(kp) else kp
        if len(kp) == 0:
            self.bg_keypoints = empty
            self.prev_gray, self.prev_small = gray, small
            return {}, None
        kp = unscale_pts(kp, self.bg_scale)
        bg_begin = int(ends[-1]) if tracks else 0
        all_prev.append(kp)
        P = np.concatenate(all_prev).astype(np.float32)
        sp = P * np.array(self.opt_scale, np.float32)
        C, st, err = self.cv.calc_optical_flow_pyr_lk(self.prev_small, small, sp)
        st = lk_status(st, err, self.max_error)
        C = unscale_pts(C, self.opt_scale, st)
        self.prev_gray, self.prev_small = gray, small
        tl = np.array([t.tlbr for t in tracks], float).reshape(-1, 4)
        H, res, est, nm, inl = self.cv.flow_estimate(P, C, st, begins, ends, bg_begin, max(len(P) - 1, bg_begin), tl,
                                                self.size, self.ransac_max_iter, self.ransac_conf, self.inlier_thresh)
        if H is None:
            self.bg_keypoints = empty
            return {}, None
        self.prev_bg_keypoints, self.bg_keypoints = P[bg_begin:][inl[bg_begin:]], C[bg_begin:][inl[bg_begin:]]
        boxes = {}
        for k, t in enumerate(tracks):
            if res[k] == 0:
                t.keypoints = empty
                continue
            sl = slice(begins[k], ends[k])
            t.prev_keypoints, t.keypoints = P[sl][inl[sl]], C[sl][inl[sl]]
            if res[k] == 2:
                t.keypoints = empty
                continue
            boxes[t.trk_id] = est[k].copy()
            t.inlier_ratio = len(t.keypoints) / nm[k]
        return boxes, H


class OracleTracker:
    def __init__(self, size, metric, max_age=6, age_penalty=2, motion_weight=0.2, max_assoc_cost=0.9,
                 max_reid_cost=0.45, iou_thresh=0.4, duplicate_thresh=0.8, occlusion_thresh=0.7, conf_thresh=0.5,
                 confirm_hits=1, history_size=50, kalman_filter_cfg=None, flow_cfg=None, cv_impl=None):
        self.size, self.metric = size, metric.lower()
        self.max_age, self.age_penalty, self.motion_weight = max_age, age_penalty, motion_weight
        self.max_assoc_cost, self.max_reid_cost, self.iou_thresh = max_assoc_cost, max_reid_cost, iou_thresh
        self.duplicate_thresh, self.occlusion_thresh, self.conf_thresh = duplicate_thresh, occlusion_thresh, conf_thresh
        self.confirm_hits, self.history_size = confirm_hits, history_size
        self.p = o.KFParams(**(vars(kalman_filter_cfg) if kalman_filter_cfg is not None else {}))
        self.tracks, self.hist_tracks = {}, OrderedDict()
        self.frame_rect = np.array([0., 0., size[0] - 1., size[1] - 1.])
        self.flow = OracleFlow(size, cv_impl=cv_impl)
        self.klt_bboxes, self.homography = {}, None

    def reset(self, dt):
        self.p.reset_dt(dt)
        self.hist_tracks.clear()
        OTrack.count = 0

    def _create(self, box):
        m, c = o.kf_create(self.p, box)
        return m[0], c[0]

    def init(self, frame, detections):
        self.tracks.clear()
        self.flow.init(frame)
        for det in detections:
            t = OTrack(0, det.tlbr, self._create(det.tlbr), det.label, self.confirm_hits)
            self.tracks[t.trk_id] = t

    def track(self, frame):
        self.compute_flow(frame)
        self.apply_kalman()

    def compute_flow(self, frame):
        active = [t for t in self.tracks.values() if t.active]
        self.klt_bboxes, self.homography = self.flow.predict(frame, active)
        if self.homography is None:
            self.tracks.clear()

    def apply_kalman(self):
        items = list(self.tracks.items())
        if not items:
            return
        mean = np.array([t.state[0] for _, t in items])
        cov = np.array([t.state[1] for _, t in items])
        mean, cov = o.kf_warp(mean, cov, self.homography)
        mean, cov = o.kf_predict(self.p, mean, cov)
        for i, (tid, t) in enumerate(items):
            m, c = mean[i], cov[i]
            if tid in self.klt_bboxes:
                mult = max(self.age_penalty * t.age, 1) / t.inlier_ratio
                mu, cu = o.kf_update(self.p, m, c, self.klt_bboxes[tid], 'flow', mult)
                m, c = mu[0], cu[0]
            box = np.rint(m[:4])
            t.bboxes.append(box)
            t.state = (m, c)
            if o.ios(box, self.frame_rect)[0] < 0.5:
                self._mark_lost(tid)

    def _mark_lost(self, tid):
        t = self.tracks.pop(tid)
        if t.confirmed:
            self.hist_tracks[tid] = t
            if len(self.hist_tracks) > self.history_size:
                self.hist_tracks.popitem(last=False)

    def _lap(self, cost, row_ids, col_ids):
        r, c = o.lsa(cost)
        m, ur, uc = o.assignment_matches(cost, r, c)
        return [(row_ids[a], col_ids[b]) for a, b in m], [row_ids[a] for a in ur], [col_ids[b] for b in uc]

    def _matching_cost(self, trk_ids, dets, embs, occ):
        nt, nd = len(trk_ids), len(dets)
        if nt == 0 or nd == 0:
            return np.empty((nt, nd))
        feats = np.zeros((nt, embs.shape[1]))
        invalid = np.zeros(nt, bool)
        for i, tid in enumerate(trk_ids):
            t = self.tracks[tid]
            if t.f_count > 0:
                feats[i] = t.f_avg
            else:
                invalid[i] = True
        fill = min(self.max_assoc_cost + 0.1, 1.)
        fd = o.cdist(feats, embs, self.metric, invalid[:, None] | occ, fill)
        mean = np.array([self.tracks[t].state[0] for t in trk_ids])
        cov = np.array([self.tracks[t].state[1] for t in trk_ids])
        maha = o.kf_maha(self.p, mean, cov, dets.tlbr)
        labels = [self.tracks[t].label for t in trk_ids]
        return o.matching_cost(fd, maha, labels, dets.label, self.motion_weight, self.max_assoc_cost)

    def _iou_cost(self, trk_ids, dets):
        nt, nd = len(trk_ids), len(dets)
        if nt == 0 or nd == 0:
            return np.empty((nt, nd))
        tb = np.array([self.tracks[t].tlbr for t in trk_ids])
        return o.gate_cost(o.iou_dist(tb, dets.tlbr), [self.tracks[t].label for t in trk_ids], dets.label,
                           1. - self.iou_thresh)

    def update(self, frame_id, detections, embeddings):
        occ = o.find_occluded(detections.tlbr, self.occlusion_thresh)
        n_depth = (self.max_age + 2) // 2
        by_depth = [[] for _ in range(n_depth)]
        unconfirmed = []
        for tid, t in self.tracks.items():
            (by_depth[t.age // 2] if t.confirmed else unconfirmed).append(tid)
        matches1, u_trk1 = [], []
        u_det = list(range(len(detections)))
        for depth, ids in enumerate(by_depth):
            if len(u_det) == 0:
                u_trk1.extend(itertools.chain.from_iterable(by_depth[depth:]))
                break
            if len(ids) == 0:
                continue
            cost = self._matching_cost(ids, detections[u_det], embeddings[u_det], occ[u_det])
            m, ut, u_det = self._lap(cost, ids, u_det)
            matches1 += m
            u_trk1 += ut
        active = [t for t in u_trk1 if self.tracks[t].active]
        u_trk1 = [t for t in u_trk1 if not self.tracks[t].active]
        matches2, u_trk2, u_det = self._lap(self._iou_cost(active, detections[u_det]), active, u_det)
        matches3, u_trk3, u_det = self._lap(self._iou_cost(unconfirmed, detections[u_det]), unconfirmed, u_det)
        hist_ids = [tid for tid, t in self.hist_tracks.items() if t.f_count >= 2]
        u_det = [d for d in u_det if detections[d].conf >= self.conf_thresh]
        valid = [d for d in u_det if not occ[d]]
        invalid = [d for d in u_det if occ[d]]
        if len(hist_ids) and len(valid):
            feats = np.concatenate([self.hist_tracks[t].f_avg for t in hist_ids]).reshape(len(hist_ids), -1)
            cost = o.cdist(feats, embeddings[valid], self.metric)
            labels = list(itertools.islice((t.label for t in self.hist_tracks.values()), len(hist_ids)))
            cost = o.gate_cost(cost, labels, detections[valid].label)
        else:
            cost = np.empty((len(hist_ids), len(valid)))
        gm, _, gu = o.greedy_match(cost, self.max_reid_cost)
        reid_matches = [(hist_ids[a], valid[b]) for a, b in gm]
        reid_u_det = [valid[b] for b in gu]
        matches, u_trk = self._rectify(itertools.chain(matches1, matches2, matches3),
                                       itertools.chain(u_trk1, u_trk2, u_trk3), detections)
        for tid, d in reid_matches:
            t = self.hist_tracks.pop(tid)
            det = detections[d]
            t.start_frame = frame_id
            t.frame_ids.append(frame_id)
            t.bboxes.append(det.tlbr)
            t.state = self._create(det.tlbr)
            t.feat_update(embeddings[d])
            t.age = 0
            t.keypoints = t.prev_keypoints = np.empty((0, 2), np.float32)
            self.tracks[tid] = t
        for tid, d in matches:
            t = self.tracks[tid]
            det = detections[d]
            m, c = o.kf_update(self.p, t.state[0], t.state[1], det.tlbr, 'detector')
            box = np.rint(m[0, :4])
            is_valid = not occ[d]
            if o.ios(box, self.frame_rect)[0] < 0.5:
                is_valid = False
                self._mark_lost(tid)
            t.frame_ids.append(frame_id)
            t.bboxes.append(box)
            t.state = (m[0], c[0])
            if is_valid:
                t.feat_update(embeddings[d])
            t.age = 0
            t.hits += 1
        for tid in u_trk:
            t = self.tracks[tid]
            t.age += 1
            if not t.confirmed:
                del self.tracks[tid]
                continue
            if t.age > self.max_age:
                self._mark_lost(tid)
        for d in itertools.chain(invalid, reid_u_det):
            det = detections[d]
            t = OTrack(frame_id, det.tlbr, self._create(det.tlbr), det.label, self.confirm_hits)
            self.tracks[t.trk_id] = t

    def _rectify(self, matches, u_trk, detections):
        matches, u_trk = set(matches), set(u_trk)
        inactive = [m for m in matches if not self.tracks[m[0]].active]
        u_active = [t for t in u_trk if self.tracks[t].confirmed and self.tracks[t].active]
        if len(inactive) == 0 or len(u_active) == 0:
            return matches, u_trk
        m_inactive, det_ids = zip(*inactive)
        tb = np.array([self.tracks[t].tlbr for t in u_active])
        cost = o.iou_dist(tb, detections[det_ids,].tlbr)
        gm, _, _ = o.greedy_match(cost, 1. - self.duplicate_thresh)
        for r, col in gm:
            u_id, m_id, d = u_active[r], m_inactive[col], det_ids[col]
            tu, tm = self.tracks[u_id], self.tracks[m_id]
            if tm.end_frame < tu.start_frame:
                tm.frame_ids.extend(tu.frame_ids)
                tm.bboxes.extend(tu.bboxes)
                tm.state, tm.age = tu.state, tu.age
                tm.hits += tu.hits
                tm.keypoints, tm.prev_keypoints = tu.keypoints, tu.prev_keypoints
                tm.feat_merge(tu)
                u_trk.remove(u_id)
                del self.tracks[u_id]
            else:
                u_trk.remove(u_id)
                u_trk.add(m_id)
                matches.remove((m_id, d))
                matches.add((u_id, d))
        return matches, u_trk
