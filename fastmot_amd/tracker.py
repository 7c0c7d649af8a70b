"""MultiTracker (API of fastmot/tracker.py:18-422) re-hosted on the batched HIP kernels.

What stays in Python is exactly the bookkeeping that decides track IDs in the reference: dict /
OrderedDict insertion order, CPython set iteration order, list concatenation order
(SURVEY.md section 8c, quirks Q4-Q8).  Every numeric body runs on the GPU:

  apply_kalman  -> one fm_trk_step launch for all tracks            (tracker.py:164-183)
  update        -> fm_find_occluded, fm_assoc_prepare (all pairwise terms, one launch),
                   fm_assoc_stage per cascade stage (cost gather/gate + LAP or greedy kernel),
                   one fm_trk_update_det launch for all matches, one fm_feat_update launch for
                   all running-mean embeddings, one fm_trk_create launch for all new tracks.
"""
from types import SimpleNamespace
from collections import OrderedDict
import itertools
import logging
import os

import numpy as np

from . import _lib
from .track import Track
from .flow import Flow
from .detector import bind_frame
from .kalman_filter import KalmanFilter
from .runtime import get_context
from .utils.setorder import unmatched_order

LOGGER = logging.getLogger(__name__)

_METRICS = {'EUCLIDEAN': _lib.METRIC_EUCLIDEAN, 'COSINE': _lib.METRIC_COSINE}
# 0: every association stage through fm_assoc_stage (launch + wait per stage), nothing enqueued ahead of the embeddings --
# the path of round 5, kept for the A/B and for the tests that compare the two
_HOST_CASCADE = os.environ.get('FASTMOT_HOST_CASCADE', '1') != '0'


def _frame_rect(size):
    # to_tlbr((0, 0, w, h)) of utils/rect.py:49-57
    return np.array([0., 0., round(float(size[0]) - 1.), round(float(size[1]) - 1.)])


class MultiTracker:
    def __init__(self, size, metric,
                 max_age=6,
                 age_penalty=2,
                 motion_weight=0.2,
                 max_assoc_cost=0.9,
                 max_reid_cost=0.45,
                 iou_thresh=0.4,
                 duplicate_thresh=0.8,
                 occlusion_thresh=0.7,
                 conf_thresh=0.5,
                 confirm_hits=1,
                 history_size=50,
                 kalman_filter_cfg=None,
                 flow_cfg=None,
                 gallery_sync=None):
        """Tracks multiple objects with KLT + Kalman filtering and associates detections to
        tracklets by motion and appearance.  Parameters, defaults and range checks follow
        fastmot/tracker.py:19-107.  `gallery_sync` (optional, NOT in the reference): a
        gallery.GallerySync that all-gathers the lost-track galleries of all streams (RCCL); None
        keeps every stream bit-identical to a single-GPU run."""
        self.size = size
        if metric.upper() not in _METRICS:
            raise KeyError(metric.upper())
        self.metric = metric.upper()
        self._metric_id = _METRICS[self.metric]
        assert max_age >= 1
        self.max_age = max_age
        assert age_penalty >= 1
        self.age_penalty = age_penalty
        assert 0 <= motion_weight <= 1
        self.motion_weight = motion_weight
        assert 0 <= max_assoc_cost <= 2
        self.max_assoc_cost = max_assoc_cost
        assert 0 <= max_reid_cost <= 2
        self.max_reid_cost = max_reid_cost
        assert 0 <= iou_thresh <= 1
        self.iou_thresh = iou_thresh
        assert 0 <= duplicate_thresh <= 1
        self.duplicate_thresh = duplicate_thresh
        assert 0 <= occlusion_thresh <= 1
        self.occlusion_thresh = occlusion_thresh
        assert 0 <= conf_thresh <= 1
        self.conf_thresh = conf_thresh
        assert confirm_hits >= 1
        self.confirm_hits = confirm_hits
        assert history_size >= 0
        self.history_size = history_size

        if kalman_filter_cfg is None:
            kalman_filter_cfg = SimpleNamespace()
        if flow_cfg is None:
            flow_cfg = SimpleNamespace()

        self.ctx = get_context()
        self.tracks = {}
        self.hist_tracks = OrderedDict()
        self._prepared = None
        self.kf = KalmanFilter(**vars(kalman_filter_cfg))
        self.flow = Flow(self.size, **vars(flow_cfg))
        self.frame_rect = _frame_rect(self.size)
        self.ctx.set_frame_rect(self.frame_rect)

        self.klt_bboxes = {}
        self.homography = None
        self.gallery_sync = gallery_sync
        self._foreign_slots = []
        self._foreign_key = None
        self._gallery_cache = {}

    # ------------------------------------------------------------------ lifecycle
    def reset(self, dt):
        """Reset the tracker for a new input context (tracker.py:109-119)."""
        self.kf.reset_dt(dt)
        for track in self.hist_tracks.values():
            track.release()
        self.hist_tracks.clear()
        Track._count = 0

    def _clear_tracks(self):
        for track in self.tracks.values():
            track.release()
        self.tracks.clear()

    def _new_tracks(self, frame_id, tlbrs, labels):
        """kf.create + Track() for a batch of detections, in order (one create launch)."""
        if len(tlbrs) == 0:
            return
        new = [Track(frame_id, tlbr, None, label, self.confirm_hits) for tlbr, label in zip(tlbrs, labels)]
        self.kf.create_slots([t.slot for t in new], np.asarray(tlbrs))
        debug = LOGGER.isEnabledFor(logging.DEBUG)
        for trk in new:
            self.tracks[trk.trk_id] = trk
            if debug:
                LOGGER.debug(f"{'Detected:':<14}{trk}")

    def init(self, frame, detections):
        """Initializes the tracker from the detections of the first frame (tracker.py:121-137)."""
        self._clear_tracks()
        self.flow.init(frame)
        tlbrs = np.ascontiguousarray(detections.tlbr, np.float64).reshape(-1, 4)
        self._new_tracks(0, [tlbrs[i] for i in range(len(tlbrs))],
                         [int(l) for l in np.asarray(detections.label).reshape(-1)])

    def track(self, frame):
        """compute_flow + apply_kalman (tracker.py:139-148)."""
        self.compute_flow(frame)
        self.apply_kalman()

    def compute_flow(self, frame):
        """KLT prediction of track boxes + camera motion (tracker.py:150-162)."""
        active_tracks = [track for track in self.tracks.values() if track.active]
        self.klt_bboxes, self.homography = self.flow.predict(frame, active_tracks)
        if self.homography is None:
            # clear tracks when camera motion cannot be estimated
            self._clear_tracks()

    def predict_async(self, frame):
        """compute_flow + apply_kalman of this frame on the library's worker thread (fm_track_predict_async): this
        method marshals on the calling thread and returns a job for `predict_finish`.  Nothing may touch the tracks
        in between.  Same results as compute_flow(frame); apply_kalman()."""
        flow = self.flow
        bind_frame(self.ctx, frame, flow.size)
        items = list(self.tracks.items())
        active, inside, tlbrs, kps, kp_off = flow.marshal([track for _, track in items if track.active])
        pos = {id(track): k for k, track in enumerate(active)}
        n = len(items)
        slots = np.fromiter((track.slot for _, track in items), np.int32, n)
        ages = np.fromiter((track.age for _, track in items), np.int32, n)
        sorted_idx = np.fromiter((pos.get(id(track), -1) for _, track in items), np.int32, n)
        job = self.ctx.track_predict_async(inside, tlbrs, kps, kp_off, flow._params, slots, ages, sorted_idx,
                                           self.age_penalty)
        job.items, job.active = items, active
        return job

    def predict_finish(self, job):
        pred, next_tlbrs, lost = self.ctx.track_predict_wait(job)
        self.klt_bboxes, self.homography = self.flow.scatter(job.active, *pred)
        if self.homography is None:
            # clear tracks when camera motion cannot be estimated
            self._clear_tracks()
            return
        if next_tlbrs is not None:
            self._after_kalman(job.items, next_tlbrs, lost)

    def _after_kalman(self, items, next_tlbrs, lost):
        for (trk_id, track), row in zip(items, list(next_tlbrs)):
            track.bboxes.append(row)
        if lost.any():
            for i in np.flatnonzero(lost).tolist():
                trk_id, track = items[i]
                if track.confirmed:
                    LOGGER.info(f"{'Out:':<14}{track}")
                self._mark_lost(trk_id)

    def apply_kalman(self):
        """Kalman predict + KLT update of every track in one launch (tracker.py:164-183)."""
        items = list(self.tracks.items())
        n = len(items)
        if n == 0:
            return
        slots = np.fromiter((track.slot for _, track in items), np.int32, n)
        klt = np.zeros((n, 4))
        has_klt = np.zeros(n, np.uint8)
        mult = np.ones(n)
        klt_bboxes = self.klt_bboxes
        if klt_bboxes:
            age_penalty = self.age_penalty
            hit = [(i, klt_bboxes[trk_id], track) for i, (trk_id, track) in enumerate(items) if trk_id in klt_bboxes]
            if hit:
                idx = [h[0] for h in hit]
                klt[idx] = [h[1] for h in hit]
                has_klt[idx] = 1
                # give large KLT uncertainty for occluded tracks (large age / low inlier ratio)
                mult[idx] = [max(age_penalty * h[2].age, 1) / h[2].inlier_ratio for h in hit]
        next_tlbrs, lost = self.kf.step_slots(slots, self.homography, klt, has_klt, mult)
        self._after_kalman(items, next_tlbrs, lost)

    # ------------------------------------------------------------------ association
    def _solve(self, stage, solver, trk_ids, rows, det_ids, cols, **kw):
        """Runs one association stage on the device; returns local (row, col) matches and the
        gated mask exactly as scipy.linear_sum_assignment / _greedy_match would order them."""
        if len(rows) == 0 or len(cols) == 0:
            return [], [], []
        m_rows, m_cols, gated, _ = self.ctx.assoc_stage(stage, solver, rows, cols, **kw)
        return m_rows.tolist(), m_cols.tolist(), gated.tolist()

    @staticmethod
    def _assignment_matches(nr, nc, row_ids, col_ids, m_rows, m_cols, gated):
        """utils/matching.py:58-70.  The two set differences run inside @njit in the reference: their iteration order
        is that of Numba's hash set, not CPython's (utils/setorder.py)."""
        unmatched_rows = unmatched_order(nr, m_rows)
        unmatched_cols = unmatched_order(nc, m_cols)
        unmatched_row_ids = [row_ids[row] for row in unmatched_rows]
        unmatched_col_ids = [col_ids[col] for col in unmatched_cols]
        matches = []
        for row, col, is_gated in zip(m_rows, m_cols, gated):
            if not is_gated:
                matches.append((row_ids[row], col_ids[col]))
            else:
                unmatched_row_ids.append(row_ids[row])
                unmatched_col_ids.append(col_ids[col])
        return matches, unmatched_row_ids, unmatched_col_ids

    def _linear_assignment(self, stage, trk_ids, det_ids, row_of, **kw):
        rows = [row_of[t] for t in trk_ids]
        m_rows, m_cols, gated = self._solve(stage, _lib.SOLVER_LAP, trk_ids, rows, det_ids, det_ids, **kw)
        return self._assignment_matches(len(trk_ids), len(det_ids), trk_ids, det_ids, m_rows, m_cols, gated)

    def prepare_detections(self, detections):
        """The part of `update` that depends on the detections only (contiguous copies + find_occluded,
        tracker.py:196).  MOT.step calls it while the ReID network is still running; `update` calls it
        itself otherwise."""
        det_tlbr = np.ascontiguousarray(detections.tlbr, np.float64).reshape(-1, 4)
        det_label = np.ascontiguousarray(detections.label, np.int64).reshape(-1)
        det_conf = np.asarray(detections.conf, np.float64).reshape(-1)
        occluded = self.ctx.find_occluded(det_tlbr, self.occlusion_thresh)
        self._prepared = (detections, det_tlbr, det_label, det_conf, occluded)

    def update_begin(self, detections, embeddings_in_flight=False):
        """The part of `update` that needs no embeddings: track grouping, the row order of the cost matrices and
        the packed arguments of the pairwise-cost launch.  MOT.step calls it while the ReID network is still
        running (after the Kalman step has finished) and hands the result to `update`; nothing may touch the
        tracks in between.

        embeddings_in_flight: the embeddings of exactly these detections are being computed by
        FeatureExtractor.extract_async on this context.  The pairwise-cost kernel is then enqueued right here, ordered
        behind the ReID network on the device (fm_assoc_prepare2), and the arguments of the one-call cascade are packed,
        so that `update` finds the cost terms in page-locked memory a kernel's length after the embeddings."""
        if self._prepared is None or self._prepared[0] is not detections:
            self.prepare_detections(detections)
        _, det_tlbr, det_label, det_conf, occluded_det_mask = self._prepared
        self._prepared = None
        groups = self._group_tracks_by_depth()
        hist_ids = [trk_id for trk_id, track in self.hist_tracks.items()
                    if track.avg_feat.count >= 2]
        row_ids = list(self.tracks.keys()) + hist_ids
        row_of = {trk_id: i for i, trk_id in enumerate(row_ids)}
        foreign = self._exchange_gallery(hist_ids) if self.gallery_sync is not None else []
        assoc_args = trk_feat_f32 = None
        if len(detections) > 0 and (row_ids or foreign):
            row_tracks = [self.tracks[t] if t in self.tracks else self.hist_tracks[t] for t in row_ids]
            # foreign gallery entries (other streams) come after all local rows
            assoc_args = ([t.slot for t in row_tracks] + self._foreign_slots[:len(foreign)],
                          np.array([t.tlbr for t in row_tracks] + [np.zeros(4)] * len(foreign)),
                          [t.label for t in row_tracks] + [e['label'] for e in foreign])
            trk_feat_f32 = [t not in self.tracks for t in row_ids] + [True] * len(foreign)
        pre = dict(detections=detections, det=(det_tlbr, det_label, det_conf, occluded_det_mask), groups=groups,
                   hist_ids=hist_ids, row_ids=row_ids, row_of=row_of, foreign=foreign, assoc_args=assoc_args,
                   trk_feat_f32=trk_feat_f32, armed=False, cascade=None)
        if embeddings_in_flight and assoc_args is not None and _HOST_CASCADE:
            pre['armed'] = True
            if self.ctx.assoc_prepare(self._metric_id, *assoc_args, det_tlbr, det_label, occluded_det_mask,
                                      trk_feat_f32=trk_feat_f32, after_extractor=True):
                pre['cascade'] = self._cascade_pack(pre)
        return pre

    def _cascade_pack(self, pre):
        """Arguments of the one-call cascade (ctx.assoc_cascade): the groups of `_group_tracks_by_depth` as rows of the
        pairwise matrices, Track.active of the confirmed tracks (tracker.py:220-221), the history rows with the labels
        tracker.py:364 gives them (quirk Q4: the first n of ALL history tracks), the thresholds."""
        confirmed_by_depth, unconfirmed = pre['groups']
        row_of, hist_ids, foreign = pre['row_of'], pre['hist_ids'], pre['foreign']
        tracks = self.tracks
        flat = list(itertools.chain.from_iterable(confirmed_by_depth))
        n_rows = len(pre['row_ids'])
        hist_labels = list(itertools.islice((t.label for t in self.hist_tracks.values()), len(hist_ids)))
        return self.ctx.cascade_pack(
            [len(g) for g in confirmed_by_depth], [row_of[t] for t in flat], [tracks[t].active for t in flat],
            [row_of[t] for t in unconfirmed],
            [row_of[t] for t in hist_ids] + list(range(n_rows, n_rows + len(foreign))),
            hist_labels + [e['label'] for e in foreign], pre['det'][2],
            self.motion_weight, self.max_assoc_cost, min(self.max_assoc_cost + 0.1, 1.), 1. - self.iou_thresh,
            self.conf_thresh, self.max_reid_cost)

    def _cascade_host(self, pre):
        """The cascade of `update` in one library call (fm_assoc_cascade) -> the lists `_cascade_stages` returns."""
        w = self.ctx.assoc_cascade(pre['cascade'])
        n1, n2, n3, nu1, nu2, nu3, n_reid, n_inv, n_rest = w[:9]
        row_ids, hist_ids, foreign = pre['row_ids'], pre['hist_ids'], pre['foreign']
        p = _lib.CASCADE_HEADER
        q = p + 2 * (n1 + n2 + n3)
        matches = [(row_ids[r], d) for r, d in zip(w[p:q:2], w[p + 1:q:2])]
        p, q = q, q + nu1 + nu2 + nu3
        u_trk_ids = [row_ids[r] for r in w[p:q]]
        p, q = q, q + 2 * n_reid
        n_hist = len(hist_ids)
        reid = list(zip(w[p:q:2], w[p + 1:q:2]))
        reid_matches = [(hist_ids[r], d) for r, d in reid if r < n_hist]
        foreign_matches = [(foreign[r - n_hist], d) for r, d in reid if r >= n_hist]
        return matches, u_trk_ids, reid_matches, foreign_matches, w[q:q + n_inv + n_rest]

    def update(self, frame_id, detections, embeddings, pre=None):
        """Associates detections to tracklets based on motion and feature embeddings
        (tracker.py:185-293).

        detections : recarray[DET_DTYPE]; embeddings : (N, M) float32 (host) -- if it is the
        array last returned by FeatureExtractor.postprocess the device copy is used directly.
        pre : the result of `update_begin(detections)` when the caller has already run it."""
        ctx = self.ctx
        n_det = len(detections)
        if pre is None or pre['detections'] is not detections:
            pre = self.update_begin(detections)
        det_tlbr, det_label, det_conf, occluded_det_mask = pre['det']
        confirmed_by_depth, unconfirmed = pre['groups']
        hist_ids, row_ids, row_of, foreign = pre['hist_ids'], pre['row_ids'], pre['row_of'], pre['foreign']

        # ---- device: embeddings + every pairwise term of this frame in one launch
        host_cascade = False
        if n_det > 0:
            on_device = embeddings is ctx.device_emb_host and embeddings is not None
            if pre['armed'] and on_device:
                host_cascade = pre['cascade'] is not None          # enqueued by update_begin behind the ReID network
            else:
                if pre['armed']:
                    ctx.synchronize()                              # (the early launch read other embeddings: redo)
                if on_device:
                    ctx.emb_use_device(n_det)
                else:
                    ctx.emb_upload(embeddings)
                    ctx.device_emb_host = None
                if row_ids or foreign:
                    host_cascade = ctx.assoc_prepare(self._metric_id, *pre['assoc_args'], det_tlbr, det_label,
                                                     occluded_det_mask, trk_feat_f32=pre['trk_feat_f32'])
                    host_cascade = host_cascade and _HOST_CASCADE
                    if host_cascade:
                        pre['cascade'] = self._cascade_pack(pre)

        if host_cascade:
            matches, u_trk_ids, reid_matches, foreign_matches, new_det_ids = self._cascade_host(pre)
        else:
            matches, u_trk_ids, reid_matches, foreign_matches, new_det_ids = self._cascade_stages(pre, n_det)

        # rectify matches that may cause duplicate tracks
        matches, u_trk_ids = self._rectify_matches(matches, u_trk_ids, det_tlbr)

        feat_slots, feat_rows = [], []
        info = LOGGER.isEnabledFor(logging.INFO)

        # ---- reinstate matched tracks (one create launch)
        if reid_matches:
            reid_tracks = [self.hist_tracks.pop(trk_id) for trk_id, _ in reid_matches]
            self.kf.create_slots([t.slot for t in reid_tracks],
                                 det_tlbr[[det_id for _, det_id in reid_matches]])
            for track, (trk_id, det_id) in zip(reid_tracks, reid_matches):
                if info:
                    LOGGER.info(f"{'Reidentified:':<14}{track}")
                track.reinstate(frame_id, det_tlbr[det_id], None, embeddings[det_id], on_device=True)
                feat_slots.append(track.slot)
                feat_rows.append(det_id)
                self.tracks[trk_id] = track

        # ---- update matched tracks (one Kalman launch for all matches, set order as reference)
        matches = list(matches)
        if matches:
            m_tracks = [self.tracks[trk_id] for trk_id, _ in matches]
            next_tlbrs, lost = self.kf.update_det_slots([t.slot for t in m_tracks],
                                                        det_tlbr[[det_id for _, det_id in matches]])
            for i, (trk_id, det_id) in enumerate(matches):
                track = m_tracks[i]
                is_valid = not occluded_det_mask[det_id]
                if info and track.hits == self.confirm_hits - 1:
                    LOGGER.info(f"{'Found:':<14}{track}")
                if lost[i]:
                    is_valid = False
                    if info and track.confirmed:
                        LOGGER.info(f"{'Out:':<14}{track}")
                    self._mark_lost(trk_id)
                track.add_detection(frame_id, next_tlbrs[i], None, embeddings[det_id], is_valid,
                                    on_device=True)
                if is_valid:
                    feat_slots.append(track.slot)
                    feat_rows.append(det_id)
        if feat_slots:
            ctx.feat_update(feat_slots, feat_rows)

        # ---- clean up lost tracks
        for trk_id in u_trk_ids:
            track = self.tracks[trk_id]
            track.mark_missed()
            if not track.confirmed:
                if LOGGER.isEnabledFor(logging.DEBUG):
                    LOGGER.debug(f"{'Unconfirmed:':<14}{track}")
                del self.tracks[trk_id]
                track.release()
                continue
            if track.age > self.max_age:
                if info:
                    LOGGER.info(f"{'Lost:':<14}{track}")
                self._mark_lost(trk_id)

        # ---- start new tracks (one create launch), IDs in the reference's order
        self._new_tracks(frame_id, [det_tlbr[d] for d in new_det_ids],
                         [int(det_label[d]) for d in new_det_ids])

        # ---- identities re-identified from ANOTHER stream's gallery (opt-in extension): a new local
        # track seeded with the foreign appearance, tagged with its global identity
        for entry, det_id in foreign_matches if self.gallery_sync is not None else ():
            self._new_tracks(frame_id, [det_tlbr[det_id]], [int(det_label[det_id])])
            trk = next(reversed(self.tracks.values()))
            trk.global_id = (entry['rank'], entry['trk_id'])
            self.gallery_sync.consume(entry['rank'], entry['trk_id'])
            self._foreign_key = None
            ctx.feat_write([trk.slot], entry['feat'][None], [entry['count']])
            trk.avg_feat.count = entry['count']
            ctx.feat_update([trk.slot], [det_id])
            trk.avg_feat.count += 1
            trk.hits = self.confirm_hits

    def _cascade_stages(self, pre, n_det):
        """The association cascade stage by stage on the device (fm_assoc_stage: cost gather + gate kernel, LAP /
        greedy kernel or the host solver per stage) -- the path of problems too large for the one-call host cascade,
        and of `host_lap_elems = 0`.  Returns matches, unmatched track ids, re-identified (history id, detection)
        pairs, foreign-gallery matches and the detections that start new tracks, in the reference's order."""
        det_tlbr, det_label, det_conf, occluded_det_mask = pre['det']
        confirmed_by_depth, unconfirmed = pre['groups']
        hist_ids, row_ids, row_of, foreign = pre['hist_ids'], pre['row_ids'], pre['row_of'], pre['foreign']
        # ---- 1st association: motion + embeddings, tracks with small age are prioritized
        fill_val = min(self.max_assoc_cost + 0.1, 1.)
        matches1 = []
        u_trk_ids1 = []
        u_det_ids = list(range(n_det))
        for depth, trk_ids in enumerate(confirmed_by_depth):
            if len(u_det_ids) == 0:
                u_trk_ids1.extend(itertools.chain.from_iterable(confirmed_by_depth[depth:]))
                break
            if len(trk_ids) == 0:
                continue
            matches, u_trk_ids, u_det_ids = self._linear_assignment(
                _lib.STAGE_MATCHING, trk_ids, u_det_ids, row_of, motion_weight=self.motion_weight,
                max_cost=self.max_assoc_cost, fill_val=fill_val)
            matches1 += matches
            u_trk_ids1 += u_trk_ids

        # ---- 2nd association with IoU
        active = [trk_id for trk_id in u_trk_ids1 if self.tracks[trk_id].active]
        u_trk_ids1 = [trk_id for trk_id in u_trk_ids1 if not self.tracks[trk_id].active]
        matches2, u_trk_ids2, u_det_ids = self._linear_assignment(
            _lib.STAGE_IOU, active, u_det_ids, row_of, max_cost=1. - self.iou_thresh)

        # ---- 3rd association with unconfirmed tracks
        matches3, u_trk_ids3, u_det_ids = self._linear_assignment(
            _lib.STAGE_IOU, unconfirmed, u_det_ids, row_of, max_cost=1. - self.iou_thresh)

        # ---- reID with track history
        u_det_ids = [det_id for det_id in u_det_ids if det_conf[det_id] >= self.conf_thresh]
        valid_u_det_ids = [det_id for det_id in u_det_ids if not occluded_det_mask[det_id]]
        invalid_u_det_ids = [det_id for det_id in u_det_ids if occluded_det_mask[det_id]]

        n_hist = len(hist_ids)
        # quirk Q4 (tracker.py:364): labels come from the first n_hist of ALL history tracks
        hist_labels = list(itertools.islice((t.label for t in self.hist_tracks.values()), n_hist))
        n_rows = len(row_ids)
        m_rows, m_cols, _ = self._solve(_lib.STAGE_REID, _lib.SOLVER_GREEDY, hist_ids,
                                        [row_of[t] for t in hist_ids] + list(range(n_rows, n_rows + len(foreign))),
                                        valid_u_det_ids, valid_u_det_ids, max_cost=self.max_reid_cost,
                                        row_labels=hist_labels + [e['label'] for e in foreign])
        reid_matches = [(hist_ids[r], valid_u_det_ids[c]) for r, c in zip(m_rows, m_cols) if r < n_hist]
        foreign_matches = [(foreign[r - n_hist], valid_u_det_ids[c]) for r, c in zip(m_rows, m_cols) if r >= n_hist]
        taken = set(m_cols)
        reid_u_det_ids = [det_id for c, det_id in enumerate(valid_u_det_ids) if c not in taken]

        matches = list(itertools.chain(matches1, matches2, matches3))
        u_trk_ids = list(itertools.chain(u_trk_ids1, u_trk_ids2, u_trk_ids3))
        return matches, u_trk_ids, reid_matches, foreign_matches, list(itertools.chain(invalid_u_det_ids, reid_u_det_ids))

    def _exchange_gallery(self, hist_ids):
        """Hands the local history {id, label, count, avg feature} to the gallery exchange (an asynchronous RCCL
        all-gather on a side stream, gallery.py) and parks the foreign features it returns -- the result of the
        PREVIOUS exchange -- in device slots.  A history entry's feature is read from the device once, when the
        track enters the history (it does not change there); foreign features are re-written only when the
        foreign set changed."""
        ctx = self.ctx
        cache = self._gallery_cache
        for tid in [t for t in cache if t not in self.hist_tracks]:
            del cache[tid]
        # (a history track that was re-identified and lost again between two exchanges keeps its id but has a new
        # feature: the cached copy is valid only for the count it was read with)
        fresh = [t for t in hist_ids if t not in cache or cache[t][0] != self.hist_tracks[t].avg_feat.count]
        if fresh:
            avg, cnt = ctx.feat_read([self.hist_tracks[t].slot for t in fresh])
            for tid, a, c in zip(fresh, avg, cnt):
                cache[tid] = (int(c), a.copy())
        foreign = self.gallery_sync.exchange([(tid, self.hist_tracks[tid].label, cache[tid][0], cache[tid][1])
                                              for tid in hist_ids])
        key = [(e['rank'], e['trk_id'], e['count']) for e in foreign]
        if key != self._foreign_key:
            while len(self._foreign_slots) < len(foreign):
                self._foreign_slots.append(ctx.slots.alloc())
            if foreign:
                ctx.feat_write(self._foreign_slots[:len(foreign)], np.array([e['feat'] for e in foreign]),
                               [e['count'] for e in foreign])
            self._foreign_key = key
        return foreign

    def _mark_lost(self, trk_id):
        track = self.tracks.pop(trk_id)
        if track.confirmed:
            self.hist_tracks[trk_id] = track
            if len(self.hist_tracks) > self.history_size:
                _, evicted = self.hist_tracks.popitem(last=False)
                evicted.release()
        else:
            track.release()

    def _group_tracks_by_depth(self, group_size=2):
        n_depth = (self.max_age + group_size) // group_size
        confirmed_by_depth = [[] for _ in range(n_depth)]
        unconfirmed = []
        for trk_id, track in self.tracks.items():
            if track.confirmed:
                depth = track.age // group_size
                confirmed_by_depth[depth].append(trk_id)
            else:
                unconfirmed.append(trk_id)
        return confirmed_by_depth, unconfirmed

    def _rectify_matches(self, matches, u_trk_ids, det_tlbr):
        """Merges / swaps an inactive matched track that duplicates an unmatched active one
        (tracker.py:368-401); IoU distances and the greedy solver run on the device."""
        matches, u_trk_ids = set(matches), set(u_trk_ids)
        inactive_matches = [match for match in matches if not self.tracks[match[0]].active]
        u_active = [trk_id for trk_id in u_trk_ids
                    if self.tracks[trk_id].confirmed and self.tracks[trk_id].active]

        n_inactive_matches = len(inactive_matches)
        if n_inactive_matches == 0 or len(u_active) == 0:
            return matches, u_trk_ids

        m_inactive, det_ids = zip(*inactive_matches)
        t_bboxes = np.array([self.tracks[trk_id].tlbr for trk_id in u_active])
        d_bboxes = det_tlbr[list(det_ids)]
        iou_cost = self.ctx.iou_dist(t_bboxes, d_bboxes)
        g_rows, g_cols = self.ctx.greedy(iou_cost, 1. - self.duplicate_thresh)
        dup_matches = [(u_active[r], c) for r, c in zip(g_rows.tolist(), g_cols.tolist())]

        debug = LOGGER.isEnabledFor(logging.DEBUG)
        for u_trk_id, col in dup_matches:
            m_trk_id, det_id = m_inactive[col], det_ids[col]
            t_u_active, t_m_inactive = self.tracks[u_trk_id], self.tracks[m_trk_id]
            if t_m_inactive.end_frame < t_u_active.start_frame:
                if debug:
                    LOGGER.debug(f"{'Merged:':<14}{u_trk_id} -> {m_trk_id}")
                t_m_inactive.merge_continuation(t_u_active)
                u_trk_ids.remove(u_trk_id)
                del self.tracks[u_trk_id]
                t_u_active.release()
            else:
                if debug:
                    LOGGER.debug(f"{'Duplicate:':<14}{m_trk_id} -> {u_trk_id}")
                u_trk_ids.remove(u_trk_id)
                u_trk_ids.add(m_trk_id)
                matches.remove((m_trk_id, det_id))
                matches.add((u_trk_id, det_id))
        return matches, u_trk_ids
