"""Deterministic synthetic tracking scenes shared by oracle/make_golden.py (which runs the
REFERENCE MultiTracker on them) and the parity tests (which run fastmot_amd on them).

A scene scripts everything upstream of MultiTracker: detections (DET_DTYPE recarray),
L2-normalised embeddings, and a stand-in for Flow.predict that derives KLT boxes / inlier ratios /
a camera homography from the tracker's own active tracks with a hash-seeded RNG, so that two
trackers that behave identically see identical inputs (SURVEY.md section 8d synthetic input).
"""
from types import SimpleNamespace

import numpy as np

DET_DTYPE = np.dtype([('tlbr', float, 4), ('label', int), ('conf', float)], align=True)

SCENES = {
    # name: (n_ids, frame size, n_frames, detector_frame_skip, metric, n_classes, seed)
    's20_skip5_euclid': dict(n_ids=20, size=(1920, 1080), n_frames=61, skip=5, metric='euclidean', n_classes=1, seed=1),
    's50_skip1_cosine': dict(n_ids=50, size=(1920, 1080), n_frames=40, skip=1, metric='cosine', n_classes=1, seed=2),
    's50_skip2_euclid': dict(n_ids=50, size=(1920, 1080), n_frames=60, skip=2, metric='euclidean', n_classes=1, seed=3),
    's300_4k_multiclass': dict(n_ids=300, size=(3840, 2160), n_frames=12, skip=1, metric='euclidean', n_classes=3, seed=4),
    's8_flowfail': dict(n_ids=8, size=(1280, 720), n_frames=30, skip=3, metric='cosine', n_classes=1, seed=5, fail_frame=14),
    # duplicate tracks: an identity is missed for 3 detector frames (its track turns inactive), then detected
    # for a while with a foreign appearance (-> a second track is born on top of it), then with its own
    # appearance again -> MultiTracker._rectify_matches merge / duplicate branches (tracker.py:368-401)
    's30_impostor_skip2': dict(n_ids=30, size=(1920, 1080), n_frames=90, skip=2, metric='euclidean', n_classes=1,
                               seed=7, impostor=True),
    # + a concurrent foreign-looking detection on the same object -> two live tracks on one target ->
    # the "Duplicate" branch as well (seeds chosen so that the reference logs both branches / a re-ID)
    's30_ghosts': dict(n_ids=30, size=(1920, 1080), n_frames=90, skip=2, metric='euclidean', n_classes=1,
                       seed=9, impostor=True, ghosts=True),
    's30_ghosts_b': dict(n_ids=30, size=(1920, 1080), n_frames=90, skip=2, metric='euclidean', n_classes=1,
                         seed=19, impostor=True, ghosts=True),
    's30_ghosts_cosine': dict(n_ids=30, size=(1920, 1080), n_frames=90, skip=2, metric='cosine', n_classes=1,
                              seed=19, impostor=True, ghosts=True),
    # three classes together with the lost-track history: class-wise gating of the re-identification stage,
    # duplicate / merge branches across classes
    's40_multiclass_reid': dict(n_ids=40, size=(1920, 1080), n_frames=80, skip=2, metric='cosine', n_classes=3,
                                seed=23, impostor=True, ghosts=True),
    # life-cycle edges: tracks need 3 hits to be confirmed, die after 4 missed detector frames, and the detector
    # returns NOTHING for 7 consecutive frames (empty association stages, every track ages / is removed / comes
    # back through the re-identification history)
    's16_blackout_confirm3': dict(n_ids=16, size=(1280, 720), n_frames=64, skip=2, metric='euclidean', n_classes=1,
                                  seed=11, blackout=(21, 7), cfg=dict(confirm_hits=3, max_age=4)),
}

TRACKER_CFG = dict(max_age=6, age_penalty=2, motion_weight=0.2, max_assoc_cost=0.8, max_reid_cost=0.6,
                   iou_thresh=0.4, duplicate_thresh=0.8, occlusion_thresh=0.7, conf_thresh=0.5,
                   confirm_hits=1, history_size=50)


def tracker_kwargs(name=None):
    """cfg/mot.json tracker_cfg (reference cfg/mot.json:43-96) as constructor kwargs; a scene may override
    entries through its `cfg` key."""
    kw = dict(TRACKER_CFG)
    if name is not None:
        kw.update(SCENES[name].get('cfg', {}))
    kw['kalman_filter_cfg'] = SimpleNamespace(
        std_factor_acc=2.25, std_offset_acc=78.5, std_factor_det=(0.08, 0.08),
        std_factor_klt=(0.14, 0.14), min_std_det=(4.0, 4.0), min_std_klt=(5.0, 5.0),
        init_pos_weight=5, init_vel_weight=12, vel_coupling=0.6, vel_half_life=2)
    kw['flow_cfg'] = SimpleNamespace(
        bg_feat_scale_factor=(0.1, 0.1), opt_flow_scale_factor=(0.5, 0.5), feat_density=0.005,
        feat_dist_factor=0.06, ransac_max_iter=500, ransac_conf=0.99, max_error=100, inlier_thresh=4,
        bg_feat_thresh=10,
        obj_feat_params=SimpleNamespace(maxCorners=1000, qualityLevel=0.06, blockSize=3),
        opt_flow_params=SimpleNamespace(winSize=(5, 5), maxLevel=5, criteria=(3, 10, 0.03)))
    return kw


class Scene:
    def __init__(self, name):
        cfg = dict(SCENES[name])
        self.name = name
        self.__dict__.update(cfg)
        self.fail_frame = cfg.get('fail_frame')
        rng = np.random.default_rng(self.seed)
        W, H = self.size
        s = W / 1920.
        n = self.n_ids
        w = rng.uniform(40, 90, n) * s
        h = rng.uniform(120, 250, n) * s
        x = rng.uniform(0, W - w)
        y = rng.uniform(0, H - h)
        vel = rng.normal(0, 3 * s, (n, 2))
        self.labels = rng.integers(0, self.n_classes, n) if self.n_classes > 1 else np.ones(n, int)
        base = rng.normal(0, 1, (n, 512))
        self.id_feats = base / np.linalg.norm(base, axis=1, keepdims=True)
        # ground-truth boxes per frame, reflective borders
        self.gt = np.empty((self.n_frames, n, 4))
        pos = np.stack([x, y], 1)
        for f in range(self.n_frames):
            self.gt[f, :, :2] = pos
            self.gt[f, :, 2] = pos[:, 0] + w - 1
            self.gt[f, :, 3] = pos[:, 1] + h - 1
            pos = pos + vel
            for a, lim, sz in ((0, W, w), (1, H, h)):
                lo = pos[:, a] < -0.3 * sz
                hi = pos[:, a] > lim - 0.7 * sz
                vel[lo | hi, a] *= -1
        # occlusion spells: identity not detected for a while (-> lost -> history -> reID)
        self.hidden = np.zeros((self.n_frames, n), bool)
        for i in range(n):
            if rng.random() < 0.25 and self.n_frames > 20:
                start = int(rng.integers(5, self.n_frames - 10))
                self.hidden[start:start + int(rng.integers(3, 9)) * self.skip, i] = True
        self.camera = rng.normal(0, 1, (self.n_frames, 2)) * s   # per-frame camera translation
        # impostor spells (see SCENES): hidden for 3 detector frames, foreign appearance for 1-3 more
        self.impostor = np.zeros((self.n_frames, n), bool)
        self.ghost = np.zeros((self.n_frames, n), bool)
        if cfg.get('impostor'):
            rng2 = np.random.default_rng((self.seed, 77))
            alt = rng2.normal(0, 1, (n, 512))
            self.alt_feats = alt / np.linalg.norm(alt, axis=1, keepdims=True)
            self.hidden[:] = False
            for i in range(n):
                if rng2.random() < 0.6:
                    start = int(rng2.integers(4, self.n_frames - 12 * self.skip))
                    gone = 3 * self.skip
                    fake = int(rng2.integers(1, 4)) * self.skip
                    self.hidden[start:start + gone, i] = True
                    if cfg.get('ghosts') and rng2.random() < 0.5:
                        # a second, foreign-looking detection on top of the identity from 3 detector frames
                        # before it is missed until it is seen again: two concurrent tracks on one object
                        self.ghost[max(start - 3 * self.skip, 1):start + gone + fake, i] = True
                    else:
                        self.impostor[start + gone:start + gone + fake, i] = True
        self._det_cache = {}

    def detections(self, frame_id):
        """(recarray[DET_DTYPE] sorted by label, embeddings float32 [N,512])"""
        if frame_id in self._det_cache:
            return self._det_cache[frame_id]
        start, length = getattr(self, 'blackout', None) or (0, 0)
        if start <= frame_id < start + length:          # the detector sees nothing at all
            self._det_cache[frame_id] = (np.zeros(0, DET_DTYPE).view(np.recarray), np.zeros((0, 512), np.float32))
            return self._det_cache[frame_id]
        rng = np.random.default_rng((self.seed, 1000 + frame_id))
        keep = ~self.hidden[frame_id] & (rng.random(self.n_ids) > 0.08)
        ids = np.flatnonzero(keep)
        boxes = np.rint(self.gt[frame_id, ids] + rng.normal(0, 1, (len(ids), 4)))
        labels = self.labels[ids]
        conf = rng.uniform(0.3, 1, len(ids))
        feats = self.id_feats[ids] + rng.normal(0, 0.02, (len(ids), 512))
        fake = self.impostor[frame_id, ids]
        if fake.any():
            feats[fake] = self.alt_feats[ids[fake]] + rng.normal(0, 0.02, (int(fake.sum()), 512))
        g_ids = np.flatnonzero(self.ghost[frame_id])
        if len(g_ids):
            boxes = np.concatenate([boxes, np.rint(self.gt[frame_id, g_ids] + rng.normal(0, 1.5, (len(g_ids), 4)))])
            labels = np.concatenate([labels, self.labels[g_ids]])
            conf = np.concatenate([conf, rng.uniform(0.6, 1, len(g_ids))])
            feats = np.concatenate([feats, self.alt_feats[g_ids] + rng.normal(0, 0.02, (len(g_ids), 512))])
        # a few false positives with random appearance
        n_fp = int(rng.integers(0, max(2, self.n_ids // 25)))
        W, H = self.size
        if n_fp:
            fx = rng.uniform(0, W - 100, n_fp)
            fy = rng.uniform(0, H - 250, n_fp)
            fb = np.rint(np.stack([fx, fy, fx + rng.uniform(40, 90, n_fp), fy + rng.uniform(120, 250, n_fp)], 1))
            boxes = np.concatenate([boxes, fb])
            labels = np.concatenate([labels, rng.integers(0, self.n_classes, n_fp) if self.n_classes > 1
                                     else np.ones(n_fp, int)])
            conf = np.concatenate([conf, rng.uniform(0.3, 0.7, n_fp)])
            feats = np.concatenate([feats, rng.normal(0, 1, (n_fp, 512))])
        feats = (feats / np.linalg.norm(feats, axis=1, keepdims=True)).astype(np.float32)
        order = np.argsort(labels, kind='stable')      # detections arrive sorted by class id
        dets = np.zeros(len(order), DET_DTYPE).view(np.recarray)
        dets.tlbr = boxes[order]
        dets.label = labels[order]
        dets.conf = conf[order]
        self._det_cache[frame_id] = (dets, np.ascontiguousarray(feats[order]))
        return self._det_cache[frame_id]

    def make_flow(self):
        return ScriptedFlow(self)


class ScriptedFlow:
    """Stand-in for Flow (flow.py:121-264): deterministic KLT boxes from the tracker's own state."""

    def __init__(self, scene):
        self.scene = scene
        self.frame_id = 0
        self.bg_keypoints = np.empty((0, 2), np.float32)
        self.prev_bg_keypoints = np.empty((0, 2), np.float32)

    def init(self, frame):
        self.frame_id = 0

    def predict(self, frame, tracks):
        sc = self.scene
        self.frame_id += 1
        f = self.frame_id
        if sc.fail_frame is not None and f == sc.fail_frame:
            return {}, None                       # camera-motion failure path (flow.py:191-196)
        cam = sc.camera[f]
        H = np.eye(3)
        H[0, 2], H[1, 2] = cam
        H[0, 1], H[1, 0] = 1e-4 * cam[0], -1e-4 * cam[1]
        H[2, 0], H[2, 1] = 1e-7 * cam[1], 1e-7 * cam[0]
        gt_prev, gt_cur = sc.gt[f - 1], sc.gt[f]
        boxes = {}
        for track in tracks:
            rng = np.random.default_rng((sc.seed, f, int(track.trk_id)))
            if rng.random() < 0.12:
                continue                          # KLT lost this target
            t = np.asarray(track.tlbr, float)
            # ground-truth identity that overlaps the track most in the previous frame
            iw = np.minimum(t[2], gt_prev[:, 2]) - np.maximum(t[0], gt_prev[:, 0]) + 1
            ih = np.minimum(t[3], gt_prev[:, 3]) - np.maximum(t[1], gt_prev[:, 1]) + 1
            inter = np.where((iw > 0) & (ih > 0), iw * ih, 0.)
            j = int(np.argmax(inter))
            if inter[j] <= 0:
                continue
            shift = gt_cur[j, :2] - gt_prev[j, :2]
            box = np.rint(np.concatenate([t[:2] + shift, t[2:] + shift]) + rng.normal(0, 0.7, 4))
            boxes[track.trk_id] = box
            track.inlier_ratio = float(rng.uniform(0.45, 1.0))
        return boxes, H


def run_scene(tracker, scene, record_states=True):
    """Drives `tracker` (reference or fastmot_amd MultiTracker) like MOT.step (mot.py:125-168).
    Returns per-frame records: (frame, [(trk_id, tlbr, confirmed, active, age, hits)], hist ids)."""
    tracker.flow = scene.make_flow()
    tracker.reset(1 / 30.)
    records = []
    for frame_id in range(scene.n_frames):
        if frame_id == 0:
            dets, _ = scene.detections(0)
            tracker.init(None, dets)
        elif frame_id % scene.skip == 0:
            tracker.compute_flow(None)
            tracker.apply_kalman()
            dets, embs = scene.detections(frame_id)
            tracker.update(frame_id, dets, embs)
        else:
            tracker.track(None)
        rows = [(int(tid), np.asarray(t.tlbr, float).copy(), bool(t.confirmed), bool(t.active),
                 int(t.age), int(t.hits)) for tid, t in tracker.tracks.items()]
        records.append((frame_id, rows, [int(k) for k in tracker.hist_tracks.keys()]))
    final = None
    if record_states:
        final = {int(tid): tuple(np.array(a, float) for a in t.state) for tid, t in tracker.tracks.items()}
    return records, final


def pack_records(records, final):
    """Flattens run_scene output into arrays for np.savez."""
    out = {}
    rows = []
    hist = []
    for frame_id, trk_rows, hist_ids in records:
        for order, (tid, tlbr, conf, act, age, hits) in enumerate(trk_rows):
            rows.append([frame_id, order, tid, *tlbr, int(conf), int(act), age, hits])
        for order, tid in enumerate(hist_ids):
            hist.append([frame_id, order, tid])
    out['tracks'] = np.array(rows, np.float64).reshape(-1, 11)
    out['hist'] = np.array(hist, np.int64).reshape(-1, 3)
    if final is not None:
        ids = sorted(final)
        out['final_ids'] = np.array(ids, np.int64)
        out['final_mean'] = np.array([final[i][0] for i in ids]).reshape(-1, 8)
        out['final_cov'] = np.array([final[i][1] for i in ids]).reshape(-1, 8, 8)
    return out


# ------------------------------------------------------------------------------------------------
# MOT-level scenes: the same scripted inputs behind the Detector / FeatureExtractor interfaces, so that MOT.step
# itself (mot.py:125-168: schedule, one extractor per class id, _split_bboxes_by_cls, embedding concatenation) can be
# run -- by the reference (oracle/make_golden_mot.py) and by fastmot_amd (tests/test_mot_multiclass_gpu.py).
# ------------------------------------------------------------------------------------------------
class FakeDetector:
    """Detector interface (detector.py:26-43) returning the scene's scripted detections.  It is called on frame 0
    and then on every detector frame, in order: the frame id follows from the call count."""
    scene = None
    state = None

    def __init__(self, size, class_ids=None, *args, **kwargs):
        self.size = size
        self.calls = 0

    def __call__(self, frame):
        self.detect_async(frame)
        return self.postprocess()

    def detect_async(self, frame):
        self.state['frame_id'] = self.calls * self.scene.skip
        self.calls += 1

    def prefetch(self, frame):
        pass

    def postprocess(self):
        dets, _ = self.scene.detections(self.state['frame_id'])
        return dets


class FakeExtractor:
    """FeatureExtractor interface (feature_extractor.py:39-74): embeddings of the boxes it is handed, looked up in the
    scene by box (first unused exact match); no boxes -> np.empty((0, dim)) float64, as the reference returns."""
    scene = None
    state = None
    log = None                    # [(frame_id, extractor index, number of boxes)] -- which extractor got what

    def __init__(self, *args, **kwargs):
        self.index = len(FakeExtractor.instances)
        FakeExtractor.instances.append(self)
        self.rows = []

    instances = []

    @property
    def metric(self):
        return self.scene.metric

    def extract_async(self, frame, tlbrs):
        dets, _ = self.scene.detections(self.state['frame_id'])
        used = self.state.setdefault(('used', self.state['frame_id']), set())
        self.rows = []
        for box in np.asarray(tlbrs, float).reshape(-1, 4):
            for i in range(len(dets)):
                if i not in used and np.array_equal(dets.tlbr[i], box):
                    used.add(i)
                    self.rows.append(i)
                    break
            else:
                raise AssertionError('box handed to the extractor is not a detection of this frame')
        FakeExtractor.log.append((self.state['frame_id'], self.index, len(self.rows)))

    def postprocess(self):
        _, embs = self.scene.detections(self.state['frame_id'])
        if not self.rows:
            return np.empty((0, embs.shape[1]))
        return embs[self.rows]

    def __call__(self, frame, tlbrs):
        self.extract_async(frame, tlbrs)
        return self.postprocess()


def bind_fakes(scene):
    state = {}
    FakeDetector.scene = FakeExtractor.scene = scene
    FakeDetector.state = FakeExtractor.state = state
    FakeExtractor.instances = []
    FakeExtractor.log = []
    return state


def run_scene_mot(mot, scene, frame_for=lambda f: f):
    """Drives a MOT object (reference or fastmot_amd, built with the fakes above) over the scene; same record
    format as run_scene."""
    mot.tracker.flow = scene.make_flow()
    mot.reset(1 / 30.)
    records = []
    for frame_id in range(scene.n_frames):
        mot.step(frame_for(frame_id))
        tracker = mot.tracker
        rows = [(int(tid), np.asarray(t.tlbr, float).copy(), bool(t.confirmed), bool(t.active),
                 int(t.age), int(t.hits)) for tid, t in tracker.tracks.items()]
        records.append((frame_id, rows, [int(k) for k in tracker.hist_tracks.keys()]))
    return records
