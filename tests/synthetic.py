"""Seeded synthetic workload (BASELINE.md section 4 / SURVEY.md section 8d): 1080p BGR frames with a
textured background under a slow camera drift and K textured rectangles moving with constant
velocity, plus the matching detections.  Used by bench.py and the end-to-end tests; no datasets or
weights are available offline, so detections are injected at the Detector.postprocess() boundary
while the detector network, decode and NMS still execute on the GPU."""
import numpy as np

from fastmot_amd.detector import YOLODetector, DET_DTYPE
from fastmot_amd.models.graph import RandomWeights


class SyntheticVideo:
    def __init__(self, size=(1920, 1080), n_ids=50, n_frames=64, seed=0, block=6):
        self.size = size
        self.n_ids, self.n_frames = n_ids, n_frames
        rng = np.random.default_rng(seed)
        W, H = size
        s = W / 1920.
        margin = 64
        bh, bw = (H + 2 * margin) // block + 2, (W + 2 * margin) // block + 2
        bg = np.kron(rng.integers(30, 226, (bh, bw, 3)).astype(np.uint8), np.ones((block, block, 1), np.uint8))
        bg = np.clip(bg.astype(np.int16) + rng.integers(-14, 15, bg.shape), 0, 255).astype(np.uint8)
        w = rng.uniform(40, 90, n_ids) * s
        h = rng.uniform(120, 250, n_ids) * s
        pos = np.stack([rng.uniform(0, W - w), rng.uniform(0, H - h)], 1)
        vel = rng.normal(0, 3 * s, (n_ids, 2))
        patches = []
        for i in range(n_ids):
            ph, pw = int(round(h[i])), int(round(w[i]))
            p = np.kron(rng.integers(0, 256, (ph // 5 + 2, pw // 5 + 2, 3)).astype(np.uint8), np.ones((5, 5, 1), np.uint8))
            patches.append(np.ascontiguousarray(p[:ph, :pw]))
        cam = np.cumsum(rng.normal(0, 1.0 * s, (n_frames, 2)), axis=0)
        cam = np.clip(np.rint(cam), -margin + 2, margin - 2).astype(int)
        self.frames, self.gt = [], np.empty((n_frames, n_ids, 4))
        for f in range(n_frames):
            ox, oy = margin + cam[f, 0], margin + cam[f, 1]
            frame = bg[oy:oy + H, ox:ox + W].copy()
            order = np.argsort(pos[:, 1] + h)            # farthest (smallest bottom y) first
            for i in order:
                ph, pw = patches[i].shape[:2]
                x0, y0 = int(round(pos[i, 0])), int(round(pos[i, 1]))
                xs, ys = max(x0, 0), max(y0, 0)
                xe, ye = min(x0 + pw, W), min(y0 + ph, H)
                if xe > xs and ye > ys:
                    frame[ys:ye, xs:xe] = patches[i][ys - y0:ye - y0, xs - x0:xe - x0]
                self.gt[f, i] = (x0, y0, x0 + pw - 1, y0 + ph - 1)
            self.frames.append(frame)
            pos = pos + vel
            for a, lim, sz in ((0, W, w), (1, H, h)):
                flip = (pos[:, a] < 0) | (pos[:, a] > lim - sz)
                vel[flip, a] *= -1
                pos[:, a] = np.clip(pos[:, a], 0, lim - sz)
        self._rng_seed = seed

    def detections(self, frame_idx, label=1, labels=None):
        """Exactly n_ids detections: ground truth + N(0, 1) px jitter, conf U(0.5, 1).  `labels` (several class
        ids): object i has class labels[i % len(labels)] and the detections come sorted by class id, as
        YOLODetector._filter_dets delivers them (detector.py:344)."""
        rng = np.random.default_rng((self._rng_seed, 7, frame_idx))
        dets = np.zeros(self.n_ids, DET_DTYPE).view(np.recarray)
        dets.tlbr = np.rint(self.gt[frame_idx % self.n_frames] + rng.normal(0, 1, (self.n_ids, 4)))
        dets.label = label
        dets.conf = rng.uniform(0.5, 1, self.n_ids)
        if labels is not None and len(labels) > 1:
            lab = np.sort(np.asarray(labels))[np.arange(self.n_ids) % len(labels)]
            order = np.argsort(lab, kind='stable')
            dets = dets[order]
            dets.label = lab[order]
        return dets


NET_TIMING_EVERY = 4     # bench / traces: every 4th detector pass carries the HIP-event pair (each pair costs ~1 % of the frame rate)


class InjectedYOLODetector(YOLODetector):
    """YOLODetector whose postprocess() waits for the real GPU pipeline (network + decode + NMS
    on the seeded-random weights) and then returns the scripted detections of the synthetic video."""

    def bind_video(self, video, label=1, labels=None):
        self._video, self._label = video, label
        self._labels = labels if labels is not None and len(labels) > 1 else None
        if labels is not None and len(labels) == 1:
            self._label = labels[0]
        self._frame_idx = 0
        self.last_real_count = self.last_candidates = 0
        self.net_ms = []          # HIP-event time of the detector's layer sequence, one entry per timed pass
        self.ctx.set_option('net_timing', NET_TIMING_EVERY)

    def detect_async(self, frame):
        super().detect_async(frame)

    def postprocess(self):
        real = super().postprocess()
        self.last_real_count = len(real)
        self.last_candidates = self.ctx.detect_last_counts()[0]
        ms = self.ctx.detect_net_ms()                      # (None: this pass carried no timing events, option 'net_timing')
        if ms is not None:
            self.net_ms.append(ms)                         # the events of THIS frame's network are complete here
        dets = self._video.detections(self._frame_idx, self._label, self._labels)
        self._frame_idx += 1
        return dets


class ScriptedHeadWeights(RandomWeights):
    """The seeded random parameters of `RandomWeights(seed)` (same random stream: every other layer is unchanged) with
    the YOLO head convolutions scripted so that a chosen share of the candidate boxes passes the detector's confidence
    threshold with class `label`: with purely random heads nothing passes `conf_thresh` and the DIoU-NMS stage of a
    benchmark would run on an empty list.  The class logits get biases of +-4; the objectness rows of the weight are
    amplified by `obj_gain` (the random head's logits are all but constant over the image) and get the per-anchor
    biases obj_bias[head][anchor] (heads in build order).  `label` may be a tuple of class ids: anchor a of every head
    then votes for label[a % len(label)] (several classes among the candidates, BASELINE config[4])."""

    def __init__(self, seed, num_classes, label, obj_bias, obj_gain=1.0):
        super().__init__(seed)
        self.num_classes, self.label, self.obj_bias, self.obj_gain = num_classes, label, obj_bias, obj_gain
        self._head = 0

    def conv(self, name, cout, cin, k, bn=True, gain=1.0, groups=1):
        p = super().conv(name, cout, cin, k, bn=bn, gain=gain, groups=groups)
        rec = 5 + self.num_classes
        if not bn and cout % rec == 0:
            b = p['bias'].reshape(-1, rec)
            b[:, 4] = np.asarray(self.obj_bias[self._head], np.float32)
            b[:, 5:] = -4.0
            labels = self.label if isinstance(self.label, (tuple, list)) else (self.label,)
            for a in range(len(b)):
                b[a, 5 + labels[a % len(labels)]] = 4.0
            p['w'].reshape(-1, rec, *p['w'].shape[1:])[:, 4] *= self.obj_gain
            self._head += 1
        return p


def scripted_head_weights(size, model, label, frame, target=1500, conf_thresh=0.25, seed=0, obj_gain=40.0):
    """ScriptedHeadWeights for `model` whose heads let about `target` candidates per frame through `conf_thresh` on
    frames like `frame`: one calibration pass with an objectness bias that lets nothing through, then the bias of every
    (head, anchor) is set to the quantile of its objectness logits that leaves its share of the target above the
    threshold."""
    from fastmot_amd import models
    m = models.YOLO.get_model(model)
    n_anchors = [len(a) // 2 for a in m.ANCHORS]
    labels = tuple(label) if isinstance(label, (tuple, list)) else (label,)
    # class probability ~ sigmoid(4) = 0.982: box_conf * cls_prob >= thr  <=>  objectness logit >= logit(thr / 0.982)
    need = conf_thresh / (1.0 / (1.0 + np.exp(-4.0)))
    cut = float(np.log(need / (1.0 - need)))
    rec = 5 + m.NUM_CLASSES
    # NEW_COORDS heads carry their logistic activation inside the network: the head tensor holds sigmoid(logit), which
    # saturates -- the bias is then found in a few passes (quantiles commute with the monotonic sigmoid)
    bias = [[-12.0] * n for n in n_anchors]          # (lets nothing through: the candidate list cannot overflow)
    for _ in range(1 if not m.NEW_COORDS else 6):
        det = YOLODetector(size, labels, model=model, conf_thresh=conf_thresh,
                           weights=ScriptedHeadWeights(seed, m.NUM_CLASSES, label, bias, obj_gain))
        try:
            det.detect_async(frame)
            det.postprocess()
            vals = []
            for head in det.heads:
                t = det.backend.read(head, 1)[0]                       # (h, w, anchors * rec)
                vals.append(t.reshape(t.shape[0] * t.shape[1], -1, rec)[..., 4].astype(np.float64))   # [cells, anchors]
        finally:
            det.backend.close()
        total = sum(v.size for v in vals)
        share = min(0.5, target / total)                              # the same share of every (head, anchor)'s cells
        step = 0.0
        for h, v in enumerate(vals):
            for a in range(v.shape[1]):
                q = float(np.quantile(v[:, a], 1.0 - share))
                if m.NEW_COORDS:
                    q = min(max(q, 1e-30), 1.0 - 1e-7)
                    q = float(np.log(q / (1.0 - q)))
                bias[h][a] += cut - q
                step = max(step, abs(cut - q))
        if step < 0.02:
            break
    return ScriptedHeadWeights(seed, m.NUM_CLASSES, label, bias, obj_gain)
