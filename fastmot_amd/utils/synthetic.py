"""Seeded synthetic workload (BASELINE.md section 4 / SURVEY.md section 8d): 1080p BGR frames with a
textured background under a slow camera drift and K textured rectangles moving with constant
velocity, plus the matching detections.  Used by bench.py and the end-to-end tests; no datasets or
weights are available offline, so detections are injected at the Detector.postprocess() boundary
while the detector network, decode and NMS still execute on the GPU."""
import numpy as np

from ..detector import YOLODetector, DET_DTYPE


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


class InjectedYOLODetector(YOLODetector):
    """YOLODetector whose postprocess() waits for the real GPU pipeline (network + decode + NMS
    on the seeded-random weights) and then returns the scripted detections of the synthetic video."""

    def bind_video(self, video, label=1, labels=None):
        self._video, self._label = video, label
        self._labels = labels if labels is not None and len(labels) > 1 else None
        if labels is not None and len(labels) == 1:
            self._label = labels[0]
        self._frame_idx = 0
        self.last_real_count = 0
        self.net_ms = []          # HIP-event time of the detector's layer sequence, one entry per postprocess()

    def detect_async(self, frame):
        super().detect_async(frame)

    def postprocess(self):
        real = super().postprocess()
        self.last_real_count = len(real)
        self.net_ms.append(self.ctx.detect_net_ms())      # the events of THIS frame's network are complete here
        dets = self._video.detections(self._frame_idx, self._label, self._labels)
        self._frame_idx += 1
        return dets
