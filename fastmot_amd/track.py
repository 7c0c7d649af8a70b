"""Track bookkeeping (API of fastmot/track.py:129-225).

A Track is a light host object (id, deques of boxes / frame ids, age, hits, keypoints); its
Kalman state and running-mean ReID feature live in row `slot` of the device track table and
are only copied to the host when `.state` / `.avg_feat()` are read.
"""
from collections import deque

import numpy as np

from .models import get_label_name
from .runtime import get_context


class AverageFeature:
    """Running mean embedding, re-normalised (fastmot/track.py:91-126) -- device resident."""

    def __init__(self, ctx, slot):
        self._ctx = ctx
        self._slot = slot
        self.count = 0

    def __call__(self):
        if self.count == 0:
            return None
        return self._ctx.feat_get(self._slot)[1]

    @property
    def sum(self):
        return None if self.count == 0 else self._ctx.feat_get(self._slot)[0]

    @property
    def avg(self):
        return self.__call__()

    def is_valid(self):
        return self.count > 0

    def update(self, embedding):
        """Single-vector update from a host embedding (off the hot path)."""
        ctx = self._ctx
        ctx.emb_upload(np.asarray(embedding, np.float32).reshape(1, -1))
        ctx.device_emb_host = None
        ctx.feat_update([self._slot], [0])
        self.count += 1

    def merge(self, other):
        self._ctx.feat_merge(self._slot, other._slot)
        self.count += other.count


class Track:
    _count = 0

    def __init__(self, frame_id, tlbr, state, label, confirm_hits=1, buffer_size=30, slot=None):
        self.trk_id = self.next_id()
        self.start_frame = frame_id
        self.frame_ids = deque([frame_id], maxlen=buffer_size)
        self.bboxes = deque([tlbr], maxlen=buffer_size)
        self.confirm_hits = confirm_hits
        self.label = label

        self._ctx = get_context()
        self.slot = self._ctx.slots.alloc() if slot is None else slot
        self._ctx.feat_reset([self.slot])
        if state is not None:
            self.state = state

        self.age = 0
        self.hits = 0
        self.avg_feat = AverageFeature(self._ctx, self.slot)
        self.last_feat = None

        self.inlier_ratio = 1.
        self.keypoints = np.empty((0, 2), np.float32)
        self.prev_keypoints = np.empty((0, 2), np.float32)

    def __str__(self):
        x, y = (self.tlbr[0] + self.tlbr[2]) / 2, (self.tlbr[1] + self.tlbr[3]) / 2
        return f'{get_label_name(self.label):<10} {self.trk_id:>3} at ({int(x):>4}, {int(y):>4})'

    __repr__ = __str__

    def __len__(self):
        return self.end_frame - self.start_frame

    def __lt__(self, other):
        # ordered by approximate distance to the image plane, closer is greater (track.py:160-162)
        return (self.tlbr[-1], -self.age) < (other.tlbr[-1], -other.age)

    @property
    def state(self):
        mean, cov = self._ctx.trk_get_state([self.slot])
        return mean[0], cov[0]

    @state.setter
    def state(self, value):
        mean, cov = value
        self._ctx.trk_set_state([self.slot], mean, cov)

    @property
    def tlbr(self):
        return self.bboxes[-1]

    @property
    def end_frame(self):
        return self.frame_ids[-1]

    @property
    def active(self):
        return self.age < 2

    @property
    def confirmed(self):
        return self.hits >= self.confirm_hits

    def update(self, tlbr, state=None):
        self.bboxes.append(tlbr)
        if state is not None:
            self.state = state

    def add_detection(self, frame_id, tlbr, state, embedding, is_valid=True, on_device=False):
        """`on_device=True`: the caller batches the feature update on the GPU (MultiTracker)."""
        self.frame_ids.append(frame_id)
        self.bboxes.append(tlbr)
        if state is not None:
            self.state = state
        if is_valid:
            self.last_feat = embedding
            if on_device:
                self.avg_feat.count += 1
            else:
                self.avg_feat.update(embedding)
        self.age = 0
        self.hits += 1

    def reinstate(self, frame_id, tlbr, state, embedding, on_device=False):
        self.start_frame = frame_id
        self.frame_ids.append(frame_id)
        self.bboxes.append(tlbr)
        if state is not None:
            self.state = state
        self.last_feat = embedding
        if on_device:
            self.avg_feat.count += 1
        else:
            self.avg_feat.update(embedding)
        self.age = 0
        self.keypoints = np.empty((0, 2), np.float32)
        self.prev_keypoints = np.empty((0, 2), np.float32)

    def mark_missed(self):
        self.age += 1

    def merge_continuation(self, other):
        self.frame_ids.extend(other.frame_ids)
        self.bboxes.extend(other.bboxes)
        self._ctx.trk_copy_state(self.slot, other.slot)
        self.age = other.age
        self.hits += other.hits

        self.keypoints = other.keypoints
        self.prev_keypoints = other.prev_keypoints

        if other.last_feat is not None:
            self.last_feat = other.last_feat
        self.avg_feat.merge(other.avg_feat)

    def release(self):
        """Returns the device slot to the allocator (track deleted / evicted from history)."""
        if self.slot is not None:
            self._ctx.slots.free(self.slot)
            self.slot = None

    @staticmethod
    def next_id():
        Track._count += 1
        return Track._count
