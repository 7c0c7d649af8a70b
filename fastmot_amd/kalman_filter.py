"""KalmanFilter front end (API of fastmot/kalman_filter.py:12-225) over the batched HIP kernels.

The filter state of every track lives in the device-resident track table; the hot path
(MultiTracker.apply_kalman / update) drives the `*_slots` methods, which process all tracks in
one launch.  The reference's ndarray methods (`create`, `warp`, `predict`, `update`,
`motion_distance`) are kept for drop-in use: they run the SAME kernels through a scratch slot.
"""
from enum import Enum

import numpy as np

from . import _lib
from .runtime import get_context, SCRATCH_SLOT


class MeasType(Enum):
    FLOW = 0
    DETECTOR = 1


_EYE3 = np.eye(3)


class KalmanFilter:
    def __init__(self,
                 std_factor_acc=2.25,
                 std_offset_acc=78.5,
                 std_factor_det=(0.08, 0.08),
                 std_factor_klt=(0.14, 0.14),
                 min_std_det=(4.0, 4.0),
                 min_std_klt=(5.0, 5.0),
                 init_pos_weight=5,
                 init_vel_weight=12,
                 vel_coupling=0.6,
                 vel_half_life=2):
        """8-state (x1, y1, x2, y2 + velocities) bounding-box Kalman filter; parameters and
        range checks as fastmot/kalman_filter.py:13-85."""
        assert std_factor_acc >= 0
        assert std_factor_det[0] >= 0 and std_factor_det[1] >= 0
        assert std_factor_klt[0] >= 0 and std_factor_klt[1] >= 0
        assert min_std_det[0] >= 0 and min_std_det[1] >= 0
        assert min_std_klt[0] >= 0 and min_std_klt[1] >= 0
        assert init_pos_weight >= 0
        assert init_vel_weight >= 0
        assert 0 <= vel_coupling <= 1
        assert vel_half_life > 0
        self.std_factor_acc = std_factor_acc
        self.std_offset_acc = std_offset_acc
        self.std_factor_det = std_factor_det
        self.std_factor_klt = std_factor_klt
        self.min_std_det = min_std_det
        self.min_std_klt = min_std_klt
        self.init_pos_weight = init_pos_weight
        self.init_vel_weight = init_vel_weight
        self.vel_coupling = vel_coupling
        self.vel_half_life = vel_half_life
        self.ctx = get_context()
        self.reset_dt(1 / 30.)

    def reset_dt(self, dt):
        """Rebuilds transition / process-noise constants for a new frame interval
        (fastmot/kalman_filter.py:87-94,294-306) and pushes them to the device."""
        self.dt = dt
        self.ctx.kf_configure(dt, self.std_factor_acc, self.std_offset_acc, self.std_factor_det,
                              self.std_factor_klt, self.min_std_det, self.min_std_klt,
                              self.init_pos_weight, self.init_vel_weight, self.vel_coupling,
                              self.vel_half_life)

    # ------------------------------------------------------------------ batched (hot path)
    def create_slots(self, slots, det_tlbrs):
        self.ctx.trk_create(slots, det_tlbrs)

    def step_slots(self, slots, homography, klt_tlbrs, has_klt, multipliers):
        """warp -> predict -> optional KLT update for all `slots`; returns rounded boxes and the
        out-of-frame mask (tracker.py:164-183)."""
        return self.ctx.trk_step(slots, homography, klt_tlbrs, has_klt, multipliers)

    def update_det_slots(self, slots, det_tlbrs):
        return self.ctx.trk_update_det(slots, det_tlbrs)

    # ------------------------------------------------------------------ ndarray API (drop-in)
    def _run(self, ops, mean, covariance, H=_EYE3, meas=None, mult=1.):
        ctx = self.ctx
        ctx.trk_set_state([SCRATCH_SLOT], mean, covariance)
        klt = np.zeros((1, 4)) if meas is None else np.asarray(meas, np.float64).reshape(1, 4)
        ctx.trk_step_ops(ops, [SCRATCH_SLOT], H, klt, [meas is not None], [mult])
        m, c = ctx.trk_get_state([SCRATCH_SLOT])
        return m[0], c[0]

    def create(self, det_meas):
        self.ctx.trk_create([SCRATCH_SLOT], np.asarray(det_meas, np.float64).reshape(1, 4))
        m, c = self.ctx.trk_get_state([SCRATCH_SLOT])
        return m[0], c[0]

    def warp(self, mean, covariance, H):
        return self._run(1, mean, covariance, H=H)

    def predict(self, mean, covariance):
        return self._run(2, mean, covariance)

    def update(self, mean, covariance, measurement, meas_type, multiplier=1.):
        if meas_type == MeasType.FLOW:
            return self._run(4, mean, covariance, meas=measurement, mult=multiplier)
        if meas_type == MeasType.DETECTOR:
            if multiplier != 1.:
                raise ValueError('detector measurements use multiplier 1 (tracker.py:261)')
            self.ctx.trk_set_state([SCRATCH_SLOT], mean, covariance)
            self.ctx.trk_update_det([SCRATCH_SLOT], np.asarray(measurement, np.float64).reshape(1, 4))
            m, c = self.ctx.trk_get_state([SCRATCH_SLOT])
            return m[0], c[0]
        raise ValueError('Invalid measurement type')

    def motion_distance(self, mean, covariance, measurements):
        """Squared Mahalanobis distances to N boxes (fastmot/kalman_filter.py:206-225)."""
        meas = np.asarray(measurements, np.float64).reshape(-1, 4)
        n = len(meas)
        if n == 0:
            return np.empty(0)
        ctx = self.ctx
        ctx.trk_set_state([SCRATCH_SLOT], mean, covariance)
        saved = ctx.device_emb_host
        ctx.emb_upload(np.zeros((n, ctx.feat_dim), np.float32))
        ctx.device_emb_host = None if saved is None else None
        ctx.assoc_prepare(_lib.METRIC_EUCLIDEAN, [SCRATCH_SLOT], np.zeros((1, 4)), [0], meas,
                          np.zeros(n, np.int64), np.zeros(n, np.uint8))
        _, maha, _ = ctx.assoc_get_pairwise(1, n)
        return maha[0]
