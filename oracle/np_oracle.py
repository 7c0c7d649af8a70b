"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy, fp64) of the reference's tracker math.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product package (fastmot_amd) never does.

Every function restates one reference routine (file:line cited, paths relative
to /root/reference).  The restatement is batched/vectorised where the reference
loops per track, so floating-point results agree with the reference to
round-off (tests pin 1e-9 relative), while every integer / index / boolean
result must be identical.  Parity is pinned by tests/test_oracle_vs_reference.py
(runs where /root/reference exists) and by the committed golden vectors in
tests/golden/ produced by oracle/make_golden.py from the reference itself.

Third-party arithmetic the reference delegates to:
  * scipy.optimize.linear_sum_assignment (requirements.txt: scipy>=1.5; 1.15.3
    here) -- used directly as the oracle for the LAP kernel (lsa()).
"""
import numpy as np
from scipy.optimize import linear_sum_assignment

import numba_set

INF_COST = 1e5            # utils/matching.py:7
CHI_SQ_INV_95 = 9.4877    # utils/matching.py:6


# ----------------------------------------------------------------------------
# box algebra -- utils/rect.py
# ----------------------------------------------------------------------------
def round_half_even(x):
    """round(float, 0) of CPython == np.rint (utils/rect.py:6-13, 49-57)."""
    return np.rint(np.asarray(x, dtype=np.float64))


def to_tlbr(tlwh):
    """utils/rect.py:49-57 (batched)."""
    tlwh = np.asarray(tlwh, dtype=np.float64).reshape(-1, 4)
    out = np.empty_like(tlwh)
    out[:, 0] = np.rint(tlwh[:, 0])
    out[:, 1] = np.rint(tlwh[:, 1])
    out[:, 2] = np.rint(tlwh[:, 0] + tlwh[:, 2] - 1.)
    out[:, 3] = np.rint(tlwh[:, 1] + tlwh[:, 3] - 1.)
    return out


def box_area(tlbr):
    """utils/rect.py:28-32 (batched): 0 when w<=0 or h<=0."""
    tlbr = np.asarray(tlbr, dtype=np.float64).reshape(-1, 4)
    w = tlbr[:, 2] - tlbr[:, 0] + 1
    h = tlbr[:, 3] - tlbr[:, 1] + 1
    return np.where((w <= 0) | (h <= 0), 0., w * h)


def ios(tlbr, other):
    """utils/rect.py:101-109 (batched over `tlbr`, single `other`)."""
    tlbr = np.asarray(tlbr, dtype=np.float64).reshape(-1, 4)
    iw = np.minimum(tlbr[:, 2], other[2]) - np.maximum(tlbr[:, 0], other[0]) + 1
    ih = np.minimum(tlbr[:, 3], other[3]) - np.maximum(tlbr[:, 1], other[1]) + 1
    valid = (iw > 0) & (ih > 0)
    with np.errstate(divide='ignore', invalid='ignore'):
        val = iw * ih / box_area(tlbr)
    return np.where(valid, val, 0.)


def _pair_inter(a, b):
    iw = np.minimum(a[:, None, 2], b[None, :, 2]) - np.maximum(a[:, None, 0], b[None, :, 0]) + 1
    ih = np.minimum(a[:, None, 3], b[None, :, 3]) - np.maximum(a[:, None, 1], b[None, :, 1]) + 1
    return iw, ih


def iou_dist(a, b):
    """utils/distance.py:91-108."""
    a = np.asarray(a, dtype=np.float64).reshape(-1, 4)
    b = np.asarray(b, dtype=np.float64).reshape(-1, 4)
    iw, ih = _pair_inter(a, b)
    valid = (iw > 0) & (ih > 0)
    inter = iw * ih
    union = box_area(a)[:, None] + box_area(b)[None, :] - inter
    with np.errstate(divide='ignore', invalid='ignore'):
        d = 1. - inter / union
    return np.where(valid, d, 1.)


def find_occluded(tlbrs, thresh):
    """utils/rect.py:143-157."""
    t = np.asarray(tlbrs, dtype=np.float64).reshape(-1, 4)
    n = len(t)
    if n == 0:
        return np.zeros(0, np.bool_)
    iw, ih = _pair_inter(t, t)
    valid = (iw > 0) & (ih > 0) & ~np.eye(n, dtype=bool)
    with np.errstate(divide='ignore', invalid='ignore'):
        frac = iw * ih / box_area(t)[:, None]
    return np.any(valid & (frac >= thresh), axis=1)


# ----------------------------------------------------------------------------
# feature distances -- utils/distance.py:17-87
# ----------------------------------------------------------------------------
def cdist(XA, XB, metric, empty_mask=None, fill_val=None):
    """metric in {'euclidean','cosine'}.  utils/distance.py:47-87.

    Mixed precision exactly as the reference's Numba typing gives it: accumulators are float64
    (`norm = 0.`), each product/difference has the NumPy result type of its operands.  In
    _matching_cost XA is float64 (f32 averages copied into an f64 array, tracker.py:321-326) and XB
    float32, so dot and a_norm terms are f64 while b_norm terms are f32-rounded squares; in
    _reid_cost both are float32 (tracker.py:360-362), so every term is an f32 product.
    (The no-op-numba shim under NumPy 2 additionally ACCUMULATES f32 terms in f32 -- NEP 50 weak
    python scalars -- which Numba does not; goldens from the shim agree to ~1e-7 only.)"""
    XA = np.asarray(XA)
    XB = np.asarray(XB)
    filler = 1. if fill_val is None else fill_val
    ta = XA.dtype if XA.dtype in (np.float32, np.float64) else np.float64
    tb = XB.dtype if XB.dtype in (np.float32, np.float64) else np.float64
    tab = np.result_type(ta, tb)
    A = XA.astype(ta)[:, None, :]
    B = XB.astype(tb)[None, :, :]
    if metric == 'euclidean':
        diff = (A.astype(tab) - B.astype(tab))
        Y = np.sqrt(np.sum((diff * diff).astype(np.float64), axis=2))
    elif metric == 'cosine':
        dot = np.sum((A.astype(tab) * B.astype(tab)).astype(np.float64), axis=2)
        na = np.sqrt(np.sum((XA.astype(ta) * XA.astype(ta)).astype(np.float64), axis=1))
        nb_ = np.sqrt(np.sum((XB.astype(tb) * XB.astype(tb)).astype(np.float64), axis=1))
        Y = 1. - dot / (na[:, None] * nb_[None, :])
    else:
        raise ValueError('Unsupported distance metric')
    if empty_mask is not None:
        Y = np.where(empty_mask, filler, Y)
    return Y


# ----------------------------------------------------------------------------
# Kalman filter -- kalman_filter.py
# ----------------------------------------------------------------------------
class KFParams:
    """Tunables of KalmanFilter.__init__ (kalman_filter.py:13-85) + matrices
    of _init_mat (kalman_filter.py:294-306)."""

    def __init__(self, dt=1 / 30., std_factor_acc=2.25, std_offset_acc=78.5,
                 std_factor_det=(0.08, 0.08), std_factor_klt=(0.14, 0.14),
                 min_std_det=(4.0, 4.0), min_std_klt=(5.0, 5.0),
                 init_pos_weight=5, init_vel_weight=12, vel_coupling=0.6, vel_half_life=2):
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
        self.reset_dt(dt)

    def reset_dt(self, dt):
        self.dt = dt
        eye4 = np.eye(4)
        self.acc_cov = np.block([[eye4 * (0.25 * dt**4), eye4 * (0.5 * dt**3)],
                                 [eye4 * (0.5 * dt**3), eye4 * dt**2]])
        F = np.eye(8)
        for i in range(4):
            F[i, i + 4] = self.vel_coupling * dt
            F[i, (i + 2) % 4 + 4] = (1. - self.vel_coupling) * dt
            F[i + 4, i + 4] = 0.5**(dt / self.vel_half_life)
        self.trans_mat = F


def kf_create(p, boxes):
    """kalman_filter.py:96-126, batched: boxes (N,4) -> mean (N,8), cov (N,8,8)."""
    boxes = np.asarray(boxes, dtype=np.float64).reshape(-1, 4)
    n = len(boxes)
    mean = np.concatenate([boxes, np.zeros_like(boxes)], axis=1)
    w = boxes[:, 2] - boxes[:, 0] + 1
    h = boxes[:, 3] - boxes[:, 1] + 1
    sp_w = np.maximum(p.init_pos_weight * p.std_factor_det[0] * w, p.min_std_det[0])
    sp_h = np.maximum(p.init_pos_weight * p.std_factor_det[1] * h, p.min_std_det[1])
    sv_w = np.maximum(p.init_vel_weight * p.std_factor_det[0] * w, p.min_std_det[0])
    sv_h = np.maximum(p.init_vel_weight * p.std_factor_det[1] * h, p.min_std_det[1])
    std = np.stack([sp_w, sp_h, sp_w, sp_h, sv_w, sv_h, sv_w, sv_h], axis=1)
    cov = np.zeros((n, 8, 8))
    idx = np.arange(8)
    cov[:, idx, idx] = std**2
    return mean, cov


def kf_warp(mean, cov, H):
    """kalman_filter.py:227-292 restated per corner (Appendix B of SURVEY.md):
    for corner position p with velocity v,  a = h3.p + 1, b = h3.v, q = H1 p + h2,
    p' = q/a,  v' = H1 v/a - b q/a^2, covariance propagated with the analytic
    Jacobian (block form of F_tl + F_br)."""
    mean = np.asarray(mean, dtype=np.float64).reshape(-1, 8)
    cov = np.asarray(cov, dtype=np.float64).reshape(-1, 8, 8)
    n = len(mean)
    H1, h2, h3 = H[:2, :2], H[:2, 2], H[2, :2]
    out = np.empty_like(mean)
    J = np.zeros((n, 8, 8))
    for c in (0, 2):               # corner offset inside the position block
        p = mean[:, c:c + 2]
        v = mean[:, c + 4:c + 6]
        a = p @ h3 + 1.
        b = v @ h3
        q = p @ H1.T + h2
        Hv = v @ H1.T
        out[:, c:c + 2] = q / a[:, None]
        out[:, c + 4:c + 6] = Hv / a[:, None] - b[:, None] * q / (a**2)[:, None]
        q_h3 = q[:, :, None] * h3[None, None, :]
        Hv_h3 = Hv[:, :, None] * h3[None, None, :]
        dpp = H1[None] / a[:, None, None] - q_h3 / (a**2)[:, None, None]
        dvp = (-(Hv_h3 + b[:, None, None] * H1[None]) / (a**2)[:, None, None]
               + 2 * b[:, None, None] * q_h3 / (a**3)[:, None, None])
        J[:, c:c + 2, c:c + 2] = dpp
        J[:, c + 4:c + 6, c + 4:c + 6] = dpp
        J[:, c + 4:c + 6, c:c + 2] = dvp
    return out, J @ cov @ J.transpose(0, 2, 1)


def kf_predict(p, mean, cov):
    """kalman_filter.py:308-319 batched."""
    mean = np.asarray(mean, dtype=np.float64).reshape(-1, 8)
    cov = np.asarray(cov, dtype=np.float64).reshape(-1, 8, 8)
    w = mean[:, 2] - mean[:, 0] + 1
    h = mean[:, 3] - mean[:, 1] + 1
    std = p.std_factor_acc * np.maximum(w, h) + p.std_offset_acc
    motion_cov = p.acc_cov[None] * (std**2)[:, None, None]
    F = p.trans_mat
    mean = mean @ F.T
    cov = F[None] @ cov @ F.T[None] + motion_cov
    cov = 0.5 * (cov + cov.transpose(0, 2, 1))
    return mean, cov


def kf_project(p, mean, cov, meas_type, multiplier=1.):
    """kalman_filter.py:141-171,321-336 batched.  meas_type: 'flow'|'detector'.
    multiplier may be per-track."""
    mean = np.asarray(mean, dtype=np.float64).reshape(-1, 8)
    cov = np.asarray(cov, dtype=np.float64).reshape(-1, 8, 8)
    if meas_type == 'flow':
        fac, mn = p.std_factor_klt, p.min_std_klt
    elif meas_type == 'detector':
        fac, mn = p.std_factor_det, p.min_std_det
    else:
        raise ValueError('Invalid measurement type')
    w = mean[:, 2] - mean[:, 0] + 1
    h = mean[:, 3] - mean[:, 1] + 1
    sw = np.maximum(fac[0] * w, mn[0])
    sh = np.maximum(fac[1] * h, mn[1])
    std = np.stack([sw, sh, sw, sh], axis=1) * np.asarray(multiplier, dtype=np.float64).reshape(-1, 1)
    S = cov[:, :4, :4].copy()
    idx = np.arange(4)
    S[:, idx, idx] += std**2
    return mean[:, :4].copy(), S


def kf_update(p, mean, cov, meas, meas_type, multiplier=1.):
    """kalman_filter.py:173-204,338-345 batched."""
    mean = np.asarray(mean, dtype=np.float64).reshape(-1, 8)
    cov = np.asarray(cov, dtype=np.float64).reshape(-1, 8, 8)
    meas = np.asarray(meas, dtype=np.float64).reshape(-1, 4)
    pm, S = kf_project(p, mean, cov, meas_type, multiplier)
    PHt = cov[:, :, :4]                                   # (N,8,4)
    K = np.linalg.solve(S, PHt.transpose(0, 2, 1)).transpose(0, 2, 1)   # (N,8,4)
    innov = meas - pm
    mean = mean + np.einsum('nj,nij->ni', innov, K)
    cov = cov - K @ S @ K.transpose(0, 2, 1)
    return mean, cov


def kf_maha(p, mean, cov, meas):
    """kalman_filter.py:206-225,347-353: (N tracks) x (D measurements) squared
    Mahalanobis distances in box space."""
    mean = np.asarray(mean, dtype=np.float64).reshape(-1, 8)
    meas = np.asarray(meas, dtype=np.float64).reshape(-1, 4)
    pm, S = kf_project(p, mean, cov, 'detector')
    out = np.empty((len(mean), len(meas)))
    for i in range(len(mean)):
        L = np.linalg.cholesky(S[i])
        y = np.linalg.solve(L, (meas - pm[i]).T)
        out[i] = np.sum(y**2, axis=0)
    return out


# ----------------------------------------------------------------------------
# association cost -- tracker.py:314-366, utils/matching.py:101-116
# ----------------------------------------------------------------------------
def matching_cost(feat_dist, maha, t_labels, d_labels, motion_weight, max_cost):
    """fuse_motion + gate_cost applied to a feature-distance matrix."""
    cost = (1. - motion_weight) * feat_dist + motion_weight * (1. / CHI_SQ_INV_95) * maha
    cost = np.where(maha > CHI_SQ_INV_95, INF_COST, cost)
    gate = (np.asarray(t_labels)[:, None] != np.asarray(d_labels)[None, :]) | (cost > max_cost)
    return np.where(gate, INF_COST, cost)


def gate_cost(cost, t_labels, d_labels, max_cost=None):
    """utils/matching.py:109-116."""
    gate = np.asarray(t_labels)[:, None] != np.asarray(d_labels)[None, :]
    if max_cost is not None:
        gate = gate | (cost > max_cost)
    return np.where(gate, INF_COST, cost)


def lsa(cost):
    """scipy.optimize.linear_sum_assignment -- the algorithm the reference calls
    at utils/matching.py:27."""
    if cost.shape[0] == 0 or cost.shape[1] == 0:
        return np.empty(0, np.int64), np.empty(0, np.int64)
    r, c = linear_sum_assignment(cost)
    return r.astype(np.int64), c.astype(np.int64)


def assignment_matches(cost, m_rows, m_cols):
    """utils/matching.py:58-70 on local indices: matches, unmatched rows, unmatched cols.
    The unmatched lists are `list(set(range(n)) - set(matched))` evaluated INSIDE @njit: the iteration order of
    Numba's hash set (numba_set.py), which is neither ascending nor CPython's in general -- reproduced because it
    decides the order in which new track IDs are handed out."""
    nr, nc = cost.shape
    u_rows = numba_set.difference_order(nr, [int(r) for r in m_rows])
    u_cols = numba_set.difference_order(nc, [int(c) for c in m_cols])
    matches = []
    for r, c in zip(m_rows, m_cols):
        if cost[r, c] < INF_COST:
            matches.append((int(r), int(c)))
        else:
            u_rows.append(int(r))
            u_cols.append(int(c))
    return matches, u_rows, u_cols


def greedy_match(cost, max_cost):
    """utils/matching.py:74-97 on local indices (first-minimum argmin order)."""
    cost = np.array(cost, dtype=np.float64)
    rows = list(range(cost.shape[0]))
    cols = list(range(cost.shape[1]))
    matches = []
    while cost.shape[0] > 0 and cost.shape[1] > 0:
        idx = int(np.argmin(cost))
        i, j = divmod(idx, cost.shape[1])
        if cost[i, j] <= max_cost:
            matches.append((rows[i], cols[j]))
            del rows[i]
            del cols[j]
            cost = np.delete(np.delete(cost, i, axis=0), j, axis=1)
        else:
            break
    return matches, rows, cols


# ----------------------------------------------------------------------------
# running-mean embedding -- track.py:119-126 (fp32, in place)
# ----------------------------------------------------------------------------
def average_feature(fsum, vec, count):
    """returns (new_sum, new_avg), all float32 like the reference arrays."""
    fsum = (fsum + vec).astype(np.float32)
    avg = (fsum * np.float32(1. / count)).astype(np.float32)
    norm = np.float32(1. / np.linalg.norm(avg))
    return fsum, (avg * norm).astype(np.float32)


# ----------------------------------------------------------------------------
# detector pre/post-processing -- detector.py:289-365, plugins/yolo_layer.cu:127-230
# ----------------------------------------------------------------------------
def yolo_preprocess(frame, in_hw, roi=None):
    """YOLODetector._preprocess (detector.py:289-300) + _create_letterbox (:302-320).
    cupyx.scipy.ndimage.zoom(order=1, mode='opencv', grid_mode=True) on uint8 is (CuPy 9.x
    cupyx/scipy/ndimage/interpolation.py): src = dst*z + (z-1)/2 with z = in/out, linear
    interpolation in float64, edge clamp ('nearest'), result cast to uint8 by rint.  Then BGR->RGB,
    HWC->CHW, * 1/255 into float32.  Returns float32 [3, in_h, in_w]; outside `roi`
    (x, y, w, h) the letterbox value is 0.5."""
    in_h, in_w = in_hw
    rx, ry, rw, rh = (0, 0, in_w, in_h) if roi is None else roi
    out = np.full((3, in_h, in_w), 0.5, np.float32)
    val = np.clip(np.rint(zoom_linear_opencv(frame, rh, rw)), 0, 255)
    rgb = val[..., ::-1].transpose(2, 0, 1)
    out[:, ry:ry + rh, rx:rx + rw] = (rgb * (1 / 255.)).astype(np.float32)
    return out


def zoom_linear_opencv(frame, rh, rw):
    """The interpolated values of cupyx.scipy.ndimage.zoom(frame, ..., order=1, mode='opencv') before the cast to the
    integer output type, float64 [rh, rw, C]: CuPy turns mode='opencv' into affine_transform(zoom = in / out,
    offset = (zoom - 1) / 2, mode='nearest').  tests/test_preprocess_scipy_pin.py holds this against SciPy's
    affine_transform with the same arguments (the CPU sibling of the CuPy routine): identical but for exact .5 ties,
    which SciPy's integer cast rounds up and CuPy's rint() rounds to even."""
    fh, fw = frame.shape[:2]
    zy, zx = fh / rh, fw / rw
    sy = np.arange(rh) * zy + (zy - 1.) / 2.
    sx = np.arange(rw) * zx + (zx - 1.) / 2.
    y0 = np.floor(sy); wy = sy - y0
    x0 = np.floor(sx); wx = sx - x0
    y0i = np.clip(y0.astype(int), 0, fh - 1); y1i = np.clip(y0.astype(int) + 1, 0, fh - 1)
    x0i = np.clip(x0.astype(int), 0, fw - 1); x1i = np.clip(x0.astype(int) + 1, 0, fw - 1)
    f = frame.astype(np.float64)
    top = (1. - wx)[None, :, None] * f[y0i][:, x0i] + wx[None, :, None] * f[y0i][:, x1i]
    bot = (1. - wx)[None, :, None] * f[y1i][:, x0i] + wx[None, :, None] * f[y1i][:, x1i]
    return (1. - wy)[:, None, None] * top + wy[:, None, None] * bot


def yolo_decode(head, anchors, num_classes, in_wh, scale_xy, new_coords=False):
    """CalDetection / CalDetection_NewCoords (plugins/yolo_layer.cu:127-230) for one head.
    head: float32 [(5+C)*A, H, W] (the plugin's NCHW input).  Returns [A*H*W, 7] float32 rows
    (x, y, w, h, box_conf, class_id, class_prob), (anchor, cell) order, fp32 arithmetic."""
    f32 = np.float32
    info = 5 + num_classes
    A = len(anchors) // 2
    _, H, W = head.shape
    h = head.reshape(A, info, H, W).astype(f32)
    cls_logit = h[:, 5:]
    cls = np.argmax(cls_logit, axis=1)               # first maximum wins (strict '>')
    best = np.max(cls_logit, axis=1)
    col = np.arange(W, dtype=f32)[None, None, :]
    row = np.arange(H, dtype=f32)[None, :, None]
    s = f32(scale_xy)
    sig = lambda v: (f32(1) / (f32(1) + np.exp(-v, dtype=f32))).astype(f32)
    aw = np.asarray(anchors[0::2], f32)[:, None, None]
    ah = np.asarray(anchors[1::2], f32)[:, None, None]
    if not new_coords:
        cls_prob, box_prob = sig(best), sig(h[:, 4])
        bx = (col + (s * sig(h[:, 0]) - (s - f32(1)) * f32(0.5))) / f32(W)
        by = (row + (s * sig(h[:, 1]) - (s - f32(1)) * f32(0.5))) / f32(H)
        bw = np.exp(h[:, 2], dtype=f32) * aw / f32(in_wh[0])
        bh = np.exp(h[:, 3], dtype=f32) * ah / f32(in_wh[1])
    else:
        cls_prob, box_prob = best, h[:, 4]
        bx = (col + (s * h[:, 0] - (s - f32(1)) * f32(0.5))) / f32(W)
        by = (row + (s * h[:, 1] - (s - f32(1)) * f32(0.5))) / f32(H)
        bw = h[:, 2] * h[:, 2] * f32(4) * aw / f32(in_wh[0])
        bh = h[:, 3] * h[:, 3] * f32(4) * ah / f32(in_wh[1])
    bx = bx - bw / f32(2)
    by = by - bh / f32(2)
    rows = np.stack([bx, by, bw, bh, box_prob, cls.astype(f32), cls_prob], axis=-1)
    return rows.reshape(-1, 7).astype(f32)


def diou_nms(tlwhs, scores, thresh, beta=0.6, tie_rank=None):
    """utils/rect.py:199-244 with the operand types Numba assigns (float32 rows; `- 1` and `/ 2`
    promote to float64; areas stay float32).  Deterministic order: descending score, ties by
    ascending index (the reference's argsort is an unstable quicksort: tie order undefined).
    `tie_rank` (optional, one number per row) replaces the index as the tie-breaker: the tests use it to
    ask whether a result depends on the order the reference leaves undefined."""
    t = np.asarray(tlwhs, np.float32)
    n = len(t)
    tr = np.arange(n) if tie_rank is None else np.asarray(tie_rank)
    order = sorted(range(n), key=lambda i: (-float(scores[i]), tr[i]))
    areas = t[:, 2] * t[:, 3]                                    # float32
    tls = t[:, :2]
    brs = (t[:, :2] + t[:, 2:]).astype(np.float64) - 1
    centers = (tls.astype(np.float64) + brs) / 2
    keep = []
    alive = list(order)
    while alive:
        i = alive[0]
        keep.append(i)
        rest = np.array(alive[1:], int)
        if len(rest) == 0:
            break
        ixmin = np.maximum(tls[i, 0], tls[rest, 0]).astype(np.float64)
        iymin = np.maximum(tls[i, 1], tls[rest, 1]).astype(np.float64)
        ixmax = np.minimum(brs[i, 0], brs[rest, 0])
        iymax = np.minimum(brs[i, 1], brs[rest, 1])
        inter = np.maximum(0, ixmax - ixmin + 1) * np.maximum(0, iymax - iymin + 1)
        union = (areas[i] + areas[rest]).astype(np.float64) - inter
        iou = inter / union
        exmin = np.minimum(tls[i, 0], tls[rest, 0]).astype(np.float64)
        eymin = np.minimum(tls[i, 1], tls[rest, 1]).astype(np.float64)
        ew = np.maximum(brs[i, 0], brs[rest, 0]) - exmin + 1
        eh = np.maximum(brs[i, 1], brs[rest, 1]) - eymin + 1
        c = ew**2 + eh**2
        d = np.sum((centers[i] - centers[rest])**2, axis=1)
        diou = iou - (d / c)**beta
        alive = rest[diou <= thresh].tolist()
    return np.array(keep, int)


def filter_scale(det_out, size, offset, label_mask, conf_thresh):
    """First half of YOLODetector._filter_dets (detector.py:329-341): class mask + score threshold, then the rows
    scaled to pixels in place (float32).  Returns (rows float32 [k,7] in candidate order, their indices in det_out)."""
    d = np.asarray(det_out, np.float32)
    cls = d[:, 5].astype(int)
    ok = (cls >= 0) & (cls < len(label_mask))
    keep = np.zeros(len(d), bool)
    keep[ok] = np.asarray(label_mask, bool)[cls[ok]]
    keep &= (d[:, 4] * d[:, 6]) >= conf_thresh                    # float32 product
    idx = np.flatnonzero(keep)
    d = d[idx].copy()
    sz = np.append(np.asarray(size, np.float64), np.asarray(size, np.float64))
    d[:, :4] = (d[:, :4].astype(np.float64) * sz).astype(np.float32)
    d[:, :2] = (d[:, :2].astype(np.float64) - np.asarray(offset, np.float64)).astype(np.float32)
    return d, idx


def nms_finalize(d, nms_thresh, max_area, min_ar, tie_rank=None):
    """Second half (detector.py:343-364) on scaled candidate rows `d` (float32 [k,7], candidate order): per-class
    DIoU-NMS in ascending class order, to_tlbr, area / aspect filters.  Returns (tlbr f64, label i64, conf f64)."""
    d = np.asarray(d, np.float32)
    tl, lb, cf = [], [], []
    for c in np.unique(d[:, 5]):
        rows = np.flatnonzero(d[:, 5] == c)
        # deterministic tie-break: original candidate index
        k = diou_nms(d[rows, :4], d[rows, 4], nms_thresh,
                     tie_rank=None if tie_rank is None else np.asarray(tie_rank)[rows])
        for r in rows[k]:
            x, y, w, h = (float(v) for v in d[r, :4])
            box = np.array([np.rint(x), np.rint(y), np.rint(x + w - 1.), np.rint(y + h - 1.)])
            bw, bh = box[2] - box[0] + 1, box[3] - box[1] + 1
            area = 0. if bw <= 0 or bh <= 0 else bw * bh
            ar = bh / bw if bw > 0 else 0.
            if 0 < area <= max_area and ar >= min_ar:
                tl.append(box); lb.append(int(d[r, 5])); cf.append(float(d[r, 4] * d[r, 6]))
    return (np.array(tl, np.float64).reshape(-1, 4), np.array(lb, np.int64), np.array(cf, np.float64))


def filter_dets(det_out, size, offset, label_mask, conf_thresh, nms_thresh, max_area, min_ar):
    """YOLODetector._filter_dets (detector.py:322-365).  det_out float32 [n,7].
    Returns (tlbr [m,4] f64, label [m] i64, conf [m] f64) sorted by class, then NMS keep order."""
    d, _ = filter_scale(det_out, size, offset, label_mask, conf_thresh)
    return nms_finalize(d, nms_thresh, max_area, min_ar)
