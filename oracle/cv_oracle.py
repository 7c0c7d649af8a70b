"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the OpenCV routines the reference calls.

OpenCV (pinned 4.1.1 in the reference's Dockerfile:5, >=3.3 in README.md:50) is NOT under
/root/reference and cannot be installed here, so these functions restate the published algorithms
of the OpenCV sources and are themselves **parity unpinned**: no golden vector from a real cv2
exists.  Call sites restated:
  cv2.resize (INTER_LINEAR, 8-bit)     feature_extractor.py:85, flow.py:129-130,154,187
  cv2.resize (INTER_NEAREST)           flow.py:188-189
  cv2.cvtColor(BGR2GRAY)               flow.py:129,153
"""
import numpy as np


def _lin_coef(dsize, ssize):
    """imgproc/resize.cpp (resizeGeneric / HResizeLinear tables): fx = (dx+0.5)*scale-0.5,
    sx = floor(fx), clamp with zero weight at the borders, 11-bit coefficients (cvRound)."""
    scale = ssize / dsize
    fx = ((np.arange(dsize) + 0.5) * scale - 0.5).astype(np.float32)
    sx = np.floor(fx).astype(np.int64)
    fx = fx - sx.astype(np.float32)
    lo = sx < 0
    fx[lo] = 0; sx[lo] = 0
    hi = sx >= ssize - 1
    fx[hi] = 0; sx[hi] = ssize - 1
    a0 = np.rint((np.float32(1) - fx) * np.float32(2048)).astype(np.int64)
    a1 = np.rint(fx * np.float32(2048)).astype(np.int64)
    return sx, np.minimum(sx + 1, ssize - 1), a0, a1


def resize_linear_u8(img, dsize):
    """cv2.resize(img, dsize=(w, h)) for uint8, INTER_LINEAR, any channel count.
    Exact 2x2 decimation is routed to INTER_AREA by cv::resize (resize.cpp: `if (interpolation ==
    INTER_LINEAR && is_area_fast && iscale_x == 2 && iscale_y == 2) interpolation = INTER_AREA`)."""
    dw, dh = dsize
    sh, sw = img.shape[:2]
    src = img.reshape(sh, sw, -1).astype(np.int64)
    if sw == 2 * dw and sh == 2 * dh:
        out = (src[0::2, 0::2] + src[0::2, 1::2] + src[1::2, 0::2] + src[1::2, 1::2] + 2) >> 2
        return out.astype(np.uint8).reshape((dh, dw) + img.shape[2:])
    x0, x1, ax0, ax1 = _lin_coef(dw, sw)
    y0, y1, ay0, ay1 = _lin_coef(dh, sh)
    rows = src[:, x0] * ax0[None, :, None] + src[:, x1] * ax1[None, :, None]       # int, scale 2^11
    S0, S1 = rows[y0], rows[y1]
    out = (((ay0[:, None, None] * (S0 >> 4)) >> 16) + ((ay1[:, None, None] * (S1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8).reshape((dh, dw) + img.shape[2:])


def resize_nearest(img, dsize):
    """cv2.resize(..., interpolation=INTER_NEAREST): sx = min(floor(dx * scale), ssize-1)
    (resize.cpp resizeNN; scale = ssize/dsize in double)."""
    dw, dh = dsize
    sh, sw = img.shape[:2]
    xs = np.minimum(np.floor(np.arange(dw) * (sw / dw)).astype(int), sw - 1)
    ys = np.minimum(np.floor(np.arange(dh) * (sh / dh)).astype(int), sh - 1)
    return img[ys][:, xs]


# cv2.cvtColor(BGR2GRAY) fixed-point width.  OpenCV's RGB2Gray<uchar> (imgproc/color_rgb) used the 14-bit constants of
# color.hpp (B2Y 1868, G2Y 9617, R2Y 4899, yuv_shift 14 -- with the comment "can be changed to 15-shift coeffs") up to the
# 4.2 era and switched to 15 bits (3735 / 19235 / 9798) later in 4.x.  The reference pins OpenCV 4.1.1 (Dockerfile:5;
# README says >= 3.3): 14 is the default here, in the kernels (fm_flow_cfg.gray_coeff_bits) and in c_baseline.c.
# Neither source tree is on this disk: PARITY UNPINNED, both widths are implemented and tested.
GRAY_COEFF_BITS = 14
GRAY_COEFFS = {14: (1868, 9617, 4899), 15: (3735, 19235, 9798)}


def bgr2gray(img, bits=None):
    """cv2.cvtColor(COLOR_BGR2GRAY) for uint8: (B cb + G cg + R cr + (1 << (bits - 1))) >> bits."""
    bits = GRAY_COEFF_BITS if bits is None else bits
    cb, cg, cr = GRAY_COEFFS[bits]
    b = img[..., 0].astype(np.int64)
    g = img[..., 1].astype(np.int64)
    r = img[..., 2].astype(np.int64)
    return ((b * cb + g * cg + r * cr + (1 << (bits - 1))) >> bits).astype(np.uint8)


def reid_preprocess(frame, tlbrs, in_wh=(128, 256)):
    """FeatureExtractor._preprocess/_normalize (feature_extractor.py:84-98) + multi_crop
    (utils/rect.py:93-97): returns float32 [n, 3, h, w] (RGB, ImageNet normalised)."""
    t = np.maximum(np.asarray(tlbrs).astype(np.int_), 0)
    out = np.empty((len(t), 3, in_wh[1], in_wh[0]), np.float32)
    mean = np.array([0.485, 0.456, 0.406])
    std = np.array([0.229, 0.224, 0.225])
    for i, (x1, y1, x2, y2) in enumerate(t):
        crop = frame[y1:y2 + 1, x1:x2 + 1]
        img = resize_linear_u8(crop, in_wh)
        rgb = img[..., ::-1].transpose(2, 0, 1)
        out[i] = ((rgb / 255. - mean[:, None, None]) / std[:, None, None]).astype(np.float32)
    return out


# ----------------------------------------------------------------------------
# Pyramids / Scharr / pyramidal Lucas-Kanade  (video/lkpyramid.cpp, imgproc/pyramids.cpp)
# ----------------------------------------------------------------------------
def pyr_down(img):
    """cv::pyrDown 8UC1: [1 4 6 4 1]/16 separable, BORDER_REFLECT_101, (sum + 128) >> 8."""
    h, w = img.shape
    dw, dh = (w + 1) // 2, (h + 1) // 2
    p = np.pad(img.astype(np.int64), 2, mode='reflect') if min(h, w) > 2 else None
    if p is None:
        raise ValueError('image too small')
    k = np.array([1, 4, 6, 4, 1], np.int64)
    rows = sum(k[i] * p[:, i:i + w] for i in range(5))               # horizontal
    full = sum(k[j] * rows[j:j + h] for j in range(5))               # vertical
    return ((full[0:2 * dh:2, 0:2 * dw:2] + 128) >> 8).astype(np.uint8)


def build_pyramid(img, win, max_level):
    """buildOpticalFlowPyramid: stops when a level would not be larger than the window."""
    pyr = [img]
    for _ in range(max_level):
        h, w = pyr[-1].shape
        nw, nh = (w + 1) // 2, (h + 1) // 2
        if nw <= win or nh <= win:
            break
        pyr.append(pyr_down(pyr[-1]))
    return pyr


def scharr_deriv(img):
    """calcSharrDeriv: int16 (dx, dy) with rows/cols mirrored around the border pixel."""
    p = np.pad(img.astype(np.int64), 1, mode='reflect')
    t0 = (p[:-2] + p[2:]) * 3 + p[1:-1] * 10          # vertical smoothing   [h, w+2]
    t1 = p[2:] - p[:-2]                                # vertical difference
    dx = t0[:, 2:] - t0[:, :-2]
    dy = (t1[:, 2:] + t1[:, :-2]) * 3 + t1[:, 1:-1] * 10
    return dx.astype(np.int16), dy.astype(np.int16)


def _reflect(i, n):
    i = np.asarray(i)
    i = np.where(i < 0, -i, i)
    i = np.where(i >= n, 2 * n - 2 - i, i)
    i = np.where(i < 0, -i, i)
    return i


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def calc_optical_flow_pyr_lk(prev_img, next_img, prev_pts, win=5, max_level=5, max_count=10, epsilon=0.03,
                             min_eig_threshold=1e-4):
    """cv2.calcOpticalFlowPyrLK(prev, next, pts, None, winSize=(win,win), maxLevel, criteria
    (COUNT|EPS, max_count, epsilon)), flags = 0.  float32 arithmetic as LKTrackerInvoker's scalar
    path.  Returns next_pts [n,2] f32, status [n] u8, err [n] f32."""
    f32 = np.float32
    pts = np.asarray(prev_pts, f32).reshape(-1, 2)
    n = len(pts)
    I_pyr = build_pyramid(prev_img, win, max_level)
    J_pyr = build_pyramid(next_img, win, max_level)
    levels = len(I_pyr)
    max_count = min(max(max_count, 0), 100)
    eps2 = f32(min(max(epsilon, 0.), 10.) ** 2)
    half = f32((win - 1) * 0.5)
    nxt = np.zeros((n, 2), f32)
    status = np.ones(n, bool)
    err = np.zeros(n, f32)
    FLT_SCALE = f32(1. / (1 << 20))
    wy, wx = np.meshgrid(np.arange(win), np.arange(win), indexing='ij')

    def weights(fx, fy):
        iw00 = np.rint((f32(1) - fx) * (f32(1) - fy) * f32(1 << 14)).astype(np.int64)
        iw01 = np.rint(fx * (f32(1) - fy) * f32(1 << 14)).astype(np.int64)
        iw10 = np.rint((f32(1) - fx) * fy * f32(1 << 14)).astype(np.int64)
        return iw00, iw01, iw10, (1 << 14) - iw00 - iw01 - iw10

    def sample(img, ix, iy, w4, shift, zero_outside=False):
        h, w = img.shape
        out = 0
        for (dy, dx), wt in zip(((0, 0), (0, 1), (1, 0), (1, 1)), w4):
            yy = iy[:, None, None] + wy[None] + dy
            xx = ix[:, None, None] + wx[None] + dx
            if zero_outside:
                ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
                v = np.where(ok, img[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)].astype(np.int64), 0)
            else:
                v = img[_reflect(yy, h), _reflect(xx, w)].astype(np.int64)
            out = out + v * wt[:, None, None]
        return _descale(out, shift)

    for level in range(levels - 1, -1, -1):
        I, J = I_pyr[level], J_pyr[level]
        h, w = I.shape
        dx_img, dy_img = scharr_deriv(I)
        sc = f32(1. / (1 << level))
        pp = pts * sc
        if level == levels - 1:
            nxt = pp.copy()
        else:
            nxt = nxt * f32(2)
        pp = pp - half
        ip = np.floor(pp).astype(np.int64)
        oob = (ip[:, 0] < -win) | (ip[:, 0] >= w) | (ip[:, 1] < -win) | (ip[:, 1] >= h)
        if level == 0:
            status &= ~oob
            err[oob] = 0
        act = ~oob
        fa = (pp[:, 0] - ip[:, 0].astype(f32)).astype(f32)
        fb = (pp[:, 1] - ip[:, 1].astype(f32)).astype(f32)
        w4 = weights(fa, fb)
        Ip = sample(I, ip[:, 0], ip[:, 1], w4, 14 - 5)
        dIx = sample(dx_img, ip[:, 0], ip[:, 1], w4, 14, zero_outside=True)
        dIy = sample(dy_img, ip[:, 0], ip[:, 1], w4, 14, zero_outside=True)

        def fsum(a):                       # sequential float32 accumulation in (y, x) order
            acc = np.zeros(len(a), f32)
            flat = a.reshape(len(a), -1)
            for k in range(flat.shape[1]):
                acc = (acc + flat[:, k].astype(f32)).astype(f32)
            return acc
        A11 = fsum(dIx * dIx) * FLT_SCALE
        A12 = fsum(dIx * dIy) * FLT_SCALE
        A22 = fsum(dIy * dIy) * FLT_SCALE
        D = A11 * A22 - A12 * A12
        minEig = (A22 + A11 - np.sqrt((A11 - A22) * (A11 - A22) + f32(4) * A12 * A12)) / f32(2 * win * win)
        bad = act & ((minEig < f32(min_eig_threshold)) | (D < np.finfo(f32).eps))
        if level == 0:
            status &= ~bad
        act = act & ~bad
        with np.errstate(divide='ignore', invalid='ignore'):
            Dinv = (f32(1) / D).astype(f32)
        cur = nxt - half                     # working next point (minus half window)
        out = nxt.copy()
        prev_delta = np.zeros((n, 2), f32)
        running = act.copy()
        out[act] = (cur[act] + half)
        for j in range(max_count):
            if not running.any():
                break
            inx = np.floor(cur).astype(np.int64)
            oob2 = running & ((inx[:, 0] < -win) | (inx[:, 0] >= w) | (inx[:, 1] < -win) | (inx[:, 1] >= h))
            if level == 0:
                status &= ~oob2
            running = running & ~oob2
            if not running.any():
                break
            fa = (cur[:, 0] - inx[:, 0].astype(f32)).astype(f32)
            fb = (cur[:, 1] - inx[:, 1].astype(f32)).astype(f32)
            w4 = weights(fa, fb)
            diff = sample(J, inx[:, 0], inx[:, 1], w4, 14 - 5) - Ip
            b1 = fsum(diff * dIx) * FLT_SCALE
            b2 = fsum(diff * dIy) * FLT_SCALE
            dx = ((A12 * b2 - A22 * b1) * Dinv).astype(f32)
            dy = ((A12 * b1 - A11 * b2) * Dinv).astype(f32)
            delta = np.stack([dx, dy], 1)
            cur[running] = cur[running] + delta[running]
            out[running] = cur[running] + half
            small = running & ((dx * dx + dy * dy) <= eps2)
            osc = running & ~small & (j > 0) & (np.abs(dx + prev_delta[:, 0]) < f32(0.01)) & \
                (np.abs(dy + prev_delta[:, 1]) < f32(0.01))
            out[osc] = out[osc] - delta[osc] * f32(0.5)
            running = running & ~small & ~osc
            prev_delta[running] = delta[running]
        nxt = np.where(act[:, None], out, nxt).astype(f32)
        if level == 0:
            e = nxt - half
            ie = np.floor(e).astype(np.int64)
            oob3 = status & ((ie[:, 0] < -win) | (ie[:, 0] >= w) | (ie[:, 1] < -win) | (ie[:, 1] >= h))
            status &= ~oob3
            fa = (e[:, 0] - ie[:, 0].astype(f32)).astype(f32)
            fb = (e[:, 1] - ie[:, 1].astype(f32)).astype(f32)
            w4 = weights(fa, fb)
            diff = np.abs(sample(J, ie[:, 0], ie[:, 1], w4, 14 - 5) - Ip)
            ev = fsum(diff) * f32(1.) / f32(32 * win * win)
            err = np.where(status & act, ev, err).astype(f32)
    return nxt, status.astype(np.uint8), err


# ----------------------------------------------------------------------------
# goodFeaturesToTrack / FAST   (imgproc/featureselect.cpp, corner.cpp; features2d/fast.cpp)
# ----------------------------------------------------------------------------
def corner_min_eigen_val(img, block_size=3):
    """cornerMinEigenVal(ksize=3): Sobel * 1/(4*block*255), products, unnormalised box sum,
    (a + c) - sqrt((a - c)^2 + b^2) with a = sum(dx^2)/2, c = sum(dy^2)/2, b = sum(dxdy).
    Borders BORDER_REFLECT_101 (the crop is an isolated image)."""
    f32 = np.float32
    p = np.pad(img.astype(np.int64), 1, mode='reflect')
    gx = (p[:-2, 2:] + 2 * p[1:-1, 2:] + p[2:, 2:]) - (p[:-2, :-2] + 2 * p[1:-1, :-2] + p[2:, :-2])
    gy = (p[2:, :-2] + 2 * p[2:, 1:-1] + p[2:, 2:]) - (p[:-2, :-2] + 2 * p[:-2, 1:-1] + p[:-2, 2:])
    scale = f32(1.) / (f32(4.) * f32(block_size) * f32(255.))
    dx, dy = gx.astype(f32) * scale, gy.astype(f32) * scale
    r = block_size // 2

    def box(a):
        q = np.pad(a, r, mode='reflect')
        h, w = a.shape
        rows = []
        for j in range(block_size):
            rs = np.zeros((h, w), f32)
            for i in range(block_size):
                rs = (rs + q[j:j + h, i:i + w]).astype(f32)
            rows.append(rs)
        s = np.zeros((h, w), f32)
        for rs in rows:
            s = (s + rs).astype(f32)
        return s
    a = box(dx * dx) * f32(0.5)
    b = box(dx * dy)
    c = box(dy * dy) * f32(0.5)
    return ((a + c) - np.sqrt((a - c) * (a - c) + b * b)).astype(f32)


def good_features_to_track(img, mask, max_corners, quality, min_distance, block_size=3):
    """cv2.goodFeaturesToTrack(img, mask=mask, maxCorners, qualityLevel, minDistance, blockSize).
    Returns [n, 2] float32 (x, y) in strongest-first order."""
    eig = corner_min_eigen_val(img, block_size)
    h, w = eig.shape
    m = mask != 0
    if not m.any():
        return np.empty((0, 2), np.float32)
    thr = np.float32(eig[m].max() * np.float32(quality)) if True else 0
    thr = np.float32(np.float32(eig[m].max()) * np.float32(quality))
    cands = []
    for y in range(1, h - 1):
        for x in range(1, w - 1):
            v = eig[y, x]
            if not (v > thr) or v == 0 or not m[y, x]:
                continue
            if v >= eig[y - 1:y + 2, x - 1:x + 2].max():
                cands.append((float(v), y * w + x))
    cands.sort(key=lambda t: (-t[0], -t[1]))
    out = []
    md2 = min_distance * min_distance
    for _, idx in cands:
        y, x = divmod(idx, w)
        if all((x - ox) ** 2 + (y - oy) ** 2 >= md2 for ox, oy in out):
            out.append((x, y))
            if len(out) >= max_corners:
                break
    return np.array(out, np.float32).reshape(-1, 2)


_FAST_DX = [0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1]
_FAST_DY = [3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3]


def fast_detect(img, threshold=10):
    """FastFeatureDetector(threshold, nonmaxSuppression=True, TYPE_9_16).detect -> [n,2] (x, y)
    in raster order.  Score = largest threshold for which the pixel is still a corner."""
    h, w = img.shape
    v = img.astype(np.int64)
    score = np.zeros((h, w), np.int64)
    c = v[3:h - 3, 3:w - 3]
    d = np.stack([c - v[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx] for dx, dy in zip(_FAST_DX, _FAST_DY)])
    best = np.zeros_like(c)
    for s in range(16):
        arc = np.stack([d[(s + k) % 16] for k in range(9)])
        best = np.maximum(best, np.maximum(arc.min(0), (-arc).min(0)))
    score[3:h - 3, 3:w - 3] = np.where(best > threshold, best - 1, 0)
    s = score
    p = np.pad(s, 1)
    nb = np.stack([p[1 + dy:h + 1 + dy, 1 + dx:w + 1 + dx] for dy in (-1, 0, 1) for dx in (-1, 0, 1)
                   if (dx, dy) != (0, 0)])
    keep = (s > 0) & (s > nb.max(0))
    ys, xs = np.nonzero(keep)
    return np.stack([xs, ys], 1).astype(np.float32)


# ----------------------------------------------------------------------------
# RANSAC model fitting  (calib3d/ptsetreg.cpp, fundam.cpp, levmarq.cpp, core/rand.cpp)
# ----------------------------------------------------------------------------
class CvRNG:
    """cv::RNG multiply-with-carry generator; RANSACPointSetRegistrator seeds it with (uint64)-1."""

    def __init__(self, state=0xffffffffffffffff):
        self.state = state

    def next(self):
        self.state = ((self.state & 0xffffffff) * 4164903690 + (self.state >> 32)) & 0xffffffffffffffff
        return self.state & 0xffffffff

    def uniform(self, a, b):
        return a if a == b else int(self.next() % (b - a) + a)


def _have_collinear(p):
    i = len(p) - 1
    eps = float(np.finfo(np.float32).eps)
    for j in range(i):
        dx1, dy1 = float(p[j][0]) - float(p[i][0]), float(p[j][1]) - float(p[i][1])
        for k in range(j):
            dx2, dy2 = float(p[k][0]) - float(p[i][0]), float(p[k][1]) - float(p[i][1])
            if abs(dx2 * dy1 - dy2 * dx1) <= eps * (abs(dx1) + abs(dy1) + abs(dx2) + abs(dy2)):
                return True
    return False


class HomographyModel:
    model_points, n_params = 4, 8

    def check_subset(self, a, b):
        if _have_collinear(a) or _have_collinear(b):
            return False
        if len(a) == 4:
            neg = 0
            for t in ((0, 1, 2), (1, 2, 3), (0, 2, 3), (0, 1, 3)):
                A = np.array([[a[i][0], a[i][1], 1.] for i in t], np.float64)
                B = np.array([[b[i][0], b[i][1], 1.] for i in t], np.float64)
                neg += np.linalg.det(A) * np.linalg.det(B) < 0
            if neg not in (0, 4):
                return False
        return True

    def run_kernel(self, M, m):
        M = np.asarray(M, np.float64); m = np.asarray(m, np.float64)
        n = len(M)
        cm, cM = m.mean(0), M.mean(0)
        sm, sM = np.abs(m - cm).sum(0), np.abs(M - cM).sum(0)
        if (np.abs(sm) < np.finfo(float).eps).any() or (np.abs(sM) < np.finfo(float).eps).any():
            return None
        sm, sM = n / sm, n / sM
        x, y = (m[:, 0] - cm[0]) * sm[0], (m[:, 1] - cm[1]) * sm[1]
        X, Y = (M[:, 0] - cM[0]) * sM[0], (M[:, 1] - cM[1]) * sM[1]
        z, o = np.zeros(n), np.ones(n)
        Lx = np.stack([X, Y, o, z, z, z, -x * X, -x * Y, -x], 1)
        Ly = np.stack([z, z, z, X, Y, o, -y * X, -y * Y, -y], 1)
        LtL = Lx.T @ Lx + Ly.T @ Ly
        w, v = np.linalg.eigh(LtL)
        h0 = v[:, 0].reshape(3, 3)
        invHnorm = np.array([[1. / sm[0], 0, cm[0]], [0, 1. / sm[1], cm[1]], [0, 0, 1]])
        Hnorm2 = np.array([[sM[0], 0, -cM[0] * sM[0]], [0, sM[1], -cM[1] * sM[1]], [0, 0, 1]])
        H = invHnorm @ h0 @ Hnorm2
        if abs(H[2, 2]) < np.finfo(float).tiny or not np.isfinite(H).all():
            return None
        return (H / H[2, 2]).reshape(9)

    def compute_error(self, M, m, H):
        f32 = np.float32
        Hf = H.astype(f32)
        M = np.asarray(M, f32); m = np.asarray(m, f32)
        ww = f32(1) / (Hf[6] * M[:, 0] + Hf[7] * M[:, 1] + f32(1))
        dx = (Hf[0] * M[:, 0] + Hf[1] * M[:, 1] + Hf[2]) * ww - m[:, 0]
        dy = (Hf[3] * M[:, 0] + Hf[4] * M[:, 1] + Hf[5]) * ww - m[:, 1]
        return (dx * dx + dy * dy).astype(f32)

    def to_params(self, H):
        return H[:8].copy()

    def from_params(self, h):
        return np.append(h, 1.)

    def residuals(self, M, m, h, want_j=True):
        M = np.asarray(M, np.float64); m = np.asarray(m, np.float64)
        ww = h[6] * M[:, 0] + h[7] * M[:, 1] + 1.
        ww = np.where(np.abs(ww) > np.finfo(float).eps, 1. / ww, 0.)
        xi = (h[0] * M[:, 0] + h[1] * M[:, 1] + h[2]) * ww
        yi = (h[3] * M[:, 0] + h[4] * M[:, 1] + h[5]) * ww
        r = np.stack([xi - m[:, 0], yi - m[:, 1]], 1).reshape(-1)
        if not want_j:
            return r, None
        z = np.zeros(len(M))
        j0 = np.stack([M[:, 0] * ww, M[:, 1] * ww, ww, z, z, z, -M[:, 0] * ww * xi, -M[:, 1] * ww * xi], 1)
        j1 = np.stack([z, z, z, M[:, 0] * ww, M[:, 1] * ww, ww, -M[:, 0] * ww * yi, -M[:, 1] * ww * yi], 1)
        return r, np.stack([j0, j1], 1).reshape(-1, 8)


class AffinePartialModel:
    model_points, n_params = 2, 4

    def check_subset(self, a, b):
        return not _have_collinear(a)

    def run_kernel(self, f, t):
        (x1, y1), (x2, y2) = (float(f[0][0]), float(f[0][1])), (float(f[1][0]), float(f[1][1]))
        (X1, Y1), (X2, Y2) = (float(t[0][0]), float(t[0][1])), (float(t[1][0]), float(t[1][1]))
        with np.errstate(divide='ignore', invalid='ignore'):
            d = np.float64(1.) / np.float64((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2))
            S0 = d * ((X1 - X2) * (x1 - x2) + (Y1 - Y2) * (y1 - y2))
            S1 = d * ((Y1 - Y2) * (x1 - x2) - (X1 - X2) * (y1 - y2))
            S2 = d * ((Y1 - Y2) * (x1 * y2 - x2 * y1) - (X1 * y2 - X2 * y1) * (y1 - y2) - (X1 * x2 - X2 * x1) * (x1 - x2))
            S3 = d * (-(X1 - X2) * (x1 * y2 - x2 * y1) - (Y1 * x2 - Y2 * x1) * (x1 - x2) - (Y1 * y2 - Y2 * y1) * (y1 - y2))
        return np.array([S0, -S1, S2, S1, S0, S3, 0, 0, 1], np.float64)

    def compute_error(self, f, t, M):
        f32 = np.float32
        F = M.astype(f32)
        f = np.asarray(f, f32); t = np.asarray(t, f32)
        with np.errstate(invalid='ignore', over='ignore'):
            a = F[0] * f[:, 0] + F[1] * f[:, 1] + F[2] - t[:, 0]
            b = F[3] * f[:, 0] + F[4] * f[:, 1] + F[5] - t[:, 1]
            return (a * a + b * b).astype(f32)

    def to_params(self, M):
        return np.array([M[0], M[3], M[2], M[5]])

    def from_params(self, h):
        return np.array([h[0], -h[1], h[2], h[1], h[0], h[3], 0, 0, 1], np.float64)

    def residuals(self, f, t, h, want_j=True):
        f = np.asarray(f, np.float64); t = np.asarray(t, np.float64)
        r = np.stack([h[0] * f[:, 0] - h[1] * f[:, 1] + h[2] - t[:, 0],
                      h[1] * f[:, 0] + h[0] * f[:, 1] + h[3] - t[:, 1]], 1).reshape(-1)
        if not want_j:
            return r, None
        o, z = np.ones(len(f)), np.zeros(len(f))
        j0 = np.stack([f[:, 0], -f[:, 1], o, z], 1)
        j1 = np.stack([f[:, 1], f[:, 0], z, o], 1)
        return r, np.stack([j0, j1], 1).reshape(-1, 4)


def _update_iters(p, ep, mp, max_iters):
    p = min(max(p, 0.), 1.)
    ep = min(max(ep, 0.), 1.)
    num = max(1. - p, np.finfo(float).tiny)
    denom = 1. - (1. - ep) ** mp
    if denom < np.finfo(float).tiny:
        return 0
    num, denom = np.log(num), np.log(denom)
    return max_iters if denom >= 0 or -num >= max_iters * (-denom) else int(np.rint(num / denom))


def ransac_run(cb, m1, m2, threshold, confidence, max_iters):
    """RANSACPointSetRegistrator::run -> (model[9] | None, mask[n] bool)."""
    m1 = np.asarray(m1, np.float32).reshape(-1, 2); m2 = np.asarray(m2, np.float32).reshape(-1, 2)
    count, mp = len(m1), cb.model_points
    if count < mp:
        return None, np.zeros(count, bool)
    if count == mp:
        M = cb.run_kernel(m1, m2)
        return (M, np.ones(count, bool)) if M is not None else (None, np.zeros(count, bool))
    rng = CvRNG()
    niters = max(max_iters, 1)
    best, best_mask, max_good = None, np.zeros(count, bool), 0
    thr2 = np.float32(threshold * threshold)
    it = 0
    while it < niters:
        found = False
        for _ in range(1000):
            idx = []
            for i in range(mp):
                while True:
                    v = rng.uniform(0, count)
                    if v not in idx:
                        break
                idx.append(v)
            if not cb.check_subset(m1[idx], m2[idx]):
                continue
            found = True
            break
        if not found:
            if it == 0:
                return None, np.zeros(count, bool)
            break
        M = cb.run_kernel(m1[idx], m2[idx])
        if M is not None:
            err = cb.compute_error(m1, m2, M)
            mask = err <= thr2
            good = int(mask.sum())
            if good > max(max_good, mp - 1):
                best, best_mask, max_good = M, mask, good
                niters = _update_iters(confidence, (count - good) / count, mp, niters)
        it += 1
    return (best, best_mask) if max_good > 0 else (None, np.zeros(count, bool))


def _sym_solve(A, b):
    w, v = np.linalg.eigh(A)
    thr = np.finfo(float).eps * 2 * np.abs(w).max() * len(w)
    inv = np.where(np.abs(w) > thr, 1. / np.where(w == 0, 1, w), 0.)
    return v @ (inv * (v.T @ b)), np.einsum('ki,i,ki->k', v, inv, v)


def lm_refine(cb, a, b, M, max_iters=10):
    """LMSolverImpl::run (levmarq.cpp) with eps = FLT_EPSILON."""
    x = cb.to_params(M)
    r, J = cb.residuals(a, b, x)
    S = float(r @ r)
    A, v = J.T @ J, J.T @ r
    D = np.diag(A).copy()
    lam, lc = 1., 0.75
    eps = float(np.finfo(np.float32).eps)
    it = 0
    while True:
        Ap = A + np.diag(lam * D)
        d, _ = _sym_solve(Ap, v)
        xd = x - d
        rd, _ = cb.residuals(a, b, xd, want_j=False)
        Sd = float(rd @ rd)
        dS = float(d @ (2 * v - A @ d))
        R = (S - Sd) / (dS if abs(dS) > np.finfo(float).eps else 1)
        if R > 0.75:
            lam *= 0.5
            if lam < lc:
                lam = 0
        elif R < 0.25:
            t = float(d @ v)
            nu = (Sd - S) / (t if abs(t) > np.finfo(float).eps else 1) + 2
            nu = min(max(nu, 2.), 10.)
            if lam == 0:
                _, inv_diag = _sym_solve(A, np.zeros(len(x)))
                lam = lc = 1. / max(np.finfo(float).eps, np.abs(inv_diag).max())
                nu *= 0.5
            lam *= nu
        if Sd < S:
            S, x = Sd, xd
            r, J = cb.residuals(a, b, x)
            A, v = J.T @ J, J.T @ r
        it += 1
        if not (it < max_iters and np.abs(d).max() >= eps and np.abs(r).max() >= eps):
            break
    return cb.from_params(x)


def fg_keep(pt, boxes, size):
    """flow.py:310-324 (_fg_filter) for one matched point: its rounded position is inside the frame and on none of the
    boxes estimated for the closer tracks so far (the reference zeroes those in fg_mask, flow.py:261-263)"""
    x, y = int(np.rint(pt[0])), int(np.rint(pt[1]))
    if x < 0 or y < 0 or x >= size[0] or y >= size[1]:
        return False
    return not any(b[0] <= x <= b[2] and b[1] <= y <= b[3] for b in boxes)


def crop_box(e):
    """the pixel rectangle rect.crop() addresses for a box (rect.py:83-89)"""
    return [max(int(e[0]), 0), max(int(e[1]), 0), max(int(e[2]), 0), max(int(e[3]), 0)]


def estimate_bbox(tb, M):
    """flow.py:274-280 (_estimate_bbox) with M = the six coefficients of the 2x3 partial-affine matrix, row-major"""
    tlx = tb[0] * M[0] + tb[1] * M[1] + M[2]
    tly = tb[0] * M[3] + tb[1] * M[4] + M[5]
    scale = np.sqrt(M[0] * M[0] + M[3] * M[3])
    if scale < 0.9 or scale > 1.1:
        scale = 1.
    w, h = tb[2] - tb[0] + 1, tb[3] - tb[1] + 1
    return np.rint([tlx, tly, tlx + w * scale - 1., tly + h * scale - 1.])


def flow_estimate(prev_pts, cur_pts, status, begins, ends, bg_begin, bg_end, track_tlbr, size,
                  ransac_max_iter, ransac_conf, inlier_thresh):
    """Second half of Flow.predict (flow.py:215-263) with cv2.findHomography /
    cv2.estimateAffinePartial2D restated.  Same outputs as fm_flow_estimate."""
    P = np.asarray(prev_pts, np.float32).reshape(-1, 2)
    Cc = np.asarray(cur_pts, np.float32).reshape(-1, 2)
    st = np.asarray(status, bool)
    nT = len(begins)
    inl = np.zeros(len(P), bool)
    result = np.zeros(nT, np.int32)
    est = np.zeros((nT, 4))
    n_matched = np.zeros(nT, np.int32)
    g = np.arange(bg_begin, bg_end)[st[bg_begin:bg_end]]
    if len(g) < 4:
        return None, result, est, n_matched, inl
    hcb = HomographyModel()
    H, mask = ransac_run(hcb, P[g], Cc[g], 3.0, ransac_conf, ransac_max_iter)
    n_in = int(mask.sum())
    if H is not None and len(g) > 4 and n_in > 0:
        H2 = hcb.run_kernel(P[g][mask], Cc[g][mask])
        if H2 is not None:
            H = lm_refine(hcb, P[g][mask], Cc[g][mask], H2, 10)
    if H is None or n_in < inlier_thresh:
        return None, result, est, n_matched, inl
    inl[g[mask]] = True
    acb = AffinePartialModel()
    boxes = []
    for k in range(nT):
        idx = []
        for i in range(begins[k], ends[k]):
            if st[i] and fg_keep(Cc[i], boxes, size):
                idx.append(i)
        n = len(idx)
        n_matched[k] = n
        if n < 3:
            continue
        idx = np.array(idx)
        M, mask = ransac_run(acb, P[idx], Cc[idx], 3.0, ransac_conf, ransac_max_iter)
        if M is None:
            continue
        if n > 2 and mask.any():
            M = lm_refine(acb, P[idx][mask], Cc[idx][mask], M, 10)
        e = estimate_bbox(track_tlbr[k], M)
        inl[idx[mask]] = True
        est[k] = e
        outside = min(e[2], size[0] - 1) < max(e[0], 0) or min(e[3], size[1] - 1) < max(e[1], 0)
        if outside or int(mask.sum()) < inlier_thresh:
            result[k] = 2
            continue
        result[k] = 1
        boxes.append(crop_box(e))
    return H.reshape(3, 3), result, est, n_matched, inl
