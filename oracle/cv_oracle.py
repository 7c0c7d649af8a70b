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


def bgr2gray(img):
    """cv2.cvtColor(COLOR_BGR2GRAY) for uint8 (imgproc/color_rgb: RGB2Gray<uchar>, 15-bit fixed
    point in 4.x: B*3735 + G*19235 + R*9798, descale (x + (1 << 14)) >> 15)."""
    b = img[..., 0].astype(np.int64)
    g = img[..., 1].astype(np.int64)
    r = img[..., 2].astype(np.int64)
    return ((b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15).astype(np.uint8)


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
