"""Overlay renderer with the interface of the reference's Visualizer (fastmot/utils/visualization.py:100-146:
same constructor flags, `render(frame, tracks, detections, klt_bboxes, prev_bg_keypoints, bg_keypoints)`
drawing IN PLACE on the BGR frame), built on Pillow because this image has no OpenCV.

What matches the reference exactly: which primitives are drawn for which flag, their geometry (integer box
corners as `tlbr.astype(int)`, rounded keypoints, every 4th box centre of a trajectory, the 95 % covariance
ellipses) and the golden-ratio track colours (`get_color`, visualization.py:51-56).  What does not: the
rasterisation itself (OpenCV's anti-aliased lines and Hershey fonts vs Pillow's) -- overlays are for people,
the tracker never reads them back.
"""
import colorsys

import numpy as np
from PIL import Image, ImageDraw, ImageFont

GOLDEN_RATIO = 0.618033988749895


def get_color(idx, s=0.8, vmin=0.7):
    """BGR colour of a track id: hue and value walk by the golden ratio (visualization.py:51-56)."""
    step = idx * GOLDEN_RATIO
    hue = np.fmod(step, 1.)
    value = 1. - np.fmod(step, 1. - vmin)
    r, g, b = colorsys.hsv_to_rgb(hue, s, value)
    return int(255 * b), int(255 * g), int(255 * r)


class _Canvas:
    """Pillow drawing context over a BGR ndarray; `commit` writes the pixels back in place."""

    def __init__(self, frame):
        self.frame = frame
        self.img = Image.fromarray(np.ascontiguousarray(frame[..., ::-1]))
        self.draw = ImageDraw.Draw(self.img)
        self.font = ImageFont.load_default()

    @staticmethod
    def rgb(bgr):
        if np.isscalar(bgr):
            return (int(bgr),) * 3
        return int(bgr[2]), int(bgr[1]), int(bgr[0])

    def commit(self):
        self.frame[...] = np.asarray(self.img)[..., ::-1]


def draw_bbox(canvas, tlbr, color, thickness, text=None):
    x0, y0, x1, y1 = (int(v) for v in np.asarray(tlbr).astype(int))
    fill = canvas.rgb(color)
    canvas.draw.rectangle([min(x0, x1), min(y0, y1), max(x0, x1), max(y0, y1)], outline=fill, width=thickness)
    if text is not None:
        l, t, r, b = canvas.draw.textbbox((x0, y0), text, font=canvas.font)
        canvas.draw.rectangle([x0, y0, x0 + (r - l) + 1, y0 + (b - t) + 2], fill=fill)
        canvas.draw.text((x0 + 1, y0 - (t - y0) + 1), text, fill=(0, 0, 0), font=canvas.font)


def draw_trajectory(canvas, bboxes, trk_id):
    boxes = np.reshape(list(bboxes), (len(bboxes), 4))[::4]
    centers = ((boxes[:, :2] + boxes[:, 2:]) / 2).astype(np.int32)         # get_center + int32 cast
    if len(centers) > 1:
        canvas.draw.line([tuple(int(v) for v in c) for c in centers], fill=canvas.rgb(get_color(trk_id)), width=1)


def draw_feature_match(canvas, prev_pts, cur_pts, color):
    if len(cur_pts) == 0:
        return
    fill = canvas.rgb(color)
    cur = np.rint(cur_pts).astype(np.int32)
    for x, y in cur:
        canvas.draw.ellipse([x - 1, y - 1, x + 1, y + 1], fill=fill)
    if len(prev_pts) > 0:
        prev = np.rint(prev_pts).astype(np.int32)
        for (xa, ya), (xb, yb) in zip(prev, cur):
            canvas.draw.line([(int(xa), int(ya)), (int(xb), int(yb))], fill=fill, width=1)


def covariance_ellipse(cov):
    """Semi-axes (rounded) and angle in degrees of the 95 % confidence ellipse of a 2x2 covariance
    (visualization.py:85-92)."""
    vals, vecs = np.linalg.eigh(cov)
    order = vals.argsort()[::-1]
    vals, vecs = np.sqrt(vals[order] * 5.9915), vecs[:, order]
    return (int(vals[0] + 0.5), int(vals[1] + 0.5)), float(np.degrees(np.arctan2(vecs[1, 0], vecs[0, 0])))


def draw_covariance(canvas, tlbr, covariance):
    x0, y0, x1, y1 = (int(v) for v in np.asarray(tlbr).astype(int))
    for (cx, cy), cov in (((x0, y0), covariance[:2, :2]), ((x1, y1), covariance[2:4, 2:4])):
        (a, b), angle = covariance_ellipse(cov)
        t = np.linspace(0, 2 * np.pi, 73)
        ca, sa = np.cos(np.radians(angle)), np.sin(np.radians(angle))
        xs = cx + a * np.cos(t) * ca - b * np.sin(t) * sa
        ys = cy + a * np.cos(t) * sa + b * np.sin(t) * ca
        canvas.draw.line([(float(x), float(y)) for x, y in zip(xs, ys)], fill=(255, 255, 255), width=1)


class Visualizer:
    def __init__(self,
                 draw_detections=False,
                 draw_confidence=False,
                 draw_covariance=False,
                 draw_klt=False,
                 draw_obj_flow=False,
                 draw_bg_flow=False,
                 draw_trajectory=False):
        """Flags as fastmot/utils/visualization.py:101-131: detections (+ confidence text), Kalman position
        covariance ellipses, KLT-predicted boxes, per-object and background flow matches, box trajectories."""
        self.draw_detections = draw_detections
        self.draw_confidence = draw_confidence
        self.draw_covariance = draw_covariance
        self.draw_klt = draw_klt
        self.draw_obj_flow = draw_obj_flow
        self.draw_bg_flow = draw_bg_flow
        self.draw_trajectory = draw_trajectory

    def render(self, frame, tracks, detections, klt_bboxes, prev_bg_keypoints, bg_keypoints, caption=None):
        """Draws onto `frame` (HxWx3 uint8 BGR) in place.  `caption` (not in the reference's signature) is the
        'visible: N' text MOT._draw adds with cv2.putText (mot.py:195-196)."""
        canvas = _Canvas(frame)
        for track in tracks:                                   # thick box + id label in the track's colour
            draw_bbox(canvas, track.tlbr, get_color(track.trk_id), 2, str(track.trk_id))
            if self.draw_trajectory:
                draw_trajectory(canvas, track.bboxes, track.trk_id)
            if self.draw_obj_flow:
                draw_feature_match(canvas, track.prev_keypoints, track.keypoints, (0, 255, 255))
            if self.draw_covariance:
                draw_covariance(canvas, track.tlbr, track.state[1])
        if self.draw_detections:                               # thin white boxes, optional "label: conf"
            for det in detections:
                draw_bbox(canvas, det.tlbr, (255, 255, 255), 1,
                          f'{det.label}: {det.conf:.2f}' if self.draw_confidence else None)
        if self.draw_klt:                                      # thin black boxes where the KLT put the tracks
            for tlbr in klt_bboxes:
                draw_bbox(canvas, tlbr, (0, 0, 0), 1)
        if self.draw_bg_flow:                                  # red matches of the camera-motion keypoints
            draw_feature_match(canvas, prev_bg_keypoints, bg_keypoints, (0, 0, 255))
        if caption:
            canvas.draw.text((30, 14), caption, fill=(0, 0, 0), font=canvas.font)
        canvas.commit()
