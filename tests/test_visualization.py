"""Overlay renderer (fastmot_amd/utils/visualization.py; next row n4 of SURVEY section 8f): the interface of
the reference's Visualizer, track colours equal to the reference's get_color, geometry of the primitives."""
from collections import deque
from types import SimpleNamespace

import numpy as np

from fastmot_amd.utils.visualization import Visualizer, get_color, covariance_ellipse


# track id -> BGR returned by the reference's get_color (fastmot/utils/visualization.py:51-56), generated in the
# build container by executing that file under oracle/ref_shim.py's stub modules:
#   spec_from_file_location('fastmot.utils.visualization', '/root/reference/fastmot/utils/visualization.py')
GOLDEN_COLORS = {0: (50, 50, 255), 1: (250, 108, 50), 2: (49, 245, 163), 3: (217, 48, 241), 17: (253, 245, 50),
                 256: (45, 225, 171), 100000: (118, 229, 45)}


def test_track_colours():
    for idx, bgr in GOLDEN_COLORS.items():
        assert get_color(idx) == bgr
    assert len({get_color(i) for i in range(1, 40)}) == 39      # neighbouring ids get distinct colours


def _track(trk_id, tlbr, n_hist=9):
    boxes = deque([np.asarray(tlbr, float) + 3 * k for k in range(-n_hist + 1, 1)], maxlen=30)
    cov = np.diag([9., 4., 16., 25., 1., 1., 1., 1.])
    return SimpleNamespace(trk_id=trk_id, tlbr=boxes[-1], bboxes=boxes, state=(np.zeros(8), cov),
                           keypoints=np.array([[60.4, 70.6], [80., 90.]], np.float32),
                           prev_keypoints=np.array([[58., 69.], [77., 88.]], np.float32))


def test_render_in_place_and_flags():
    frame = np.full((240, 320, 3), 128, np.uint8)
    tracks = [_track(7, [50, 60, 120, 200]), _track(8, [200, 30, 260, 150])]
    dets = np.rec.array([((10., 20., 40., 90.), 1, 0.75)], dtype=[('tlbr', float, 4), ('label', int), ('conf', float)])
    klt = [np.array([150., 100., 180., 160.])]
    bg_prev, bg_cur = np.array([[300., 200.]], np.float32), np.array([[305.2, 203.7]], np.float32)

    base = frame.copy()
    Visualizer().render(base, tracks, dets, klt, bg_prev, bg_cur)
    # track boxes: 2 px outline in the track's colour at the integer corners, nothing else drawn
    for t in tracks:
        x0, y0, x1, y1 = t.tlbr.astype(int)
        col = np.array(get_color(t.trk_id))
        assert (base[y1, x0:x1 + 1] == col).all() and (base[y0 + 30:y1, x0] == col).all() and (base[y0 + 30:y1, x1] == col).all()
        assert (base[y1 - 1, x0 + 2:x1 - 1] == col).all()                     # second pixel of the thick line
        assert (base[(y0 + y1) // 2, (x0 + x1) // 2] == 128).all()             # interior untouched
    assert (base[20:91, 10] == 128).all() and (base[100:161, 150] == 128).all() and (base[204, 305] == 128).all()

    full = frame.copy()
    Visualizer(draw_detections=True, draw_confidence=True, draw_covariance=True, draw_klt=True,
               draw_obj_flow=True, draw_bg_flow=True, draw_trajectory=True).render(
        full, tracks, dets, klt, bg_prev, bg_cur, caption='visible: 2')
    assert (full[50:91, 10] == 255).all() and (full[90, 10:41] == 255).all()   # white detection box, 1 px
    assert (full[100:161, 150] == 0).all() and (full[160, 150:181] == 0).all() # black KLT box
    assert (full[204, 305] == (0, 0, 255)).all()                               # background keypoint, red (BGR)
    assert (full[71, 60] == (0, 255, 255)).all() and (full[90, 80] == (0, 255, 255)).all()   # object keypoints
    assert (full != base).sum() > 500                                          # trajectory, ellipses, texts
    assert frame.min() == 128 and frame.max() == 128                           # only the passed array changes


def test_covariance_ellipse():
    (a, b), ang = covariance_ellipse(np.array([[9., 0.], [0., 4.]]))
    assert (a, b) == (int(3 * np.sqrt(5.9915) + 0.5), int(2 * np.sqrt(5.9915) + 0.5)) and abs(abs(ang) % 180) < 1e-9
    (a, b), ang = covariance_ellipse(np.array([[5., 4.], [4., 5.]]))           # principal axis at 45 degrees
    assert a > b and abs(abs(ang) - 45) < 1e-6 or abs(abs(ang) - 135) < 1e-6
