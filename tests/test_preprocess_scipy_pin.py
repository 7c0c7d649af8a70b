"""YOLO preprocessing (detector.py:289-300): the reference resizes with cupyx.scipy.ndimage.zoom(order=1, mode='opencv',
grid_mode=True), which CuPy implements as affine_transform(zoom = in / out, offset = (zoom - 1) / 2, mode='nearest').
CuPy is not available; SciPy's affine_transform -- the routine the CuPy one mirrors -- is, and pins the coordinate
mapping, the edge handling and the interpolation of the oracle's restatement (np_oracle.zoom_linear_opencv): the uint8
results agree everywhere except on exact .5 ties, where SciPy's integer cast adds 0.5 and truncates while CuPy's kernel
calls rint() (half to even), which the oracle follows."""
import numpy as np
import pytest
from scipy import ndimage

import np_oracle as o


@pytest.mark.parametrize('frame_hw,roi_hw', [((1080, 1920), (608, 608)), ((720, 1280), (288, 512)), ((1080, 1920), (360, 640)),
                                             ((480, 640), (608, 608)), ((2160, 3840), (720, 1280)), ((37, 53), (16, 24))])
def test_resize_equals_scipy_affine_transform_up_to_tie_rounding(frame_hw, roi_hw):
    rng = np.random.default_rng(sum(frame_hw) + roi_hw[0])
    frame = rng.integers(0, 256, (*frame_hw, 3), dtype=np.uint8)
    rh, rw = roi_hw
    exact = o.zoom_linear_opencv(frame, rh, rw)
    mine = np.clip(np.rint(exact), 0, 255).astype(np.uint8)
    zoom = np.array([frame_hw[0] / rh, frame_hw[1] / rw, 1.0])
    ref = ndimage.affine_transform(frame, np.diag(zoom), (zoom - 1) / 2 * [1, 1, 0], (rh, rw, 3), order=1, mode='nearest',
                                   prefilter=False)
    diff = ref.astype(int) - mine.astype(int)
    tie = np.abs(exact - np.floor(exact) - 0.5) < 1e-9
    assert np.all(diff[~tie] == 0)
    assert np.all(np.abs(diff[tie]) <= 1)
    if all(float(z * 4).is_integer() for z in zoom):
        # dyadic weights: both sides evaluate the tie exactly; SciPy gives floor + 1, rint the even neighbour -- they
        # differ exactly where floor is even (with other zooms the two evaluations of a near-tie may fall on either side)
        even_floor = np.floor(exact[tie]).astype(int) % 2 == 0
        np.testing.assert_array_equal(diff[tie], even_floor.astype(int))
        assert tie.sum() > 100 or all(float(z).is_integer() for z in zoom)      # (integer zoom: no interpolation at all)


def test_preprocess_uses_that_resize():
    rng = np.random.default_rng(5)
    frame = rng.integers(0, 256, (270, 480, 3), dtype=np.uint8)
    out = o.yolo_preprocess(frame, (160, 288), roi=(0, 8, 288, 144))        # letterbox: rows 8..151
    inner = np.rint(out[:, 8:152] * 255).astype(np.uint8)[::-1].transpose(1, 2, 0)
    np.testing.assert_array_equal(inner, np.clip(np.rint(o.zoom_linear_opencv(frame, 144, 288)), 0, 255).astype(np.uint8))
    assert np.all(out[:, :8] == 0.5) and np.all(out[:, 152:] == 0.5)
