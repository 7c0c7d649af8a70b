"""`import fastmot` drop-in (reference app.py:10-12): the reference's application file is executed UNMODIFIED
(runpy, in a subprocess) against the alias package `fastmot/` under a `cv2` stub that only provides the window
calls app.py itself makes.  /root/reference exists in the build container only, so this test is CPU-side (it
skips on the GPU box); the same flow with the tracker switched on runs on the GPU in
test_app_gpu.py::test_reference_cli_flow_with_alias_package."""
import json
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
REF_APP = Path('/root/reference/app.py')

RUNNER = r'''
import sys, types, runpy, json, inspect
sys.path.insert(0, {root!r})
cv2 = types.ModuleType('cv2')
cv2.WINDOW_AUTOSIZE = 1
cv2.namedWindow = lambda *a, **k: None
cv2.getWindowProperty = lambda *a, **k: 0
cv2.imshow = lambda *a, **k: None
cv2.waitKey = lambda *a, **k: -1
cv2.destroyAllWindows = lambda *a, **k: None
sys.modules['cv2'] = cv2
import fastmot
calls = []
if {fake_mot!r}:
    real = fastmot.MOT
    class FakeMOT:
        """Records how the reference app drives MOT and checks that the real constructor accepts the call."""
        def __init__(self, size, **kw):
            inspect.signature(real.__init__).bind(None, size, **kw)
            calls.append(('init', list(size), sorted(kw)))
            self.frame_count = 0
        def reset(self, dt):
            calls.append(('reset', dt))
        def step(self, frame):
            assert frame.shape == (180, 320, 3) and frame.dtype.name == 'uint8'
            self.frame_count += 1
        def visible_tracks(self):
            return iter(())
        @staticmethod
        def print_timing_info():
            calls.append(('timing',))
    fastmot.MOT = FakeMOT
sys.argv = ['app.py'] + {argv!r}
try:
    runpy.run_path({app!r}, run_name='__main__')
finally:
    json.dump(calls, open({log!r}, 'w'))
'''


def _make_sequence(tmp_path, n=6, size=(320, 180)):
    from PIL import Image
    rng = np.random.default_rng(0)
    seq = tmp_path / 'seq'
    seq.mkdir()
    frames = []
    for i in range(1, n + 1):
        f = rng.integers(0, 256, (size[1], size[0], 3), dtype=np.uint8)
        Image.fromarray(f[:, :, ::-1]).save(seq / f'{i:06d}.png')
        frames.append(f)
    return seq, frames


def _config(tmp_path, size):
    cfg = json.loads((REF_APP.parent / 'cfg' / 'mot.json').read_text())
    cfg['resize_to'] = list(size)
    path = tmp_path / 'mot.json'
    path.write_text(json.dumps(cfg))
    return path


def _run(tmp_path, argv, fake_mot):
    log = tmp_path / 'calls.json'
    code = RUNNER.format(root=str(ROOT), app=str(REF_APP), argv=argv, log=str(log), fake_mot=fake_mot)
    res = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    return json.loads(log.read_text()), res


@pytest.mark.skipif(not REF_APP.exists(), reason='/root/reference only exists in the build container')
def test_reference_app_runs_unmodified_capture_and_write(tmp_path):
    """Reference app.py, no tracker: VideoIO reads the sequence and writes every frame back unchanged."""
    from PIL import Image
    seq, frames = _make_sequence(tmp_path)
    cfg = _config(tmp_path, (320, 180))
    out = tmp_path / 'out'
    _run(tmp_path, ['-i', str(seq / '%06d.png'), '-c', str(cfg), '-o', str(out / '%06d.png'), '-q'], False)
    written = sorted(out.glob('*.png'))
    assert len(written) == len(frames)
    for path, f in zip(written, frames):
        np.testing.assert_array_equal(np.asarray(Image.open(path))[:, :, ::-1], f)


@pytest.mark.skipif(not REF_APP.exists(), reason='/root/reference only exists in the build container')
def test_reference_app_drives_mot_with_reference_config(tmp_path):
    """Reference app.py -m with the reference's own cfg/mot.json: the constructor call binds against
    fastmot_amd.MOT's signature (every mot_cfg key is accepted), reset/step/visible_tracks/print_timing_info are
    the calls it makes, and the MOTChallenge result file is created."""
    seq, frames = _make_sequence(tmp_path)
    cfg = _config(tmp_path, (320, 180))
    txt = tmp_path / 'res' / 'out.txt'
    calls, _ = _run(tmp_path, ['-i', str(seq / '%06d.png'), '-c', str(cfg), '-m', '-t', str(txt)], True)
    kinds = [c[0] for c in calls]
    assert kinds == ['init', 'reset', 'timing']
    assert calls[0][1] == [320, 180]
    assert set(calls[0][2]) == {'detector_type', 'detector_frame_skip', 'class_ids', 'ssd_detector_cfg',
                                'yolo_detector_cfg', 'public_detector_cfg', 'feature_extractor_cfgs', 'tracker_cfg',
                                'visualizer_cfg', 'draw'}
    assert abs(calls[1][1] - 1 / 30) < 1e-12
    assert txt.exists()


def test_alias_package_is_the_same_implementation():
    """`fastmot` re-exports, it does not re-implement: same objects under both names, loggers chained."""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import logging, fastmot, fastmot.models, fastmot_amd\n"
            "from fastmot.utils import ConfigDecoder, Profiler\n"
            "import fastmot.utils.visualization as viz, fastmot.tracker as trk\n"
            "assert fastmot.MOT is fastmot_amd.MOT and fastmot.VideoIO is fastmot_amd.VideoIO\n"
            "assert fastmot.FeatureExtractor is fastmot_amd.FeatureExtractor\n"
            "assert fastmot.models is fastmot_amd.models and trk.MultiTracker is fastmot_amd.MultiTracker\n"
            "assert Profiler is fastmot_amd.utils.Profiler and viz.Visualizer is fastmot_amd.utils.Visualizer\n"
            "logging.getLogger('fastmot').setLevel(logging.ERROR)\n"
            "assert logging.getLogger('fastmot_amd.tracker').getEffectiveLevel() == logging.ERROR\n" % str(ROOT))
    res = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr[-2000:]


def test_alias_submodule_imports_before_attribute_access():
    """`import fastmot.mot` / `from fastmot.detector import ...` as the FIRST thing a user does (the import system
    does not consult a module-level __getattr__ for submodules: a meta-path finder maps them)."""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from fastmot.mot import MOT\n"
            "import fastmot.detector, fastmot.feature_extractor, fastmot.utils\n"
            "import fastmot_amd.mot, fastmot_amd.detector\n"
            "assert MOT is fastmot_amd.mot.MOT and fastmot.detector is fastmot_amd.detector\n"
            "assert sys.modules['fastmot.mot'] is fastmot_amd.mot\n"
            "try:\n"
            "    import fastmot.no_such_module\n"
            "    raise SystemExit('imported a module that does not exist')\n"
            "except ModuleNotFoundError:\n"
            "    pass\n" % str(ROOT))
    res = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr[-2000:]
