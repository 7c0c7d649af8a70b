"""TEST INFRASTRUCTURE -- golden for MOT.step itself: runs the UNMODIFIED reference fastmot/mot.py (under
oracle/ref_shim.py, with the scripted Detector / FeatureExtractor fakes of tests/scenes.py in place of the TensorRT
classes and the scripted Flow) on the three-class scene, one extractor per class id.  Pins the step schedule,
_split_bboxes_by_cls with the reference's bisect_right (every box goes to the FIRST extractor) and the
float64 concatenation of the per-extractor embeddings.  Build container only:

    python oracle/make_golden_mot.py        -> tests/golden/mot_s40_multiclass.npz
"""
import importlib.util
import sys
import types
from pathlib import Path
from types import SimpleNamespace

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT / 'oracle'), str(ROOT / 'tests')]

import ref_shim  # noqa: E402
import scenes  # noqa: E402

NAME = 's40_multiclass_reid'


def load_reference_mot():
    ns = ref_shim.load_reference()
    root = ref_shim.REF_ROOT / 'fastmot'

    def _load(modname, relpath):
        full = 'fastmot.' + modname
        spec = importlib.util.spec_from_file_location(full, root / relpath)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[full] = mod
        spec.loader.exec_module(mod)
        parent, _, leaf = full.rpartition('.')
        setattr(sys.modules[parent], leaf, mod)
        return mod
    prof = _load('utils.profiler', 'utils/profiler.py')
    sys.modules['fastmot.utils'].Profiler = prof.Profiler
    _load('utils.visualization', 'utils/visualization.py')
    # the TensorRT-backed classes are replaced by the scripted fakes BEFORE mot.py imports them
    det = types.ModuleType('fastmot.detector')
    det.SSDDetector = det.YOLODetector = det.PublicDetector = scenes.FakeDetector
    ext = types.ModuleType('fastmot.feature_extractor')
    ext.FeatureExtractor = scenes.FakeExtractor
    sys.modules['fastmot.detector'], sys.modules['fastmot.feature_extractor'] = det, ext
    ns.mot = _load('mot', 'mot.py')
    return ns


def main():
    ns = load_reference_mot()
    scene = scenes.Scene(NAME)
    scenes.bind_fakes(scene)
    ns.track.Track._count = 0
    kw = scenes.tracker_kwargs(NAME)
    mot = ns.mot.MOT(scene.size, detector_type='YOLO', detector_frame_skip=scene.skip, class_ids=(0, 1, 2),
                     feature_extractor_cfgs=(SimpleNamespace(), SimpleNamespace(), SimpleNamespace()),
                     tracker_cfg=SimpleNamespace(**kw))
    records = scenes.run_scene_mot(mot, scene)
    out = scenes.pack_records(records, None)
    out['extractor_log'] = np.array(scenes.FakeExtractor.log, np.int64)
    np.savez_compressed(ROOT / 'tests' / 'golden' / 'mot_s40_multiclass.npz', **out)
    log = out['extractor_log']
    print(f'{NAME}: {len(out["tracks"])} track rows, max id {int(out["tracks"][:, 2].max())}; boxes per extractor: '
          f'{[int(log[log[:, 1] == i, 2].sum()) for i in range(3)]}')


if __name__ == '__main__':
    main()
