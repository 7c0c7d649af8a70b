"""MOTChallenge I/O: PublicDetector against the reference's own class (imported under the
shim, build container only), the result-row format of app.py:91-97, and the CLEAR-MOT / IDF1 scorer on
hand-checkable cases."""
import importlib.util
import io
import sys
from types import SimpleNamespace

import numpy as np
import pytest

import ref_shim
from fastmot_amd.detector import PublicDetector
from fastmot_amd.utils import motchallenge as mc


def make_sequence(root, n_frames=12, seed=0):
    rng = np.random.default_rng(seed)
    (root / 'det').mkdir(parents=True)
    (root / 'seqinfo.ini').write_text('[Sequence]\nname=SYN-01\nimDir=img1\nframeRate=30\nseqLength=12\n'
                                      'imWidth=1920\nimHeight=1080\nimExt=.jpg\n')
    rows = []
    for f in range(1, n_frames + 1):
        for _ in range(rng.integers(0, 6)):
            x, y = rng.uniform(-20, 1800), rng.uniform(-20, 1000)
            w, h = rng.uniform(5, 1500), rng.uniform(5, 900)
            rows.append(f'{f},-1,{x:.2f},{y:.2f},{w:.2f},{h:.2f},{rng.uniform(0, 1):.3f},-1,-1,-1')
    (root / 'det' / 'det.txt').write_text('\n'.join(rows) + '\n')


@pytest.mark.skipif(not ref_shim.reference_available(), reason='reference tree only exists in the build container')
@pytest.mark.parametrize('size,frame_skip', [((1280, 720), 1), ((1920, 1080), 3), ((640, 360), 2)])
def test_public_detector_equals_reference(tmp_path, size, frame_skip):
    make_sequence(tmp_path / 'seq')
    ns = ref_shim.load_reference()
    try:
        sys.modules['fastmot.utils'].TRTInference = object
        for mod, rel in (('utils.decoder', 'utils/decoder.py'), ('detector', 'detector.py')):
            spec = importlib.util.spec_from_file_location('fastmot.' + mod, ref_shim.REF_ROOT / 'fastmot' / rel)
            m = importlib.util.module_from_spec(spec)
            sys.modules['fastmot.' + mod] = m
            spec.loader.exec_module(m)
        ref = m.PublicDetector(size, (1,), frame_skip, sequence_path=str(tmp_path / 'seq'), max_area=400000)
        ours = PublicDetector(size, (1,), frame_skip, sequence_path=str(tmp_path / 'seq'), max_area=400000)
        for _ in range(6):
            ref.detect_async(None)
            ours.detect_async(None)
            a, b = ref.postprocess(), ours.postprocess()
            assert len(a) == len(b)
            np.testing.assert_array_equal(a.tlbr, b.tlbr)
            np.testing.assert_array_equal(a.label, b.label)
            np.testing.assert_array_equal(a.conf, b.conf)
    finally:
        ref_shim.unload_reference(ns)


def test_result_rows_format():
    tracks = [SimpleNamespace(trk_id=7, tlbr=np.array([10., 20., 109., 219.])),
              SimpleNamespace(trk_id=12, tlbr=np.array([0., 0., 1279., 719.]))]
    buf = io.StringIO()
    mc.write_rows(buf, 3, tracks, (1280, 720), (1920, 1080))
    assert buf.getvalue() == ('3,7,15.000000,30.000000,149.500000,299.500000,-1,-1,-1\n'
                              '3,12,0.000000,0.000000,1919.500000,1079.500000,-1,-1,-1\n')


def test_scorer_hand_cases(tmp_path):
    # two objects, 10 frames, constant boxes
    gt = {f: [(1, np.array([100., 100., 50., 100.])), (2, np.array([400., 100., 50., 100.]))] for f in range(1, 11)}
    same = {f: [(i + 10, b.copy()) for i, b in v] for f, v in gt.items()}
    r = mc.evaluate(gt, same)
    assert (r['mota'], r['idf1'], r['fp'], r['fn'], r['idsw']) == (1.0, 1.0, 0, 0, 0) and abs(r['motp'] - 1) < 1e-12
    # identity switch of object 1 at frame 6, object 2 missed in frames 9-10, one false positive in frame 3
    res = {}
    for f in range(1, 11):
        rows = [(11 if f < 6 else 13, gt[f][0][1].copy())]
        if f < 9:
            rows.append((12, gt[f][1][1].copy()))
        if f == 3:
            rows.append((99, np.array([900., 500., 40., 80.])))
        res[f] = rows
    r = mc.evaluate(gt, res)
    assert (r['fp'], r['fn'], r['idsw'], r['n_gt']) == (1, 2, 1, 20)
    assert abs(r['mota'] - (1 - 4 / 20)) < 1e-12
    # IDF1: best identity map 1->11 (5 frames) or 1->13 (5), 2->12 (8): IDTP = 13, n_res = 19, n_gt = 20
    assert abs(r['idf1'] - 2 * 13 / (19 + 20)) < 1e-12
    # a box that overlaps with IoU < 0.5 is a miss and a false positive
    shifted = {1: [(5, np.array([140., 100., 50., 100.]))]}
    r = mc.evaluate({1: [(1, np.array([100., 100., 50., 100.]))]}, shifted)
    assert (r['fp'], r['fn'], r['tp']) == (1, 1, 0)
    # file round trip
    p = tmp_path / 'res.txt'
    with open(p, 'w') as f:
        for fr, rows in res.items():
            for i, b in rows:
                f.write(f'{fr},{i},{b[0]:.6f},{b[1]:.6f},{b[2]:.6f},{b[3]:.6f},-1,-1,-1\n')
    back = mc.read_txt(p)
    assert mc.evaluate(gt, back) == r or mc.evaluate(gt, back)['idsw'] == 1


@pytest.mark.skipif(not (ref_shim.REF_ROOT / 'eval' / 'results' / 'MOT20-01.txt').exists(),
                    reason='reference artefact only exists in the build container')
def test_reference_result_file_parses_and_self_scores():
    res = mc.read_txt(ref_shim.REF_ROOT / 'eval' / 'results' / 'MOT20-01.txt')
    assert len(res) > 100 and all(len(v) > 0 for v in res.values())
    sub = {f: res[f] for f in sorted(res)[:60]}
    r = mc.evaluate(sub, sub)
    assert r['mota'] == 1.0 and r['idf1'] == 1.0 and r['idsw'] == 0
