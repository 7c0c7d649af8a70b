"""TEST INFRASTRUCTURE ONLY -- build container only.  Pins the oracle's goldens with a REAL Numba.

The committed goldens (tests/golden/*.npz) come from the unmodified reference with Numba's decorators replaced by the
identity (oracle/ref_shim.py) plus a restatement of Numba's typed set where its iteration order shows in the track IDs
(oracle/numba_set.py).  That left open what the jit-compiled reference really computes: fastmath reassociation, typed
containers, integer / float typing of the njit'ed bodies.  This script closes it as far as this image allows:

    /opt/conda/bin/python3.9 oracle/pin_with_numba.py          (Numba 0.54.1; the reference pins 0.48, requirements.txt:3)

1. re-runs every golden generator (make_golden.py, make_golden_mot.py, make_golden_ssd.py) with the reference's
   @nb.njit functions compiled by the real Numba (ref_shim.load_reference(real_numba=True)) into a scratch directory
   and compares array by array with the committed files: integer / bool arrays must be equal, float arrays are
   reported with their largest absolute difference;
2. compares oracle/numba_set.difference_order with `list(set(range(n)) - set(removed))` inside an @njit function over
   ~1000 (n, removed) cases and writes them, with the real answers, to tests/golden/numba_set_order.npz -- the fixture
   tests/test_setorder.py checks the restatement AND the product's utils/setorder.py against on every machine;
3. writes tests/golden/flow_helpers_kat.npz: the @njit glue of Flow.predict that needs no OpenCV, run jit-compiled on
   seeded inputs (the oracle's own restatement of that glue is tested against it, tests/test_flow_helpers_kat.py);
4. writes tests/golden/REAL_NUMBA_PIN.json (versions, per-file verdicts) and tests/golden/real_numba_floats.npz: the
   float arrays of the real-Numba run that differ from the committed ones ("<file>:<array>"), so that the restatement
   can be held against what the jit-compiled reference computes, not only against its de-jitted source.

Nothing under /root/reference is written (NUMBA_CACHE_DIR and byte-code writing are redirected, oracle/real_numba.py).
"""
import json
import runpy
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / 'oracle'))
sys.path.insert(0, str(ROOT / 'tests'))
sys.dont_write_bytecode = True

import real_numba  # noqa: E402

numba, NUMPY_VERSION = real_numba.import_numba()
import numpy as np  # noqa: E402
import ref_shim  # noqa: E402
import numba_set  # noqa: E402

GOLDEN = ROOT / 'tests' / 'golden'


def regenerate(tmp):
    """Runs the three generators with the real Numba, their np.savez_compressed calls redirected to `tmp`."""
    real_load = ref_shim.load_reference
    real_save = np.savez_compressed

    def load(prefix='fastmot', real_numba=True):
        return real_load(prefix, real_numba=True)

    def save(path, **arrays):
        real_save(Path(tmp) / Path(path).name, **arrays)
    ref_shim.load_reference = load
    np.savez_compressed = save
    try:
        for script in ('make_golden.py', 'make_golden_mot.py', 'make_golden_ssd.py'):
            for key in [k for k in sys.modules if k == 'fastmot' or k.startswith('fastmot.')]:
                del sys.modules[key]
            runpy.run_path(str(ROOT / 'oracle' / script), run_name='__main__')
    finally:
        ref_shim.load_reference = real_load
        np.savez_compressed = real_save


def compare(tmp):
    verdicts = {}
    real_floats = {}          # float arrays of the real-Numba run that differ from the committed (de-jitted) ones
    for new in sorted(Path(tmp).glob('*.npz')):
        old = GOLDEN / new.name
        if not old.exists():
            verdicts[new.name] = {'status': 'no committed file'}
            continue
        a, b = np.load(old, allow_pickle=False), np.load(new, allow_pickle=False)
        rec = {'arrays': len(a.files), 'equal': 0, 'float_close': {}, 'different': []}
        if sorted(a.files) != sorted(b.files):
            rec['different'].append('key sets differ')
        for k in a.files:
            if k not in b.files:
                continue
            x, y = a[k], b[k]
            if x.shape != y.shape or x.dtype != y.dtype:
                rec['different'].append(f'{k}: {x.dtype}{x.shape} vs {y.dtype}{y.shape}')
            elif np.array_equal(x, y, equal_nan=x.dtype.kind == 'f'):
                rec['equal'] += 1
            elif x.dtype.kind == 'f':
                rec['float_close'][k] = float(np.nanmax(np.abs(x - y)))
                real_floats[f'{new.stem}:{k}'] = y
            else:
                rec['different'].append(f'{k}: {int((x != y).sum())} of {x.size} elements')
        rec['status'] = ('identical' if rec['equal'] == rec['arrays'] else
                         'float differences only' if not rec['different'] else 'DIFFERENT')
        verdicts[new.name] = rec
    np.savez_compressed(GOLDEN / 'real_numba_floats.npz', **real_floats)
    return verdicts


def golden_flow_helpers(ns):
    """The @njit glue of Flow.predict that needs no OpenCV (flow.py:266-364: _estimate_feature_dist, _estimate_bbox,
    _rect_filter, _ellipse_filter, _fg_filter, _scale_pts, _unscale_pts, _get_status, _get_good_match, _get_inliers) and
    the rect.py helpers it is built on (intersection, crop), on seeded inputs -- generated with the REAL Numba only: the
    de-jitted bodies do not even run on empty inputs (`np.array([])` of an empty list comprehension is float64 in NumPy
    and cannot index; Numba types it int64)."""
    F = ns.flow.Flow
    rect = ns.rect
    rng = np.random.default_rng(23)
    out = {}
    areas = np.concatenate([np.arange(0, 400), rng.integers(400, 200000, 300), [1736, 1737, 6944, 6945, 15625, 27777, 27778]])
    out['fd_area'] = areas.astype(np.int64)
    out['fd_dist'] = np.array([F._estimate_feature_dist(int(a), 0.06) for a in areas], np.int64)
    W, H = 160, 120
    frame_rect = np.array([0., 0., W - 1., H - 1.])
    n_cases = 24
    for c in range(n_cases):
        fg = np.full((H, W), 255, np.uint8)
        holes = []
        for _ in range(int(rng.integers(0, 4))):                       # earlier (closer) tracks already zeroed
            b = np.sort(rng.integers(-10, W + 10, 2)).tolist() + np.sort(rng.integers(-10, H + 10, 2)).tolist()
            hole = np.array([b[0], b[2], b[1], b[3]], float)            # (flow.py:261-263 crops the estimated box itself)
            rect.crop(fg, hole)[:] = 0
            holes.append(hole)
        out[f'c{c}_holes'] = np.array(holes, float).reshape(-1, 4)
        tl = rng.uniform(-20, [W - 20, H - 20])
        tlbr = np.rint(np.concatenate([tl, tl + rng.uniform(8, 90, 2)]))
        ins = rect.intersection(tlbr, frame_rect)
        if ins is None:
            tlbr = np.array([10., 12., 70., 90.])
            ins = rect.intersection(tlbr, frame_rect)
        n = 0 if c == 0 else int(rng.integers(1, 60))
        pts = (rng.uniform(-6, 6, (n, 2)) + rng.uniform(ins[:2] - 4, ins[2:] + 4, (n, 2))).astype(np.float32)
        if n > 4:
            pts[0] = ins[:2] - 0.5                                      # rint half cases on the rectangle's edge
            pts[1] = ins[2:] + 0.5
            pts[2] = ins[:2] + np.float32(0.5)
        out[f'c{c}_fg'] = fg.copy()
        out[f'c{c}_tlbr'] = tlbr
        out[f'c{c}_ins'] = ins
        out[f'c{c}_pts'] = pts
        out[f'c{c}_rect'] = F._rect_filter(pts, ins, fg)
        local = rng.uniform(-2, [ins[2] - ins[0] + 3, ins[3] - ins[1] + 3], (max(n, 3), 2)).astype(np.float32).reshape(-1, 1, 2)
        out[f'c{c}_gftt'] = local
        out[f'c{c}_ellipse'] = F._ellipse_filter(local, tlbr, ins[:2])
        # LK outputs -> status, unscale (masked, in place on a copy), good matches, fg filter, inliers
        m = max(n, 5)
        prev = rng.uniform(0, [W, H], (m, 2)).astype(np.float32)
        cur = (prev * np.float32(0.5) + rng.normal(0, 1.5, (m, 2)).astype(np.float32)).reshape(-1, 1, 2)
        st = (rng.random((m, 1)) < 0.8).astype(np.uint8)
        err = rng.uniform(0, 160, (m, 1)).astype(np.float32)
        status = F._get_status(st, err, 100)
        cur_un = F._unscale_pts(cur.copy(), (0.5, 0.5), status)
        out.update({f'c{c}_prev': prev, f'c{c}_cur': cur, f'c{c}_st': st, f'c{c}_err': err, f'c{c}_status': status,
                    f'c{c}_cur_un': cur_un, f'c{c}_scaled': F._scale_pts(prev, (0.5, 0.5)),
                    f'c{c}_bg_un': F._unscale_pts(prev.copy(), (0.1, 0.1))})
        b, e = int(rng.integers(0, m // 2)), int(rng.integers(m // 2, m + 1))
        gp, gc = F._get_good_match(prev, cur_un, status, b, e)
        fp, fc = F._fg_filter(gp, gc, fg, (W, H))
        inl = (rng.random((len(fc), 1)) < 0.7).astype(np.uint8)
        ip, ic = F._get_inliers(fp, fc, inl)
        out.update({f'c{c}_range': np.array([b, e]), f'c{c}_good_prev': gp, f'c{c}_good_cur': gc, f'c{c}_fgf_prev': fp,
                    f'c{c}_fgf_cur': fc, f'c{c}_inl': inl, f'c{c}_inl_prev': ip, f'c{c}_inl_cur': ic})
        ang, sc = rng.normal(0, 0.05), rng.choice([0.85, 0.9, 0.95, 1.0, 1.04, 1.1, 1.12]) * (1 + rng.normal(0, 1e-3))
        A = np.array([[sc * np.cos(ang), -sc * np.sin(ang), rng.normal(0, 6)],
                      [sc * np.sin(ang), sc * np.cos(ang), rng.normal(0, 6)]])
        out[f'c{c}_affine'] = A
        out[f'c{c}_est'] = F._estimate_bbox(tlbr, A)
    out['n_cases'] = np.array(n_cases)
    np.savez_compressed(GOLDEN / 'flow_helpers_kat.npz', **out)
    print('flow_helpers_kat: ok', len(out), 'arrays')



def pin_set_order():
    @numba.njit
    def real(n, removed):
        return list(set(range(n)) - set(removed))

    @numba.njit
    def real_discard(n, removed):
        keep = set(range(n))
        for k in removed:
            keep.discard(k)
        return list(keep)
    rng = np.random.default_rng(0)
    ns, offs, rem, out_d, out_k, out_off = [], [0], [], [], [], [0]
    bad = 0
    for n in list(range(0, 70)) + [100, 127, 128, 129, 200, 255, 256, 257, 300, 511, 512, 513, 700]:
        for trial in range(12):
            k = 0 if trial == 0 else n if trial == 1 else int(rng.integers(0, n + 1))
            r = rng.permutation(n)[:k].astype(np.int64)
            d = np.array(real(n, r), np.int64)
            kd = np.array(real_discard(n, r), np.int64)          # detector.py:196-211: set(range(n)), discard() one by one
            mine = numba_set.difference_order(n, [int(x) for x in r])
            s = numba_set.NumbaIntSet(range(n))
            for x in r:
                s.discard(int(x))
            bad += list(d) != mine or list(kd) != list(s)
            ns.append(n); rem.append(r); offs.append(offs[-1] + k)
            out_d.append(d); out_k.append(kd); out_off.append(out_off[-1] + len(d))
            assert len(d) == len(kd) == n - k
    np.savez_compressed(GOLDEN / 'numba_set_order.npz', n=np.array(ns, np.int64), removed=np.concatenate(rem),
                        removed_off=np.array(offs, np.int64), difference=np.concatenate(out_d),
                        discarded=np.concatenate(out_k), out_off=np.array(out_off, np.int64))
    return len(ns), bad


def quick_check():
    """--quick-check: a subset that fits a unit test (tests/test_real_numba_pin.py runs it where this interpreter and
    /root/reference exist): the Kalman / association / NMS known-answer files and two tracker scenes regenerated with the
    jit-compiled reference and compared with the committed goldens, the set-order restatement on every tenth case.
    Writes nothing under tests/golden; exit code 1 on any integer difference or a float difference above 1e-9."""
    import importlib
    with tempfile.TemporaryDirectory() as tmp:
        real_save = np.savez_compressed
        np.savez_compressed = lambda path, **arrays: real_save(Path(tmp) / Path(path).name, **arrays)
        try:
            mg = importlib.import_module('make_golden')
            ns = ref_shim.load_reference(real_numba=True)
            mg.golden_kalman(ns)
            mg.golden_assoc(ns)
            mg.golden_nms(ns)
            all_scenes = mg.scenes.SCENES
            mg.scenes.SCENES = {k: all_scenes[k] for k in ('s8_flowfail', 's16_blackout_confirm3')}
            try:
                mg.golden_tracker(ns)
            finally:
                mg.scenes.SCENES = all_scenes
        finally:
            np.savez_compressed = real_save
        keep = globals()['GOLDEN']
        verdicts = {}
        for new in sorted(Path(tmp).glob('*.npz')):
            a, b = np.load(keep / new.name), np.load(new)
            worst, bad = 0.0, []
            for k in a.files:
                x, y = a[k], b[k]
                if x.shape != y.shape or x.dtype != y.dtype:
                    bad.append(k)
                elif x.dtype.kind == 'f':
                    if x.size:
                        worst = max(worst, float(np.nanmax(np.abs(x - y))))
                elif not np.array_equal(x, y):
                    bad.append(k)
            verdicts[new.name] = (bad, worst)
            print(f'{new.name:40s} integer arrays {"identical" if not bad else "DIFFERENT " + str(bad)}, floats within {worst:.3g}')
    @numba.njit
    def real(n, removed):
        return list(set(range(n)) - set(removed))
    z = np.load(GOLDEN / 'numba_set_order.npz')
    mism = 0
    for i in range(0, len(z['n']), 10):
        n = int(z['n'][i])
        r = z['removed'][z['removed_off'][i]:z['removed_off'][i + 1]]
        mism += list(real(n, r)) != numba_set.difference_order(n, r.tolist())
    print(f'set order: {len(range(0, len(z["n"]), 10))} cases, {mism} mismatches')
    ok = mism == 0 and all(not bad and worst <= 1e-9 or (not bad and name.startswith('assoc') and worst <= 1e-7)
                           for name, (bad, worst) in verdicts.items())
    print('numba', numba.__version__, 'numpy', NUMPY_VERSION, 'OK' if ok else 'FAILED')
    return 0 if ok else 1


def main():
    if '--quick-check' in sys.argv:
        sys.exit(quick_check())
    with tempfile.TemporaryDirectory() as tmp:
        regenerate(tmp)
        verdicts = compare(tmp)
    cases, bad = pin_set_order()
    for key in [k for k in sys.modules if k == 'fastmot' or k.startswith('fastmot.')]:
        del sys.modules[key]
    golden_flow_helpers(ref_shim.load_reference(real_numba=True))
    rec = {'numba': numba.__version__, 'numpy': NUMPY_VERSION, 'python': sys.version.split()[0],
           'reference_pins': 'numba==0.48 (requirements.txt:3)', 'goldens': verdicts,
           'set_order': {'cases': cases, 'restatement_mismatches': bad, 'fixture': 'tests/golden/numba_set_order.npz'}}
    (GOLDEN / 'REAL_NUMBA_PIN.json').write_text(json.dumps(rec, indent=1) + '\n')
    for name, v in verdicts.items():
        print(f'{name:42s} {v["status"]}' + (f'  max |diff| {max(v["float_close"].values()):.3g} in {len(v["float_close"])} arrays'
                                             if v.get('float_close') else '') + (f'  {v["different"][:3]}' if v.get('different') else ''))
    print(f'set order: {cases} cases, {bad} mismatches of the restatement')


if __name__ == '__main__':
    main()
