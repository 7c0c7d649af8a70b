"""TEST INFRASTRUCTURE -- generates tests/golden/*.npz by running the UNMODIFIED reference
(/root/reference, via oracle/ref_shim.py) on seeded inputs.  Run here (build container) only:

    python oracle/make_golden.py

Outputs (committed):
  tests/golden/tracker_<scene>.npz   per-frame track ids / rounded boxes / lifecycle flags /
                                      history order + final Kalman states of the reference
                                      MultiTracker (tracker.py:18-422) on tests/scenes.py scenes
  tests/golden/kalman_kat.npz        KalmanFilter create/warp/predict/update/motion_distance
  tests/golden/assoc_kat.npz         cdist / iou_dist / find_occluded / fuse+gate / LAP / greedy
  tests/golden/nms_kat.npz           diou_nms + YOLODetector._filter_dets
"""
import sys
from pathlib import Path
from types import SimpleNamespace

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / 'oracle'))
sys.path.insert(0, str(ROOT / 'tests'))

import ref_shim  # noqa: E402
import scenes  # noqa: E402

GOLDEN = ROOT / 'tests' / 'golden'


def golden_tracker(ns):
    for name in scenes.SCENES:
        scene = scenes.Scene(name)
        ns.track.Track._count = 0
        tracker = ns.tracker.MultiTracker(scene.size, scene.metric, **scenes.tracker_kwargs(name))
        records, final = scenes.run_scene(tracker, scene)
        out = scenes.pack_records(records, final)
        np.savez_compressed(GOLDEN / f'tracker_{name}.npz', **out)
        ids = np.unique(out['tracks'][:, 2])
        print(f'{name}: {len(out["tracks"])} track rows, {len(ids)} ids, max id {int(ids.max())}, '
              f'hist rows {len(out["hist"])}')


def golden_kalman(ns):
    rng = np.random.default_rng(7)
    kf = ns.kalman_filter.KalmanFilter()
    MeasType = ns.kalman_filter.MeasType
    out = {}
    for tag, dt in (('dt30', 1 / 30.), ('dt12', 1 / 12.5)):
        kf.reset_dt(dt)
        n = 64
        tl = np.stack([rng.uniform(0, 1700, n), rng.uniform(0, 800, n)], 1)
        boxes = np.rint(np.concatenate([tl, tl + rng.uniform(20, 300, (n, 2))], 1))
        H = np.eye(3) + rng.normal(0, 2e-3, (3, 3))
        H[:2, 2] += rng.normal(0, 4, 2)
        H[2, :2] = rng.normal(0, 2e-6, 2)
        H[2, 2] = 1.
        klt = boxes + rng.normal(0, 4, (n, 4))
        mult = rng.uniform(1, 8, n)
        det = boxes + rng.normal(0, 6, (n, 4))
        stages = {k: [] for k in ('create_m', 'create_c', 'warp_m', 'warp_c', 'pred_m', 'pred_c',
                                  'klt_m', 'klt_c', 'det_m', 'det_c', 'maha')}
        vel = rng.normal(0, 25, (n, 4))
        for i in range(n):
            m, c = kf.create(boxes[i])
            stages['create_m'].append(m.copy()); stages['create_c'].append(c.copy())
            m = m.copy(); m[4:] = vel[i]
            # a non-diagonal covariance: one predict step first
            m, c = kf.predict(m, c)
            m0, c0 = m.copy(), c.copy()
            m, c = kf.warp(m, c, H)
            stages['warp_m'].append(m.copy()); stages['warp_c'].append(c.copy())
            m, c = kf.predict(m, c)
            stages['pred_m'].append(m.copy()); stages['pred_c'].append(c.copy())
            m, c = kf.update(m, c, klt[i], MeasType.FLOW, mult[i])
            stages['klt_m'].append(m.copy()); stages['klt_c'].append(c.copy())
            stages['maha'].append(kf.motion_distance(m, c, det))
            m, c = kf.update(m, c, det[i], MeasType.DETECTOR)
            stages['det_m'].append(m.copy()); stages['det_c'].append(c.copy())
            stages.setdefault('start_m', []).append(m0); stages.setdefault('start_c', []).append(c0)
        out.update({f'{tag}_{k}': np.array(v) for k, v in stages.items()})
        out.update({f'{tag}_boxes': boxes, f'{tag}_H': H, f'{tag}_klt': klt, f'{tag}_mult': mult,
                    f'{tag}_det': det, f'{tag}_dt': np.array(dt)})
    np.savez_compressed(GOLDEN / 'kalman_kat.npz', **out)
    print('kalman_kat: ok')


def golden_assoc(ns):
    rng = np.random.default_rng(11)
    out = {}
    Metric = ns.distance.Metric
    for tag, (nt, nd) in (('a', (17, 23)), ('b', (50, 50)), ('c', (5, 40)), ('d', (33, 6))):
        XA = rng.normal(0, 1, (nt, 512)); XA /= np.linalg.norm(XA, axis=1, keepdims=True)
        XA = XA.astype(np.float32).astype(np.float64)            # features are f32 averages copied to f64
        XB = rng.normal(0, 1, (nd, 512)); XB /= np.linalg.norm(XB, axis=1, keepdims=True)
        XB = XB.astype(np.float32)
        mask = rng.random((nt, nd)) < 0.15
        out[f'{tag}_XA'] = XA.astype(np.float32); out[f'{tag}_XB'] = XB; out[f'{tag}_mask'] = mask
        out[f'{tag}_cos'] = ns.distance.cdist(XA, XB, Metric.COSINE, mask, 0.9)
        out[f'{tag}_euc'] = ns.distance.cdist(XA, XB, Metric.EUCLIDEAN, mask, 0.9)
        tl = rng.uniform(0, 900, (nt, 2)); ta = np.rint(np.concatenate([tl, tl + rng.uniform(30, 200, (nt, 2))], 1))
        dl = rng.uniform(0, 900, (nd, 2)); db = np.rint(np.concatenate([dl, dl + rng.uniform(30, 200, (nd, 2))], 1))
        out[f'{tag}_ta'] = ta; out[f'{tag}_db'] = db
        out[f'{tag}_iou'] = ns.distance.iou_dist(ta, db)
        out[f'{tag}_occ'] = ns.rect.find_occluded(db, 0.7)
        out[f'{tag}_occ3'] = ns.rect.find_occluded(db, 0.3)
        # fuse + gate
        maha = rng.uniform(0, 20, (nt, nd))
        tlab = rng.integers(0, 2, nt); dlab = rng.integers(0, 2, nd)
        cost = out[f'{tag}_cos'].copy()
        for r in range(nt):
            ns.matching.fuse_motion(cost[r], maha[r], 0.2)
        ns.matching.gate_cost(cost, tlab, dlab, 0.8)
        out[f'{tag}_maha'] = maha; out[f'{tag}_tlab'] = tlab; out[f'{tag}_dlab'] = dlab
        out[f'{tag}_cost'] = cost
        m, ut, ud = ns.matching.linear_assignment(cost, list(range(100, 100 + nt)), list(range(nd)))
        out[f'{tag}_lap_m'] = np.array(m, np.int64).reshape(-1, 2)
        out[f'{tag}_lap_ut'] = np.array(ut, np.int64); out[f'{tag}_lap_ud'] = np.array(ud, np.int64)
        m, ut, ud = ns.matching.greedy_match(out[f'{tag}_iou'].copy(), list(range(100, 100 + nt)), list(range(nd)), 0.8)
        out[f'{tag}_gr_m'] = np.array(m, np.int64).reshape(-1, 2)
        out[f'{tag}_gr_ut'] = np.array(ut, np.int64); out[f'{tag}_gr_ud'] = np.array(ud, np.int64)
    np.savez_compressed(GOLDEN / 'assoc_kat.npz', **out)
    print('assoc_kat: ok')


def golden_nms(ns):
    """diou_nms (utils/rect.py:199-244) and YOLODetector._filter_dets (detector.py:322-365).
    detector.py itself cannot be imported (cupy/tensorrt at import time are stubbed but the module
    needs fastmot.utils.TRTInference); its staticmethod body is exec'd from the source file."""
    import ast
    src = (ref_shim.REF_ROOT / 'fastmot' / 'detector.py').read_text()
    tree = ast.parse(src)
    fn = None
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name == '_filter_dets':
            fn = node
    fn.decorator_list = []
    mod = ast.Module(body=[fn], type_ignores=[])
    glb = {'np': np, 'find_split_indices': ns.numba_utils.find_split_indices, 'diou_nms': ns.rect.diou_nms,
           'to_tlbr': ns.rect.to_tlbr, 'area': ns.rect.area, 'aspect_ratio': ns.rect.aspect_ratio}
    exec(compile(mod, 'detector.py', 'exec'), glb)
    filter_dets = glb['_filter_dets']
    rng = np.random.default_rng(21)
    out = {}
    for tag, (n_obj, n_cls) in (('p', (40, 1)), ('q', (120, 3))):
        # clusters of candidates around objects, as a YOLO head emits them (fractions of frame size)
        cx = rng.uniform(0.05, 0.9, n_obj); cy = rng.uniform(0.05, 0.8, n_obj)
        w = rng.uniform(0.02, 0.06, n_obj); h = rng.uniform(0.1, 0.25, n_obj)
        cls = rng.integers(0, n_cls, n_obj)
        rows = []
        for o in range(n_obj):
            for _ in range(int(rng.integers(2, 9))):
                jw, jh = w[o] * rng.uniform(0.85, 1.15), h[o] * rng.uniform(0.85, 1.15)
                rows.append([cx[o] + rng.normal(0, 0.004) - jw / 2, cy[o] + rng.normal(0, 0.004) - jh / 2, jw, jh,
                             rng.uniform(0.2, 1), cls[o], rng.uniform(0.5, 1)])
        for _ in range(400):   # low-score clutter
            rows.append([rng.uniform(0, 0.9), rng.uniform(0, 0.8), rng.uniform(0.01, 0.1), rng.uniform(0.02, 0.3),
                         rng.uniform(0, 0.3), rng.integers(0, n_cls + 1), rng.uniform(0, 1)])
        det_out = np.array(rows, np.float32)
        rng.shuffle(det_out)
        label_mask = np.zeros(n_cls + 1, bool); label_mask[:n_cls] = True
        size = np.array([1920, 1080]); offset = np.zeros(2)
        dets = filter_dets(det_out.copy(), size, offset, label_mask, 0.25, 0.5, 800000, 1.2)
        out[f'{tag}_det_out'] = det_out
        out[f'{tag}_tlbr'] = np.array([d[0] for d in dets]).reshape(-1, 4)
        out[f'{tag}_label'] = np.array([d[1] for d in dets], np.int64)
        out[f'{tag}_conf'] = np.array([d[2] for d in dets], np.float64)
        # stand-alone diou_nms on one class
        one = det_out[(det_out[:, 5] == 0) & (det_out[:, 4] * det_out[:, 6] >= 0.25)].copy()
        one[:, :4] *= np.append(size, size)
        out[f'{tag}_nms_in'] = one
        out[f'{tag}_nms_keep'] = ns.rect.diou_nms(one[:, :4], one[:, 4], 0.5).astype(np.int64)
    np.savez_compressed(GOLDEN / 'nms_kat.npz', **out)
    print('nms_kat: ok', {k: v.shape for k, v in out.items() if 'tlbr' in k})


if __name__ == '__main__':
    GOLDEN.mkdir(parents=True, exist_ok=True)
    ns = ref_shim.load_reference()
    golden_kalman(ns)
    golden_assoc(ns)
    golden_nms(ns)
    golden_tracker(ns)
