"""TEST INFRASTRUCTURE -- golden vectors for the SSD detector's host stages, produced by the UNMODIFIED
reference functions (fastmot/detector.py:122-217: SSDDetector._generate_tiles, _normalize, _filter_dets,
_merge_dets/_merge), exec'd from the source file because detector.py cannot be imported (TensorRT).
Run here (build container) only:   python oracle/make_golden_ssd.py   ->  tests/golden/ssd_kat.npz
"""
import ast
import sys
from pathlib import Path
from types import SimpleNamespace

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / 'oracle'))
import ref_shim  # noqa: E402

DET_DTYPE = np.dtype([('tlbr', float, 4), ('label', int), ('conf', float)], align=True)


def reference_functions(ns):
    src = (ref_shim.REF_ROOT / 'fastmot' / 'detector.py').read_text()
    cls = next(n for n in ast.walk(ast.parse(src)) if isinstance(n, ast.ClassDef) and n.name == 'SSDDetector')
    fns = [n for n in cls.body if isinstance(n, ast.FunctionDef) and
           n.name in ('_generate_tiles', '_normalize', '_filter_dets', '_merge_dets', '_merge')]
    for f in fns:
        f.decorator_list = []
    glb = {'np': np, 'nb': sys.modules['numba'], 'DET_DTYPE': DET_DTYPE}
    for name in ('to_tlbr', 'as_tlbr', 'get_size', 'area', 'iom', 'enclosing', 'multi_crop'):
        glb[name] = getattr(ns.rect, name)
    exec(compile(ast.Module(body=fns, type_ignores=[]), 'detector.py', 'exec'), glb)
    return SimpleNamespace(**{f.name: glb[f.name] for f in fns})


def synthetic_engine_output(rng, tiles, topk, n_obj, n_cls):
    """Objects placed in the tiling region; every tile that contains (most of) an object reports it with a
    little jitter, in tile fractions, rows sorted by confidence, padded with zero-confidence rows."""
    region = tiles[:, 2:].max(0) + 1
    cx, cy = rng.uniform(20, region[0] - 20, n_obj), rng.uniform(20, region[1] - 20, n_obj)
    w, h = rng.uniform(15, 60, n_obj), rng.uniform(40, 140, n_obj)
    cls = rng.integers(0, n_cls, n_obj)
    out = np.zeros((len(tiles), topk, 7), np.float32)
    for ti, t in enumerate(tiles):
        tw, th = t[2] - t[0] + 1, t[3] - t[1] + 1
        rows = []
        for o in range(n_obj):
            x0, y0, x1, y1 = cx[o] - w[o] / 2, cy[o] - h[o] / 2, cx[o] + w[o] / 2, cy[o] + h[o] / 2
            ix = max(0., min(x1, t[2]) - max(x0, t[0])) * max(0., min(y1, t[3]) - max(y0, t[1]))
            if ix < 0.5 * w[o] * h[o]:
                continue
            j = rng.normal(0, 1.5, 4)
            bx = np.clip([(x0 + j[0] - t[0]) / tw, (y0 + j[1] - t[1]) / th, (x1 + j[2] - t[0]) / tw,
                          (y1 + j[3] - t[1]) / th], 0, 1)
            rows.append([ti, cls[o], rng.uniform(0.3, 0.99), *bx])
        for _ in range(int(rng.integers(0, 6))):      # clutter, some below any threshold
            rows.append([ti, rng.integers(0, n_cls), rng.uniform(0.05, 0.6), *np.sort(rng.uniform(0, 1, 2)),
                         *np.sort(rng.uniform(0, 1, 2))][:7])
        rows = sorted(rows, key=lambda r: -r[2])[:topk]
        for k, r in enumerate(rows):
            out[ti, k] = [r[0], r[1], r[2], min(r[3], r[5]), min(r[4], r[6]), max(r[3], r[5]), max(r[4], r[6])]
    return out.reshape(-1)


if __name__ == '__main__':
    ns = ref_shim.load_reference()
    ref = reference_functions(ns)
    rng = np.random.default_rng(77)
    out = {}
    cases = {'a': dict(grid=(4, 2), overlap=0.25, size=(1920, 1080), topk=100, n_obj=40, n_cls=4, mask=(0, 1, 3),
                       thresh=0.5, merge=0.6, max_area=120000),
             'b': dict(grid=(3, 3), overlap=0.1, size=(1280, 720), topk=20, n_obj=25, n_cls=2, mask=(1,),
                       thresh=0.35, merge=0.4, max_area=9000),
             'c': dict(grid=(1, 1), overlap=0.25, size=(640, 480), topk=10, n_obj=6, n_cls=3, mask=(0, 1, 2),
                       thresh=0.5, merge=0.6, max_area=120000)}
    for tag, c in cases.items():
        self = SimpleNamespace(model=SimpleNamespace(INPUT_SHAPE=(3, 300, 300)), tiling_grid=c['grid'],
                               tile_overlap=c['overlap'], batch_size=int(np.prod(c['grid'])), merge_thresh=c['merge'],
                               _merge=ref._merge)
        tiles, region = ref._generate_tiles(self)
        scale = tuple(np.array(c['size']) / region)
        det_out = synthetic_engine_output(rng, tiles, c['topk'], c['n_obj'], c['n_cls'])
        label_mask = np.zeros(91, bool)
        label_mask[list(c['mask'])] = True
        dets, tile_ids = ref._filter_dets(det_out, tiles, c['topk'], label_mask, c['max_area'], c['thresh'], scale)
        flt = np.fromiter(dets, DET_DTYPE, len(dets))
        merged = ref._merge_dets(self, list(dets), list(tile_ids))
        out.update({f'{tag}_tiles': tiles, f'{tag}_region': np.array(region), f'{tag}_det_out': det_out,
                    f'{tag}_params': np.array([c['grid'][0], c['grid'][1], c['overlap'], c['size'][0], c['size'][1],
                                               c['topk'], c['thresh'], c['merge'], c['max_area']], float),
                    f'{tag}_mask': np.array(c['mask']),
                    f'{tag}_flt_tlbr': flt['tlbr'], f'{tag}_flt_label': flt['label'], f'{tag}_flt_conf': flt['conf'],
                    f'{tag}_flt_tile': np.array(tile_ids, int),
                    f'{tag}_mrg_tlbr': np.array(merged.tlbr).reshape(-1, 4), f'{tag}_mrg_label': np.array(merged.label),
                    f'{tag}_mrg_conf': np.array(merged.conf)})
        print(tag, 'tiles', tiles.shape, 'filtered', len(flt), 'merged', len(merged))
    # normalisation: a small region (tile 300 x 300, grid 2 x 1) -> full tensors are 2 x 3 x 300 x 300
    self = SimpleNamespace(model=SimpleNamespace(INPUT_SHAPE=(3, 300, 300)), tiling_grid=(2, 1), tile_overlap=0.25)
    tiles, region = ref._generate_tiles(self)
    frame = rng.integers(0, 256, (region[1], region[0], 3), dtype=np.uint8)
    inp = np.empty((2, 3, 300, 300), np.float32)
    ref._normalize(frame, tiles, inp)
    out.update(n_frame=frame, n_tiles=tiles, n_out_sum=inp.sum(axis=(2, 3)).astype(np.float64), n_out_corner=inp[:, :, :4, :4],
               n_out_last=inp[:, :, -1, -3:])
    np.savez_compressed(ROOT / 'tests' / 'golden' / 'ssd_kat.npz', **out)
    print('ssd_kat: ok')
