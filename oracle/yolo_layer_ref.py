"""TEST INFRASTRUCTURE ONLY -- build container only (/root/reference).

Recipe for oracle/_ref/libyolo_layer_ref.so: the reference's YOLO decode kernels (fastmot/plugins/yolo_layer.cu:115-230,
CalDetection / CalDetection_NewCoords) compiled for the HOST from the reference file where it lies -- the device
functions are cut out by line pattern into oracle/_ref/ (a git-ignored build product) and compiled behind the shims of
oracle/yolo_layer_ref.cpp.  The rest of that file is the TensorRT plugin class and needs NvInfer.h, which does not
exist here; the reference's own Makefile (nvcc + TensorRT) is not run.

    python oracle/yolo_layer_ref.py [--golden]    # build + self-check against np_oracle.yolo_decode [+ the travelling fixture]

decode(head, ...) has np_oracle.yolo_decode's signature; tests/test_oracle_vs_reference.py compares the two."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
REF = Path('/root/reference/fastmot/plugins/yolo_layer.cu')
OUT = HERE / '_ref'
LIB = OUT / 'libyolo_layer_ref.so'
BEGIN, END = 'inline __device__ float sigmoidGPU', 'void YoloLayerPlugin::forwardGpu'
_lib = None


def available():
    return REF.exists()


def build(force=False):
    if not available():
        raise RuntimeError(f'{REF} is not present (GPU box?)')
    src = HERE / 'yolo_layer_ref.cpp'
    if not force and LIB.exists() and LIB.stat().st_mtime > max(src.stat().st_mtime, REF.stat().st_mtime):
        return LIB
    OUT.mkdir(exist_ok=True)
    lines = REF.read_text().splitlines()
    b = next(i for i, l in enumerate(lines) if BEGIN in l)
    e = next(i for i, l in enumerate(lines) if END in l)
    (OUT / 'yolo_layer_kernels.inc').write_text('\n'.join(lines[b:e]) + '\n')
    res = subprocess.run(['g++', '-O2', '-ffp-contract=off', '-shared', '-fPIC', '-I', str(HERE), '-o', str(LIB), str(src)],
                         capture_output=True, text=True)
    (OUT / 'yolo_layer_kernels.inc').unlink()        # reference text: needed by the compiler only, never kept
    if res.returncode != 0:
        raise RuntimeError('g++ failed:\n' + res.stderr)
    return LIB


def decode(head, anchors, num_classes, in_wh, scale_xy, new_coords=False):
    """head: float32 [(5+C)*A, H, W] -> [A*H*W, 7] rows (x, y, w, h, box_conf, class_id, class_prob), as the plugin
    writes them."""
    global _lib
    if _lib is None:
        _lib = C.CDLL(str(build()))
    head = np.ascontiguousarray(head, np.float32)
    A = len(anchors) // 2
    _, H, W = head.shape
    out = np.zeros((A * H * W, 7), np.float32)
    anc = np.ascontiguousarray(anchors, np.float32)
    _lib.ref_yolo_decode(C.c_void_p(head.ctypes.data), C.c_void_p(out.ctypes.data), C.c_int(W), C.c_int(H), C.c_int(A),
                         C.c_void_p(anc.ctypes.data), C.c_int(num_classes), C.c_int(in_wh[0]), C.c_int(in_wh[1]),
                         C.c_float(scale_xy), C.c_int(int(new_coords)))
    return out


def make_golden():
    """tests/golden/yolo_decode_ref.npz: seeded heads and what the reference kernels make of them (the fixture travels;
    tests/test_oracle_golden.py holds np_oracle.yolo_decode against it on every machine)"""
    rng = np.random.default_rng(41)
    out = {}
    cases = [(80, False, (5, 4), (128, 160)), (2, False, (19, 19), (608, 608)), (1, False, (3, 7), (224, 96)),
             (3, True, (10, 18), (576, 320)), (80, True, (4, 4), (128, 128))]
    for k, (nc, new, (H, W), in_wh) in enumerate(cases):
        head = (rng.uniform(0, 1, ((5 + nc) * 3, H, W)) if new else rng.normal(0, 2.5, ((5 + nc) * 3, H, W))).astype(np.float32)
        if nc > 1:
            head[6, 0, 0] = head[5, 0, 0]
        anchors = rng.integers(8, 200, 6).astype(np.float32)
        out[f'k{k}_head'], out[f'k{k}_anchors'] = head, anchors
        out[f'k{k}_params'] = np.array([nc, int(new), in_wh[0], in_wh[1], 1.05 if k % 2 else 1.2])
        out[f'k{k}_rows'] = decode(head, anchors, nc, in_wh, float(out[f'k{k}_params'][4]), new)
    out['n'] = np.array(len(cases))
    np.savez_compressed(HERE.parent / 'tests' / 'golden' / 'yolo_decode_ref.npz', **out)
    print('yolo_decode_ref.npz:', {k: v.shape for k, v in out.items() if k.endswith('rows')})


if __name__ == '__main__':
    import sys
    sys.path.insert(0, str(HERE))
    import np_oracle as o
    print('built', build(force=True))
    if '--golden' in sys.argv:
        make_golden()
    rng = np.random.default_rng(0)
    for nc, new in ((80, False), (2, True), (1, False)):
        head = rng.normal(0, 2, ((5 + nc) * 3, 19, 13)).astype(np.float32)
        if new:
            head = rng.uniform(0, 1, head.shape).astype(np.float32)
        anchors = [12, 16, 19, 36, 40, 28]
        a, b = decode(head, anchors, nc, (416, 608), 1.05, new), o.yolo_decode(head, anchors, nc, (416, 608), 1.05, new)
        print(nc, new, 'class ids equal', np.array_equal(a[:, 5], b[:, 5]), 'max rel diff',
              float(np.max(np.abs(a - b) / (np.abs(b) + 1e-6))))
