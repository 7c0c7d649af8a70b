"""GPU parity of the detector / extractor pre- and post-processing (detect.hip, extract.hip).
  * filter + DIoU-NMS: oracle=reference (tests/golden/nms_kat.npz from detector.py:322-365) --
    boxes, labels identical, conf equal
  * head decode: oracle=restated (np_oracle.yolo_decode, plugins/yolo_layer.cu:127-230), fp32
    tolerance 2e-6 relative (fast-exp intrinsics differ between vendors)
  * preprocessing: oracle=restated (np_oracle.yolo_preprocess / cv_oracle.reid_preprocess),
    identical uint8 pixels -> fp16 input within 1e-3"""
import numpy as np
import pytest

import cv_oracle
import np_oracle as o
from fastmot_amd import _lib
from fastmot_amd.detector import YOLODetector
from fastmot_amd.feature_extractor import FeatureExtractor
from fastmot_amd.models import YOLO, ReID
from fastmot_amd.models.graph import RandomWeights

pytestmark = pytest.mark.gpu


def synthetic_frame(w, h, seed=0):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (h // 8 + 1, w // 8 + 1, 3)).astype(np.uint8)
    frame = np.kron(base, np.ones((8, 8, 1), np.uint8))[:h, :w]
    noise = rng.integers(-20, 20, frame.shape)
    return np.clip(frame.astype(int) + noise, 0, 255).astype(np.uint8)


class TinyYOLO(YOLO):
    NUM_CLASSES = 3
    INPUT_SHAPE = (3, 96, 128)
    LAYER_FACTORS = [8, 16, 32]
    SCALES = [1.2, 1.1, 1.05]
    ANCHORS = [[4, 7, 8, 15, 12, 30], [18, 40, 25, 60, 30, 80], [40, 90, 60, 70, 80, 95]]


class TinyLetterbox(TinyYOLO):
    LETTERBOX = True
    NEW_COORDS = True
    SCALES = [2.0, 2.0, 2.0]


@pytest.fixture(params=['fused', 'general'])
def nms_path(ctx, request):
    """Both device paths of the candidate sort + DIoU-NMS: the fused greedy kernel (default, up to 4096 candidates) and
    the three-kernel sort / bit-matrix / scan path (fm_ctx option "nms_path" = 1; beyond 4096 candidates automatically)."""
    ctx.set_option('nms_path', 1 if request.param == 'general' else 0)
    yield request.param
    ctx.set_option('nms_path', 0)


@pytest.mark.parametrize('tag,n_cls', [('p', 1), ('q', 3)])
def test_filter_dets_matches_reference(ctx, golden_dir, tag, n_cls, nms_path):
    g = np.load(golden_dir / 'nms_kat.npz')
    det = YOLODetector((1920, 1080), tuple(range(n_cls)), model='TinyYOLO' if n_cls <= 3 else None,
                       conf_thresh=0.25, nms_thresh=0.5, max_area=800000, min_aspect_ratio=1.2)
    out = ctx.filter_dets(g[f'{tag}_det_out'])
    np.testing.assert_array_equal(out.tlbr, g[f'{tag}_tlbr'])
    np.testing.assert_array_equal(out.label, g[f'{tag}_label'])
    np.testing.assert_allclose(out.conf, g[f'{tag}_conf'], rtol=1e-7)


def test_filter_dets_random_vs_oracle(ctx, nms_path):
    rng = np.random.default_rng(5)
    det = YOLODetector((1920, 1080), (0, 2), model='TinyYOLO', conf_thresh=0.3, nms_thresh=0.45,
                       max_area=200000, min_aspect_ratio=0.5)
    for n in (0, 1, 7, 64, 65, 600, 3000):
        rows = np.stack([rng.uniform(0, 0.9, n), rng.uniform(0, 0.8, n), rng.uniform(0.01, 0.15, n),
                         rng.uniform(0.02, 0.3, n), rng.uniform(0, 1, n), rng.integers(0, 3, n),
                         rng.uniform(0.3, 1, n)], 1).astype(np.float32)
        out = ctx.filter_dets(rows)
        lm = np.array([True, False, True])
        tl, lb, cf = o.filter_dets(rows, [1920, 1080], [0, 0], lm, 0.3, 0.45, 200000, 0.5)
        np.testing.assert_array_equal(out.tlbr, tl)
        np.testing.assert_array_equal(out.label, lb)
        np.testing.assert_allclose(out.conf, cf, rtol=1e-7)


@pytest.mark.parametrize('n', [63, 64, 65, 127, 128, 129, 191, 192, 193, 256, 2559, 2560, 2561, 2688, 2689, 2753])
def test_filter_dets_chunk_boundaries(ctx, n):
    """Candidate counts at the seams of the bit-matrix / scan path: 64-candidate chunks, two chunks per scan iteration,
    an odd number of chunks, and the 2560 rows whose loads the scan pipelines (beyond them rows are fetched at use).
    Every row passes the threshold, so K = n exactly; boxes cluster so that chains of suppression cross the chunks."""
    rng = np.random.default_rng(n)
    det = YOLODetector((1920, 1080), (0, 1, 2), model='TinyYOLO', conf_thresh=0.3, nms_thresh=0.45,
                       max_area=200000, min_aspect_ratio=0.5, max_candidates=16384)
    ctx.set_option('nms_path', 1)
    try:
        centres = rng.uniform(0.1, 0.8, (max(n // 12, 1), 2))
        pick = rng.integers(0, len(centres), n)
        rows = np.stack([centres[pick, 0] + rng.normal(0, 0.01, n), centres[pick, 1] + rng.normal(0, 0.01, n),
                         rng.uniform(0.05, 0.08, n), rng.uniform(0.1, 0.16, n), rng.uniform(0.6, 1, n),
                         rng.integers(0, 3, n), rng.uniform(0.6, 1, n)], 1).astype(np.float32)
        out = ctx.filter_dets(rows, cap=16384)
        assert ctx.detect_last_counts()[0] == n
        lm = np.array([True, True, True])
        tl, lb, cf = o.filter_dets(rows, [1920, 1080], [0, 0], lm, 0.3, 0.45, 200000, 0.5)
        assert 0 < len(tl) < n
        np.testing.assert_array_equal(out.tlbr, tl)
        np.testing.assert_array_equal(out.label, lb)
        np.testing.assert_allclose(out.conf, cf, rtol=1e-7)
    finally:
        ctx.set_option('nms_path', 0)


def test_filter_dets_more_candidates_than_the_fused_kernel_holds(ctx):
    """> 4096 candidates over the threshold: the fused kernel flags the pass, the general path takes it over (and the
    following passes, until the count has fallen again); results equal the oracle throughout."""
    rng = np.random.default_rng(9)
    det = YOLODetector((1920, 1080), (0, 1, 2), model='TinyYOLO', conf_thresh=0.3, nms_thresh=0.45,
                       max_area=200000, min_aspect_ratio=0.5, max_candidates=16384)
    lm = np.array([True, True, True])
    for n in (5000, 9000, 2000, 300, 5000):
        rows = np.stack([rng.uniform(0, 0.9, n), rng.uniform(0, 0.8, n), rng.uniform(0.01, 0.15, n),
                         rng.uniform(0.02, 0.3, n), rng.uniform(0, 1, n), rng.integers(0, 3, n),
                         rng.uniform(0.4, 1, n)], 1).astype(np.float32)
        out = ctx.filter_dets(rows, cap=16384)
        tl, lb, cf = o.filter_dets(rows, [1920, 1080], [0, 0], lm, 0.3, 0.45, 200000, 0.5)
        assert ctx.detect_last_counts()[0] == int(((rows[:, 4] * rows[:, 6]) >= np.float32(0.3)).sum())
        np.testing.assert_array_equal(out.tlbr, tl)
        np.testing.assert_array_equal(out.label, lb)
        np.testing.assert_allclose(out.conf, cf, rtol=1e-7)


@pytest.mark.parametrize('model', ['TinyYOLO', 'TinyLetterbox'])
def test_preprocess_decode_end_to_end(ctx, model):
    size = (320, 180)
    det = YOLODetector(size, (0, 1, 2), model=model, conf_thresh=0.1, nms_thresh=0.5,
                       weights=RandomWeights(seed=4), max_candidates=16384, reuse_buffers=False)
    frame = synthetic_frame(*size, seed=1)
    ctx.frame_configure(*size)
    ctx.frame_upload(frame)
    ctx.detect_preprocess_only()
    m = det.model
    inp = det.backend.read(det.graph.input, 1)[0]            # [h, w, 3] RGB
    exp = o.yolo_preprocess(frame, m.INPUT_SHAPE[1:], det.roi if m.LETTERBOX else None)
    np.testing.assert_allclose(inp.transpose(2, 0, 1), exp, rtol=0, atol=6e-4)   # fp16 storage of u8/255
    # the u8 pixels themselves must be identical
    np.testing.assert_array_equal(np.rint(inp.transpose(2, 0, 1) * 255), np.rint(exp * 255))
    dets = det(frame)
    # oracle: decode the engine's own head tensors, then the reference filter
    rows = []
    for i, head in enumerate(det.heads):
        t = det.backend.read(head, 1)[0]                       # [gh, gw, (5+C)*A] fp32
        rows.append(o.yolo_decode(t.transpose(2, 0, 1), m.ANCHORS[i], m.NUM_CLASSES,
                                  (m.INPUT_SHAPE[2], m.INPUT_SHAPE[1]), m.SCALES[i], m.NEW_COORDS))
    rows = np.concatenate(rows)
    tl, lb, cf = o.filter_dets(rows, det.upscaled_sz, det.bbox_offset, det.label_mask, 0.1, 0.5, 800000, 1.2)
    assert len(dets) == len(tl) and len(tl) > 0
    # The output order is by confidence; detections whose confidences agree to ~1e-4 may swap places between
    # the fast-exp device decode and the oracle (seen: 0.2328964 vs 0.2328774), so rows are matched first:
    # every oracle row pairs with the nearest unused device row of the same label.
    order, used = [], np.zeros(len(tl), bool)
    for j in range(len(tl)):
        d = np.abs(dets.tlbr - tl[j]).max(1) + 1e6 * ((dets.label != lb[j]) | used)
        order.append(int(np.argmin(d)))
        used[order[-1]] = True
    order = np.array(order)
    assert (np.abs(order - np.arange(len(tl))) <= 2).all()        # only neighbours swap
    np.testing.assert_array_equal(dets.label[order], lb)
    # boxes come from fp32 fast-exp decode: allow +-1 px on <=1% of coordinates, confidences 5e-6
    diff = np.abs(dets.tlbr[order] - tl)
    assert diff.max() <= 1 and (diff > 0).mean() <= 0.01
    np.testing.assert_allclose(dets.conf[order], cf, rtol=5e-6)


def test_stale_prefetch_is_dropped(ctx):
    """prefetch(f_x) followed by detect_async(f_1) with another frame: postprocess() returns f_1's detections
    (detector passes are collected in the order they were enqueued, so the announced-but-unused pass is
    collected and dropped first); the announced frame itself is recognised by identity and not run twice."""
    size = (320, 180)
    det = YOLODetector(size, (0, 1, 2), model='TinyYOLO', conf_thresh=0.1, nms_thresh=0.5,
                       weights=RandomWeights(seed=4), max_candidates=16384, reuse_buffers=False)
    f0, f1, fx = (synthetic_frame(*size, seed=s) for s in (1, 2, 3))
    want0, want1 = det(f0).copy(), det(f1).copy()
    assert len(want0) and len(want1) and (len(want0) != len(want1) or (want0.tlbr != want1.tlbr).any())
    det.detect_async(f0)
    det.prefetch(fx)
    got0 = det.postprocess().copy()
    det.detect_async(f1)                      # not the announced frame
    got1 = det.postprocess().copy()
    det.detect_async(f0)
    det.prefetch(f1)
    det.postprocess()
    det.detect_async(f1)                      # the announced frame: no second pass
    got1b = det.postprocess().copy()
    for got, want in ((got0, want0), (got1, want1), (got1b, want1)):
        assert len(got) == len(want)
        np.testing.assert_array_equal(got.tlbr, want.tlbr)
        np.testing.assert_array_equal(got.label, want.label)


def test_reid_crop_resize_normalise(ctx):
    size = (640, 360)
    frame = synthetic_frame(*size, seed=2)
    ext = FeatureExtractor('OSNet025', batch_size=8, weights=RandomWeights(seed=3), size=size)
    boxes = np.array([[10.7, 20.2, 70.9, 200.1], [-5., -8., 40., 90.], [600., 300., 700., 400.],
                      [100., 50., 131., 113.], [300., 10., 555., 355.]])
    ext.extract_async(frame, boxes)
    emb = ext.postprocess()
    assert emb.shape == (5, 512)
    np.testing.assert_allclose(np.linalg.norm(emb, axis=1), 1, atol=1e-5)
    inp = ctx.extract_read_input(5, 128, 256)
    exp = cv_oracle.reid_preprocess(frame, boxes).transpose(0, 2, 3, 1)
    np.testing.assert_allclose(inp, exp, rtol=0, atol=2.5e-3)     # fp16 storage of values in [-2.2, 2.7]
    # chunked extraction (n > batch_size) gives the same embeddings row by row
    many = np.concatenate([boxes] * 4)
    ext.extract_async(frame, many)
    emb2 = ext.postprocess()
    np.testing.assert_allclose(emb2[:5], emb, atol=1e-6)
    np.testing.assert_allclose(emb2[15:20], emb, atol=1e-6)


def test_reid_crop_exact_2x_uses_area_average(ctx):
    """cv2.resize(INTER_LINEAR) switches to the INTER_AREA 2x2 mean when the crop is exactly twice the network
    input in both axes (256x512 -> 128x256); crops one pixel off stay bilinear."""
    size = (800, 600)
    frame = synthetic_frame(*size, seed=6)
    ext = FeatureExtractor('OSNet025', batch_size=8, weights=RandomWeights(seed=3), size=size)
    boxes = np.array([[100., 20., 355., 531.], [101., 20., 355., 531.], [300., 60., 555., 571.9]])
    ext.extract_async(frame, boxes)
    ext.postprocess()
    inp = ctx.extract_read_input(3, 128, 256)
    exp = cv_oracle.reid_preprocess(frame, boxes).transpose(0, 2, 3, 1)
    np.testing.assert_allclose(inp, exp, rtol=0, atol=2.5e-3)
    crop = frame[20:532, 100:356].astype(np.int64)
    area = (crop[0::2, 0::2] + crop[0::2, 1::2] + crop[1::2, 0::2] + crop[1::2, 1::2] + 2) >> 2
    mean, std = np.array([0.485, 0.456, 0.406]), np.array([0.229, 0.224, 0.225])
    np.testing.assert_allclose(inp[0], (area[..., ::-1] / 255. - mean) / std, rtol=0, atol=2.5e-3)


def test_reid_split_batches_identical(ctx):
    """Two or four network instances running the parts of a batch concurrently (FM_NET_EXTRACTOR_B.., extract.hip) give
    the same embedding rows, bit for bit, as one instance running the whole batch -- for even, odd and
    below-threshold box counts."""
    size = (640, 360)
    frame = synthetic_frame(*size, seed=4)
    rng = np.random.default_rng(5)
    x0, y0 = rng.uniform(0, 500, 33), rng.uniform(0, 250, 33)
    boxes = np.stack([x0, y0, x0 + rng.uniform(20, 120, 33), y0 + rng.uniform(40, 100, 33)], 1)
    embs = {}
    for split in (4, 2, 1):
        ext = FeatureExtractor('OSNet025', batch_size=32, weights=RandomWeights(seed=3), size=size, split_batches=split)
        assert len(ext.extra_backends) == split - 1
        for n in (33, 32, 9, 7, 1):
            ext.extract_async(frame, boxes[:n])
            embs[split, n] = ext.postprocess().copy()
            assert embs[split, n].shape == (n, 512)
    for n in (33, 32, 9, 7, 1):
        np.testing.assert_array_equal(embs[2, n], embs[1, n])
        np.testing.assert_array_equal(embs[4, n], embs[1, n])


@pytest.mark.parametrize('model', ['TinyYOLO', 'TinyLetterbox', 'YOLOv4_608'])
def test_fused_input_stem_equals_preprocess_kernel(ctx, model):
    """Round 6: the detector's stem convolution computes the resized / normalised input pixels itself (fm_ctx option
    "fused_input", default 1; stemconv.hip + pixel_source.h) instead of reading the tensor preprocess_kernel wrote.
    Same pixel function on both paths: every head tensor and the detections are equal bit for bit."""
    size = (1920, 1080) if model == 'YOLOv4_608' else (320, 180)
    det = YOLODetector(size, (0, 1, 2), model=model, conf_thresh=0.1, nms_thresh=0.5, weights=RandomWeights(seed=4),
                       max_candidates=65536, reuse_buffers=False)
    frame = synthetic_frame(*size, seed=11)
    outs = {}
    try:
        for fused in (1, 0, 1):
            ctx.set_option('fused_input', fused)
            dets = det(frame)
            heads = [det.backend.read(h, 1).copy() for h in det.heads]
            stem = det.backend.read(det.graph.layers[0]['out'], 1).copy()
            outs.setdefault(fused, []).append((dets, heads, stem))
    finally:
        ctx.set_option('fused_input', 1)
    assert det.graph.layers[0]['op'] in (12, 18)                # FM_OP_STEMCONV / FM_OP_STEM2: the fused path really is in use
    for dets, heads, stem in outs[1][1:] + outs[0]:
        ref = outs[1][0]
        np.testing.assert_array_equal(stem, ref[2])
        for a, b in zip(heads, ref[1]):
            np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(dets.tlbr, ref[0].tlbr)
        np.testing.assert_array_equal(dets.conf, ref[0].conf)


@pytest.mark.parametrize('model,n', [('OSNet025', 5), ('OSNet025', 19), ('OSNet10', 5)])
def test_fused_input_stem_equals_crop_kernel(ctx, model, n):
    """The same for the ReID network: crop -> resize -> normalise inside OSNet's 7x7 stem (chunked batches included:
    n > batch_size) against crop_resize_kernel + input tensor -- embeddings equal bit for bit."""
    size = (640, 360)
    frame = synthetic_frame(*size, seed=2)
    ext = FeatureExtractor(model, batch_size=8, weights=RandomWeights(seed=3), size=size)
    rng = np.random.default_rng(n)
    tl = rng.uniform([-20, -20], [560, 250], (n, 2))
    boxes = np.concatenate([tl, tl + rng.uniform([8, 16], [140, 300], (n, 2))], axis=1)
    embs = {}
    try:
        for fused in (1, 0):
            ctx.set_option('fused_input', fused)
            ext.extract_async(frame, boxes)
            embs[fused] = ext.postprocess().copy()
    finally:
        ctx.set_option('fused_input', 1)
    # (OSNet-x1.0's first conv has 64 output channels: not a stem-kernel layer, both settings take the crop kernel)
    assert (ext.graph.layers[0]['op'] == 12) == (model == 'OSNet025')
    np.testing.assert_array_equal(embs[1], embs[0])
    if n <= 8:
        inp = ctx.extract_read_input(n, 128, 256)            # (refilled by the front-end kernel on the fused path)
        exp = cv_oracle.reid_preprocess(frame, boxes).transpose(0, 2, 3, 1)
        np.testing.assert_allclose(inp, exp, rtol=0, atol=2.5e-3)


def test_embedding_buffer_growth_keeps_captured_graphs_valid(ctx):
    """The ReID head layer writes its rows through pointers baked into captured graphs.  A batch larger than the
    embedding buffer re-allocates it (device rows + their page-locked mirror): the graphs captured for earlier, smaller
    batches must not survive that (until round 6 they did, and replayed into the freed buffer)."""
    size = (640, 360)
    frame = synthetic_frame(*size, seed=8)
    ext = FeatureExtractor('OSNet025', batch_size=8, weights=RandomWeights(seed=3), size=size)
    rng = np.random.default_rng(1)

    def boxes(n):
        tl = rng.uniform([0, 0], [500, 200], (n, 2))
        return np.concatenate([tl, tl + rng.uniform([10, 20], [120, 150], (n, 2))], axis=1)
    small = boxes(5)
    first = [ext(frame, small).copy() for _ in range(3)]            # eager validation, capture, replay
    many = boxes(150)                                               # > 64 rows (and > 128): the buffers grow
    big = ext(frame, many).copy()
    again = ext(frame, small).copy()
    np.testing.assert_array_equal(first[0], first[2])
    np.testing.assert_array_equal(again, first[0])
    np.testing.assert_array_equal(ext(frame, many), big)
    np.testing.assert_allclose(np.linalg.norm(big, axis=1), 1, atol=1e-5)
