"""Parity AT THE BENCHMARK'S SIZES (the kernel instances the profile is dominated by): YOLOv4 @608x608 / 80
classes -- every tensor of the layer table, not only the heads --, OSNet-x0.25 / x1.0 @256x128 at the batch
sizes of BASELINE config[1] / [2], and Flow.predict at 1920x1080 with 50 tracks + background points.

Conv oracle: tests/torch_ref.py (PyTorch fp32 on the CPU, the same layer table and weights, activations rounded
to fp16 at the same points as the engine stores them).  Tolerances (stated in DESIGN.md section 7):
  * any conv-network tensor:   max|gpu - ref| <= 6e-3 * max|ref| + 2e-3   and   rms(gpu - ref) <= 1.5e-3 * rms(ref)
    (a few fp16 ulps of the tensor's range: the two sides differ only by fp32 summation order and then by fp16
    rounding flips that propagate; a wrong filter tap on one border row is ~1e-1 * max|ref| on that row);
  * embeddings: |gpu - ref| <= 4e-3 per component, cosine similarity >= 0.99999.
KLT oracle: oracle/cv_oracle.py (OpenCV algorithms restated): keypoints (GFTT, FAST and the LK-tracked points),
status / inlier flags, per-track result codes and rounded boxes IDENTICAL; homography rtol 1e-6 (double precision
Jacobi / Levenberg-Marquardt on the host vs numpy)."""
import numpy as np
import pytest
import torch

import cv_oracle as cv
import cpu_tracker
import scenes
import torch_ref
from fastmot_amd.engine import HipNet, NET_DETECTOR, NET_EXTRACTOR
from fastmot_amd.models import YOLO, ReID
from fastmot_amd.models.graph import RandomWeights, View

pytestmark = pytest.mark.gpu

REL, ABS, RMS = 6e-3, 2e-3, 1.5e-3


def nchw(a):
    return torch.from_numpy(np.ascontiguousarray(a.transpose(0, 3, 1, 2)))


def check_tensor(gpu, ref, what):
    ref = np.asarray(ref, np.float32)
    err = np.abs(gpu - ref)
    lim = REL * np.abs(ref).max() + ABS
    assert err.max() <= lim, f'{what}: max err {err.max():.4g} > {lim:.4g} (ref max {np.abs(ref).max():.4g}) at ' \
                             f'{np.unravel_index(err.argmax(), err.shape)}'
    rms_ref = float(np.sqrt(np.mean(ref.astype(np.float64) ** 2)))
    rms_err = float(np.sqrt(np.mean(err.astype(np.float64) ** 2)))
    assert rms_err <= RMS * rms_ref + 1e-5, f'{what}: rms err {rms_err:.4g} vs rms {rms_ref:.4g}'
    return err.max() / max(np.abs(ref).max(), 1e-12), rms_err / max(rms_ref, 1e-12)


def test_yolov4_608_every_tensor(ctx):
    """BASELINE config[1] detector: all 3 heads AND every intermediate tensor of the 608x608 / 80-class graph
    (fused residual units, streamed 19x19 / 38x38 convs, stem, SPP, upsample-in-epilogue, in-place concats)."""
    torch.set_num_threads(max(torch.get_num_threads(), 8))
    model = YOLO.get_model('YOLOv4_608')
    g, heads = model.build_graph(RandomWeights(seed=31))
    net = HipNet(ctx, NET_DETECTOR, g, 1, reuse_buffers=False)
    rng = np.random.default_rng(32)
    x = rng.uniform(0, 1, (1, 608, 608, 3)).astype(np.float16)
    net.write(g.input, x)
    net.run(1)
    bufs, _ = torch_ref.run_graph(g, nchw(x.astype(np.float32)))
    worst = (0, None)
    written = {d['out'].tid for d in g.layers}
    for tid in sorted(written):
        h, w, c, f32 = g.tensors[tid]
        full = net.read(_whole(g, tid), 1)
        ref = bufs[tid].numpy().transpose(0, 2, 3, 1)
        rel, rms = check_tensor(full, ref, f'tensor {tid} ({h}x{w}x{c})')
        worst = max(worst, (rel, tid))
    for i, hd in enumerate(heads):
        check_tensor(net.read(hd, 1), bufs[hd.tid][:, hd.coff:hd.coff + hd.c].numpy().transpose(0, 2, 3, 1), f'head {i}')
    print(f'YOLOv4@608: {len(written)} tensors, worst max-err/max {worst[0]:.2e} (tensor {worst[1]})')
    net.close()


def _whole(g, tid):
    """View of a whole tensor (all stored channels)."""
    h, w, c, _ = g.tensors[tid]
    return View(tid, 0, c, h, w)


@pytest.mark.parametrize('model,batch', [('OSNet025', 50), ('OSNet10', 16)])
def test_osnet_256x128_embeddings_and_tensors(ctx, model, batch):
    """OSNet at its real input size: x0.25 at the 50-crop batch of config[1], x1.0 (config[2], 12x the FLOPs,
    per-depth grouped LightConv launches where the chain kernel's LDS budget is exceeded) at batch 16."""
    torch.set_num_threads(max(torch.get_num_threads(), 8))
    cls = ReID.get_model(model)
    g, _ = cls.build_graph(RandomWeights(seed=41))
    ctx.feat_configure(512)
    net = HipNet(ctx, NET_EXTRACTOR, g, batch, reuse_buffers=False)
    rng = np.random.default_rng(42)
    x = rng.normal(0, 1, (batch, 256, 128, 3)).astype(np.float16)
    net.write(g.input, x)
    net.run(batch)
    emb = net.read_embeddings(batch)
    bufs, ref = torch_ref.run_graph(g, nchw(x.astype(np.float32)))
    ref = ref.numpy()
    np.testing.assert_allclose(np.linalg.norm(emb, axis=1), 1.0, atol=1e-5)
    for tid in sorted({d['out'].tid for d in g.layers if d.get('out') is not None}):
        h, w, c, _ = g.tensors[tid]
        check_tensor(net.read(_whole(g, tid), batch), bufs[tid].numpy().transpose(0, 2, 3, 1), f'{model} tensor {tid} ({h}x{w}x{c})')
    assert np.abs(emb - ref).max() <= 4e-3, np.abs(emb - ref).max()
    assert (np.sum(emb * ref, axis=1) >= 0.99999).all(), np.sum(emb * ref, axis=1).min()
    net.close()


def test_feature_extractor_batch_64_two_instances(ctx):
    """FeatureExtractor at batch 64 on a 1080p frame: the batch runs as two concurrent 32-crop network instances;
    every embedding equals the PyTorch reference run on the crops the device produced."""
    from fastmot_amd.feature_extractor import FeatureExtractor
    from synthetic import SyntheticVideo
    size = (1920, 1080)
    video = SyntheticVideo(size, n_ids=64, n_frames=1, seed=7)
    boxes = video.detections(0).tlbr
    # the crops as the network sees them: read back from a single-instance extractor (with two instances the
    # second half of the batch lives in the second instance's input tensor)
    one = FeatureExtractor('OSNet025', batch_size=64, weights=RandomWeights(seed=43), size=size, split_batches=1,
                           reuse_buffers=False)
    emb_one = one(video.frames[0], boxes)
    inp = ctx.extract_read_input(64, 128, 256)                       # [n, h, w, 3]
    exp = cv.reid_preprocess(video.frames[0], boxes).transpose(0, 2, 3, 1)
    np.testing.assert_allclose(inp, exp, rtol=0, atol=2.5e-3)
    ext = FeatureExtractor('OSNet025', batch_size=64, weights=RandomWeights(seed=43), size=size, split_batches=2,
                           reuse_buffers=False)
    assert len(ext.extra_backends) == 1
    emb = ext(video.frames[0], boxes)
    np.testing.assert_array_equal(emb, emb_one)                      # two 32-crop instances == one 64-crop instance
    _, ref = torch_ref.run_graph(ext.graph, nchw(inp.astype(np.float32)))
    ref = ref.numpy()
    assert emb.shape == (64, 512)
    assert np.abs(emb - ref).max() <= 4e-3, np.abs(emb - ref).max()
    assert (np.sum(emb * ref, axis=1) >= 0.99999).all()


class _Trk:
    def __init__(self, trk_id, tlbr):
        self.trk_id, self.age = trk_id, 0
        self._tlbr = np.asarray(tlbr, float)
        self.keypoints = np.empty((0, 2), np.float32)
        self.prev_keypoints = np.empty((0, 2), np.float32)
        self.inlier_ratio = 1.

    tlbr = property(lambda self: self._tlbr)

    def __lt__(self, other):
        return (self.tlbr[-1], -self.age) < (other.tlbr[-1], -other.age)


def test_flow_predict_1080p_50_tracks(ctx):
    """Flow.predict (flow.py:135-264) at the benchmark's size in ONE call per frame: 50 tracks (GFTT keypoints
    under the closest-first foreground mask), FAST background points, pyramidal LK on ~5-7 k points, RANSAC.
    Three consecutive frames: the second and third call reuse propagated keypoints."""
    from fastmot_amd.flow import Flow
    from synthetic import SyntheticVideo
    size = (1920, 1080)
    video = SyntheticVideo(size, n_ids=50, n_frames=4, seed=100)
    flow = Flow(size, **vars(scenes.tracker_kwargs()['flow_cfg']))
    ora = cpu_tracker.OracleFlow(size)
    flow.init(video.frames[0])
    ora.init(video.frames[0])
    boxes0 = video.detections(0).tlbr
    a = [_Trk(i + 1, boxes0[i]) for i in range(50)]
    b = [_Trk(i + 1, boxes0[i]) for i in range(50)]
    for f in (1, 2, 3):
        ga, Ha = flow.predict(video.frames[f], a)
        gb, Hb = ora.predict(video.frames[f], b)
        assert Ha is not None and Hb is not None
        assert [t.trk_id for t in a] == [t.trk_id for t in b]                 # same closest-first order
        assert list(ga.keys()) == list(gb.keys()) and len(ga) >= 45
        n_pts = 0
        for ta, tb in zip(a, b):
            assert len(ta.keypoints) == len(tb.keypoints), (f, ta.trk_id)
            np.testing.assert_array_equal(ta.prev_keypoints, tb.prev_keypoints)   # GFTT / propagated points
            np.testing.assert_array_equal(ta.keypoints, tb.keypoints)             # LK: bit-identical
            assert ta.inlier_ratio == tb.inlier_ratio
            n_pts += len(ta.keypoints)
        np.testing.assert_array_equal(flow.prev_bg_keypoints, ora.prev_bg_keypoints)
        np.testing.assert_array_equal(flow.bg_keypoints, ora.bg_keypoints)
        for k in ga:
            np.testing.assert_array_equal(ga[k], gb[k])                           # rounded boxes
        np.testing.assert_allclose(Ha, Hb, rtol=1e-6, atol=1e-8)                  # host double arithmetic (Jacobi / LM)
        assert n_pts > 2000 and len(flow.bg_keypoints) > 100
        for ta, tb in zip(a, b):                                              # both sides continue from the same boxes
            if ta.trk_id in gb:
                ta._tlbr = tb._tlbr = np.rint(gb[ta.trk_id])
