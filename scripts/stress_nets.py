"""Are the NETWORKS reproducible bit for bit while another stream keeps the CUs busy?  Victims: YOLOv4@608 head
tensors, OSNet-x0.25 embeddings (50 crops), the batched Kalman step, the pairwise cost kernel.  Hammers: grouped
LightConv launches (the kernel that disturbs the LK kernel, scripts/stress_lk5.py), the OSNet graph, YOLOv4."""
import os as _os
_os.environ.setdefault('FASTMOT_RANDOM_WEIGHTS', '1')
import sys, threading
sys.path.insert(0, '.')
import numpy as np
from fastmot_amd import _lib
from fastmot_amd.runtime import get_context
from fastmot_amd.engine import HipNet, NET_DETECTOR, NET_EXTRACTOR, NET_EXTRACTOR_B
from fastmot_amd.models import YOLO, ReID
from fastmot_amd.models.graph import Graph, RandomWeights

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
ctx = get_context()
ctx.feat_configure(512)
rng = np.random.default_rng(0)


def hammer_net(kind):
    if kind == 'liteconv':
        g = Graph(RandomWeights(seed=1), (64, 32), 16)
        params = [g.lightconv_params(f'p{i}', 16) for i in range(4)]
        g.lightconv_group('l', [g.input] * 4, params)
        return HipNet(ctx, NET_EXTRACTOR_B + 1, g, 50, reuse_buffers=True), 50
    if kind == 'osnet':
        g, _ = ReID.get_model('OSNet025').build_graph()
        return HipNet(ctx, NET_EXTRACTOR_B + 1, g, 50, reuse_buffers=True), 50
    g, _ = YOLO.get_model('YOLOv4_608').build_graph()
    return HipNet(ctx, NET_EXTRACTOR_B + 1, g, 1, reuse_buffers=True), 1


# victims
gy, heads = YOLO.get_model('YOLOv4_608').build_graph()
ynet = HipNet(ctx, NET_DETECTOR, gy, 1, reuse_buffers=True)
ynet.write(gy.input, rng.uniform(0, 1, (1, 608, 608, 3)).astype(np.float16))
go, _ = ReID.get_model('OSNet025').build_graph()
onet = HipNet(ctx, NET_EXTRACTOR, go, 50, reuse_buffers=True)
onet.write(go.input, rng.normal(0, 1, (50, 256, 128, 3)).astype(np.float16))


def run_yolo():
    ynet.run(1)
    return np.concatenate([ynet.read(h, 1).ravel() for h in heads])


# (the victim's last feature map, not its embeddings: every OSNet graph of a context writes its head's output to the
# context's ONE embedding bank at the row offset the extractor gives it -- a hammer OSNet launched by hand writes rows 0..49
# like the victim, and round 5 first read "9 of 300 embeddings differ" off that shared buffer)
_pre_head = go.layers[-1]['ins'][0]


def run_osnet():
    onet.run(50)
    return onet.read(_pre_head, 50).astype(np.float32).ravel()


n = 50
ctx.kf_configure(1 / 30., 2.25, 78.5, (0.08, 0.08), (0.14, 0.14), (4., 4.), (5., 5.), 5, 12, 0.6, 2)
ctx.set_frame_rect(np.array([0., 0., 1919., 1079.]))
tl = rng.uniform(0, 900, (n, 2)); boxes = np.concatenate([tl, tl + rng.uniform(30, 120, (n, 2))], 1)
slots = np.arange(1, n + 1, dtype=np.int32)
emb = rng.normal(0, 1, (n, 512)).astype(np.float32); emb /= np.linalg.norm(emb, axis=1, keepdims=True)
det = boxes[rng.permutation(n)] + rng.normal(0, 2, (n, 4))
lab = np.ones(n, np.int64)
H = np.eye(3); H[0, 2] = 1.5


def run_assoc():
    ctx.trk_create(slots, boxes)
    ctx.feat_reset(slots)
    ctx.emb_upload(emb)
    ctx.feat_update(slots, np.arange(n, dtype=np.int32))
    tlbr, lost = ctx.trk_step(slots, H, boxes + 1.0, np.ones(n, np.uint8), np.ones(n))
    ctx.assoc_prepare(_lib.METRIC_EUCLIDEAN, slots, boxes, lab, det, lab, np.zeros(n, np.uint8))
    pw = ctx.assoc_get_pairwise(n, n)
    mean, cov = ctx.trk_get_state(slots)
    return np.concatenate([np.asarray(tlbr).ravel(), np.asarray(pw).ravel(), mean.ravel(), cov.ravel()])


victims = (('yolov4 heads', run_yolo), ('osnet last feature map', run_osnet), ('kalman+pairwise', run_assoc))
base = {name: fn() for name, fn in victims}
for kind in ('liteconv', 'osnet', 'yolov4'):
    net, batch = hammer_net(kind)
    net.run(batch)                 # (graph capture happens here, not concurrently with the victims' copies)
    ctx.synchronize()
    stop = []

    def hammer():
        ctx.bind_thread()
        while not stop:
            net.run(batch)
            ctx.synchronize()
    th = threading.Thread(target=hammer)
    th.start()
    res = {}
    try:
        for name, fn in victims:
            if kind == 'osnet' and name == 'kalman+pairwise':
                continue            # (reads the embedding bank the hammer's head layer writes: not a victim of this hammer)
            bad, worst = 0, 0.
            for _ in range(N):
                out = fn()
                if not np.array_equal(out, base[name]):
                    bad += 1
                    worst = max(worst, float(np.nanmax(np.abs(out - base[name]))))
            res[name] = (bad, worst)
    finally:
        stop.append(1)
        th.join()
    net.close()
    print(f'hammer={kind:<9} ' + '; '.join(f'{k}: {v[0]}/{N} runs differ (max |d| {v[1]:.3g})' for k, v in res.items()), flush=True)
