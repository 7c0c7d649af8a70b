import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['FASTMOT_RANDOM_WEIGHTS'] = '1'
import numpy as np
from fastmot_amd import models
from fastmot_amd.detector import YOLODetector
from fastmot_amd.utils.synthetic import SyntheticVideo, ScriptedHeadWeights, scripted_head_weights
size = (1920, 1080)
video = SyntheticVideo(size, n_ids=50, n_frames=2, seed=100)
m = models.YOLO.get_model('YOLOv4_608')
det = YOLODetector(size, (1,), model='YOLOv4_608', weights=ScriptedHeadWeights(0, m.NUM_CLASSES, 1, [[0.0] * 3] * 3), max_candidates=65536)
det.detect_async(video.frames[0]); r = det.postprocess()
print('zero-bias: candidates', det.ctx.detect_last_counts(), 'dets', len(r))
rec = 85
for head in det.heads:
    t = det.backend.read(head, 1)[0]
    v = t.reshape(t.shape[0], t.shape[1], -1, rec)
    print('head', t.shape, 'obj logits: min %.3f max %.3f mean %.3f std %.3f q93 %.3f' % (v[..., 4].min(), v[..., 4].max(), v[..., 4].mean(), v[..., 4].std(), np.quantile(v[..., 4], 0.934)),
          'cls1 mean %.3f cls0 mean %.3f' % (v[..., 6].mean(), v[..., 5].mean()), 'xy mean %.3f' % v[..., 0].mean())
det.backend.close()
w = scripted_head_weights(size, 'YOLOv4_608', 1, video.frames[0], 1500)
print('bias', w.obj_bias)
det = YOLODetector(size, (1,), model='YOLOv4_608', weights=w, max_candidates=65536)
det.detect_async(video.frames[0]); r = det.postprocess()
print('scripted: candidates', det.ctx.detect_last_counts(), 'dets', len(r))
for head in det.heads:
    t = det.backend.read(head, 1)[0]
    v = t.reshape(t.shape[0], t.shape[1], -1, rec)
    print('head obj logits: mean %.3f q93 %.3f; share over cut: %.4f' % (v[..., 4].mean(), np.quantile(v[..., 4], 0.934), (v[..., 4] >= -1.07).mean()))
tot = 0
sig = lambda x: 1 / (1 + np.exp(-x.astype(np.float64)))
for head in det.heads:
    t = det.backend.read(head, 1)[0]
    v = t.reshape(t.shape[0], t.shape[1], -1, rec)
    score = sig(v[..., 4]) * sig(v[..., 5:].max(-1))
    n = int((score >= 0.25).sum()); tot += n
    print('numpy count over threshold', n, 'of', score.size, 'obj logit values at/above cut:', int((v[..., 4] >= -1.0743).sum()), 'unique obj values', len(np.unique(v[..., 4])), 'argmax class != 1:', int((v[..., 5:].argmax(-1) != 1).sum()))
print('numpy total', tot)
head = det.heads[0]
t = det.backend.read(head, 1)[0]
v = t.reshape(t.shape[0], t.shape[1], -1, rec)
score = sig(v[..., 4]) * sig(v[..., 5:].max(-1))
ok = score >= 0.25
print('passing cells: obj logit min %.4f max %.4f; cls logit min %.3f max %.3f mean %.3f' % (v[..., 4][ok].min(), v[..., 4][ok].max(), v[..., 5:].max(-1)[ok].min(), v[..., 5:].max(-1)[ok].max(), v[..., 5:].max(-1)[ok].mean()))
print('all cells: cls1 logit min %.3f max %.3f std %.3f' % (v[..., 6].min(), v[..., 6].max(), v[..., 6].std()))
print('per anchor pass counts', ok.reshape(-1, 3).sum(0), 'per anchor obj mean', v[..., 4].reshape(-1, 3).mean(0))
