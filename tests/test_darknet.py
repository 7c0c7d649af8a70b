"""Darknet cfg/weights loader (fastmot_amd/models/darknet.py, next row n3 of SURVEY section 8f): the layer
table built from a cfg, evaluated by the test interpreter (tests/torch_ref.py, CPU) and by the HIP engine
(GPU), must equal an independent PyTorch interpretation of the same cfg + weights file."""
from collections import Counter

import numpy as np
import pytest
import torch

import darknet_cases as dc
import torch_ref
from fastmot_amd.models import darknet
from fastmot_amd.models import graph as G

CASES = {'mini_v4': dc.MINI_V4, 'mini_tiny': dc.MINI_TINY, 'mini_res': dc.MINI_RES}


def build(name, seed=1):
    cfg = darknet.parse_cfg(CASES[name])
    blob = dc.random_weights_file(cfg, seed)
    w = darknet.DarknetWeights(blob)
    g, heads, meta = darknet.darknet_graph(cfg, w)
    assert w.remaining() == 0
    return cfg, blob, g, heads, meta


@pytest.mark.parametrize('name', list(CASES))
def test_graph_matches_independent_interpretation(name):
    cfg, blob, g, heads, meta = build(name)
    _, H, W = meta['input_shape']
    x = torch.from_numpy(np.random.default_rng(2).uniform(0, 1, (2, 3, H, W)).astype(np.float32))
    ref = dc.torch_darknet(cfg, blob, x)
    bufs, _ = torch_ref.run_graph(g, x, emulate_fp16_storage=False)
    assert len(ref) == len(heads)
    for hv, r in zip(heads, ref):
        got = bufs[hv.tid][:, hv.coff:hv.coff + hv.c]
        assert got.shape == r.shape
        err = (got - r).abs().max().item()
        assert err <= 1e-2 * r.abs().max().item() + 1e-3, f'{name}: head err {err}'   # fp16-rounded weights


def test_lowering_mini_v4():
    _, _, g, heads, meta = build('mini_v4')
    ops = Counter(d['op'] for d in g.layers)
    # 18 convs (the first one on the stem kernel; the two sibling 1x1 convs of the CSP stage run as one),
    # SPP fused, shortcut + upsample folded into convs, every concat operand written in place:
    # 29 cfg sections -> 18 launches
    assert ops[G.OP_CONV] + ops[G.OP_CONVS] + ops[G.OP_CONVD] + ops[G.OP_STEMCONV] == 17 and ops[G.OP_STEMCONV] == 1 and len(g.layers) == 18
    merged = [d for d in g.layers if d['op'] in G.CONV_OPS and d['name'] == '004_convolutional']
    assert len(merged) == 1 and merged[0]['cout'] == 32 and merged[0]['out'].coff == 0      # [b | A]: 16 + 16
    assert ops[G.OP_SPP] == 1 and ops[G.OP_MAXPOOL] == 0
    assert ops[G.OP_ADD] == 0 and ops[G.OP_UPSAMPLE2] == 0
    assert sum(1 for d in g.layers if d['op'] in G.CONV_OPS and d['up'] == 2) == 1
    assert sum(1 for d in g.layers if d['op'] in G.CONV_OPS and d['res'] is not None) == 1
    assert ops[G.OP_COPY] == 0
    assert meta['classes'] == 2 and meta['strides'] == [2, 4] and meta['scales'] == [1.2, 1.1]
    assert meta['anchors'] == [[12, 16, 19, 36, 40, 28], [36, 75, 76, 55, 72, 146]] and meta['new_coords'] is False
    assert all(g.tensors[h.tid][3] == 1 for h in heads)          # fp32 heads


def test_lowering_mini_res(monkeypatch):
    """Residual units (1x1, 3x3, shortcut from=-3) of 64 channels lower to one fused launch each; the second
    one writes straight into the concat tensor.  FASTMOT_RESBLOCK=0 keeps the conv + folded-shortcut form."""
    _, _, g, heads, _ = build('mini_res')
    ops = Counter(d['op'] for d in g.layers)
    assert ops[G.OP_RESBLOCK] == 2 and ops[G.OP_ADD] == 0 and ops[G.OP_COPY] == 0 and len(g.layers) == 6
    res = [d for d in g.layers if d['op'] == G.OP_RESBLOCK]
    assert [d['hid'] for d in res] == [32, 64] and res[1]['out'].tid == g.layers[-1]['ins'][0].tid
    # (no sibling merge here: the concat slot below branch A is filled by a residual unit, not a plain 1x1)
    assert sum(1 for d in g.layers if d['op'] in G.CONV_OPS and d['cout'] == 128) == 0
    monkeypatch.setenv('FASTMOT_RESBLOCK', '0')
    _, _, g0, _, _ = build('mini_res')
    ops0 = Counter(d['op'] for d in g0.layers)
    assert ops0[G.OP_RESBLOCK] == 0 and len(g0.layers) == 8
    assert sum(1 for d in g0.layers if d['op'] in G.CONV_OPS and d['res'] is not None) == 2


def test_full_yolov4_cfg_lowers_like_the_builtin_graph():
    """The loader on a generated yolov4.cfg (110 conv sections, 162 sections) produces the same launch list as
    the hand-written YOLOv4 graph -- every fusion (stem, sibling 1x1 merge, residual units, SPP, upsample in
    the producer, in-place concat, streamed convs) is found from the cfg alone -- and evaluates like an
    independent PyTorch interpretation of the cfg + weights file."""
    from fastmot_amd.models import YOLO
    text = dc.yolov4_cfg(64, 64, classes=3)
    cfg = darknet.parse_cfg(text)
    assert sum(1 for L in cfg[1:] if L['type'] == 'convolutional') == 110 and len(cfg) - 1 == 162
    blob = dc.random_weights_file(cfg, seed=3)
    w = darknet.DarknetWeights(blob)
    g, heads, meta = darknet.darknet_graph(cfg, w)
    assert w.remaining() == 0 and meta['strides'] == [8, 16, 32] and meta['scales'] == [1.2, 1.1, 1.05]

    class Small(YOLO.get_model('YOLOv4')):
        INPUT_SHAPE = (3, 64, 64)
    ref_g, _ = Small.build_graph(G.RandomWeights(seed=1))
    sig = lambda gr: [(d['op'], d['k'], d['stride'], d['cin'], d['hid'], d['up'], d['out'].h) for d in gr.layers]
    got, exp = sig(g), sig(ref_g)
    assert len(got) == len(exp) == 85          # (87 before round 6: the stem, the stride-2 conv and the merged 1x1 conv behind it are one entry now)
    assert [s[:3] + s[4:] for s in got] == [s[:3] + s[4:] for s in exp]      # (cin of the heads aside: 24 vs 255 couts)
    x = torch.from_numpy(np.random.default_rng(5).uniform(0, 1, (1, 3, 64, 64)).astype(np.float32))
    ref = dc.torch_darknet(cfg, blob, x)
    bufs, _ = torch_ref.run_graph(g, x, emulate_fp16_storage=False)
    for hv, r in zip(heads, ref):
        got = bufs[hv.tid][:, hv.coff:hv.coff + hv.c]
        assert (got - r).abs().max().item() <= 1e-2 * r.abs().max().item() + 1e-3


def test_lowering_mini_tiny():
    _, _, g, heads, meta = build('mini_tiny')
    ops = Counter(d['op'] for d in g.layers)
    assert ops[G.OP_ADD] == 1 and ops[G.OP_UPSAMPLE2] == 1 and ops[G.OP_MAXPOOL] == 2
    assert meta['new_coords'] is True and meta['strides'] == [4, 4]
    assert [d['act'] for d in g.layers if d['out'].tid in {h.tid for h in heads}] == [G.ACT['logistic']] * 2


def test_weights_file_mismatch_is_an_error(tmp_path):
    cfg = darknet.parse_cfg(dc.MINI_TINY)
    blob = dc.random_weights_file(cfg)
    with pytest.raises(ValueError):
        darknet.darknet_graph(cfg, darknet.DarknetWeights(blob[:-40]))
    (tmp_path / 'a.cfg').write_text(dc.MINI_TINY)
    (tmp_path / 'a.weights').write_bytes(blob + b'\0' * 8)
    with pytest.raises(ValueError):
        darknet.load_darknet(tmp_path / 'a.cfg', tmp_path / 'a.weights')
    (tmp_path / 'a.weights').write_bytes(blob)
    g, heads, _ = darknet.load_darknet(tmp_path / 'a.cfg', tmp_path / 'a.weights')
    assert len(heads) == 2
    with pytest.raises(ValueError):
        darknet.parse_cfg('[net]\nwidth=32\n[dropout]\nprobability=.5\n')


@pytest.mark.gpu
@pytest.mark.parametrize('name', list(CASES))
def test_engine_matches_independent_interpretation(ctx, name):
    from fastmot_amd.engine import HipNet, NET_DETECTOR
    cfg, blob, g, heads, meta = build(name)
    _, H, W = meta['input_shape']
    x = np.random.default_rng(3).uniform(0, 1, (2, H, W, 3)).astype(np.float16)
    ref = dc.torch_darknet(cfg, blob, torch.from_numpy(x.astype(np.float32).transpose(0, 3, 1, 2)))
    for reuse in (False, True):
        net = HipNet(ctx, NET_DETECTOR, g, 2, reuse_buffers=reuse)
        net.write(g.input, x)
        net.run(2)
        for hv, r in zip(heads, ref):
            got = net.read(hv, 2)
            r = r.numpy().transpose(0, 2, 3, 1)
            err = np.abs(got - r).max()
            assert err <= 2e-2 * np.abs(r).max() + 2e-3, f'{name}: head err {err}'
        net.close()


def test_yolo_descriptor_builds_from_cfg_and_weights(tmp_path):
    """A YOLO descriptor whose MODEL_PATH (.weights) and cfg exist loads topology + real weights from them
    (what YOLODetector does at start-up); a cfg that contradicts the descriptor is rejected."""
    from fastmot_amd.models import YOLO
    cfg = darknet.parse_cfg(dc.MINI_V4)
    blob = dc.random_weights_file(cfg, seed=7)
    (tmp_path / 'mini.cfg').write_text(dc.MINI_V4)
    (tmp_path / 'mini.weights').write_bytes(blob)

    class Mini(YOLO):
        MODEL_PATH = tmp_path / 'mini.weights'
        NUM_CLASSES = 2
        INPUT_SHAPE = (3, 64, 96)
        LAYER_FACTORS = [2, 4]
        SCALES = [1.2, 1.1]
        ANCHORS = [[12, 16, 19, 36, 40, 28], [36, 75, 76, 55, 72, 146]]
        TOPOLOGY = 'darknet-cfg'

    g, heads = Mini.build_graph()
    x = torch.from_numpy(np.random.default_rng(4).uniform(0, 1, (1, 3, 64, 96)).astype(np.float32))
    ref = dc.torch_darknet(cfg, blob, x)
    bufs, _ = torch_ref.run_graph(g, x, emulate_fp16_storage=False)
    for hv, r in zip(heads, ref):
        assert (bufs[hv.tid][:, :hv.c] - r).abs().max().item() <= 1e-2 * r.abs().max().item() + 1e-3

    class Wrong(Mini):
        NUM_CLASSES = 3
    with pytest.raises(ValueError):
        Wrong.build_graph()

    (tmp_path / 'mini.weights').write_bytes(blob + b'\0' * 4)
    with pytest.raises(ValueError):
        Mini.build_graph()


def test_reference_model_registry_is_complete():
    """Every detector name the reference's cfg/mot.json may select (models/yolo.py:154-299) resolves."""
    from fastmot_amd.models import YOLO
    for name, shape in (('YOLOv4', (3, 512, 512)), ('YOLOv4CSP', (3, 640, 640)), ('YOLOv4xMish', (3, 640, 640)),
                        ('YOLOv4CSPSwish', (3, 640, 640)), ('YOLOv4CSPxSwish', (3, 640, 640)),
                        ('YOLOv4P5', (3, 896, 896)), ('YOLOv4P6', (3, 1280, 1280)), ('YOLOv4Tiny', (3, 416, 416)),
                        ('YOLOv3', (3, 416, 416)), ('YOLOv3SPP', (3, 608, 608)), ('YOLOv3Tiny', (3, 416, 416))):
        m = YOLO.get_model(name)
        assert m.INPUT_SHAPE == shape and len(m.ANCHORS) == len(m.LAYER_FACTORS)
