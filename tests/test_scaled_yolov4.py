"""Built-in Scaled-YOLOv4 topologies (models/scaled_yolov4.py; reference descriptors models/yolo.py:166-253):
parameter / FLOP totals against the published figures, lowering through the Darknet cfg reader, and the lowered graph
against the independent PyTorch cfg interpreter of tests/darknet_cases.py at a small input size."""
import numpy as np
import pytest
import torch

import darknet_cases as dc
import torch_ref
from fastmot_amd.models import YOLO
from fastmot_amd.models.darknet import darknet_graph, parse_cfg
from fastmot_amd.models.graph import RandomWeights
from fastmot_amd.models.scaled_yolov4 import yolov4_csp_cfg, yolov4_p6_cfg


def totals(text):
    """(parameters, FLOPs = 2 MAC) of the [convolutional] sections, from the cfg alone."""
    cfg = parse_cfg(text)
    net, layers = cfg[0], cfg[1:]
    shapes, params, flops = [], 0, 0
    for i, L in enumerate(layers):
        t = L['type']
        prev = shapes[i - 1] if i else (int(net['channels']), int(net['height']), int(net['width']))
        if t == 'convolutional':
            k, s, f = int(L['size']), int(L['stride']), int(L['filters'])
            h, w = (prev[1] + 2 * (k // 2) - k) // s + 1, (prev[2] + 2 * (k // 2) - k) // s + 1
            params += k * k * prev[0] * f
            flops += 2 * k * k * prev[0] * f * h * w
            shapes.append((f, h, w))
        elif t == 'route':
            src = [shapes[i + int(r) if int(r) < 0 else int(r)] for r in L['layers']]
            shapes.append((sum(s[0] for s in src),) + src[0][1:])
        elif t == 'upsample':
            shapes.append((prev[0], prev[1] * 2, prev[2] * 2))
        else:                       # shortcut, maxpool (stride 1), yolo
            shapes.append(prev)
    return params, flops


def test_published_totals():
    p, f = totals(yolov4_csp_cfg(640, 640, 80))
    assert abs(p / 1e6 - 52.9) < 0.3, p                 # Scaled-YOLOv4 paper / model zoo: 52.9 M parameters
    assert abs(f / 1e9 - 120) < 4, f                    # darknet: ~120 BFLOPs at 640x640 (109 at 608)
    assert abs(totals(yolov4_csp_cfg(608, 608, 80))[1] / 1e9 - 109) < 3
    p, f = totals(yolov4_p6_cfg(1280, 1280, 80))
    assert abs(p / 1e6 - 127.6) < 0.6, p                # 127.6 M parameters
    assert abs(f / 1e9 - 718) < 15, f                   # ~718 BFLOPs at 1280x1280


@pytest.mark.parametrize('name,strides,n_anchor', [('YOLOv4CSP_640', [8, 16, 32], 3), ('YOLOv4P6_1280', [8, 16, 32, 64], 4)])
def test_descriptors_build(name, strides, n_anchor):
    model = YOLO.get_model(name)
    g, heads = model.build_graph(RandomWeights(seed=0))
    _, H, W = model.INPUT_SHAPE
    assert [(h.h, h.w, h.c) for h in heads] == [(H // s, W // s, (model.NUM_CLASSES + 5) * n_anchor) for s in strides]
    assert len(model.ANCHORS) == len(strides) and all(len(a) == 2 * n_anchor for a in model.ANCHORS)
    assert model.NEW_COORDS and model.LETTERBOX


@pytest.mark.parametrize('gen,hw', [(yolov4_csp_cfg, (64, 96)), (yolov4_p6_cfg, (128, 128))])
def test_lowering_matches_independent_interpreter(gen, hw):
    text = gen(hw[1], hw[0], 3)
    cfg = parse_cfg(text)
    blob = dc.random_weights_file(cfg, seed=3)
    from fastmot_amd.models.darknet import DarknetWeights
    g, heads, meta = darknet_graph(text, DarknetWeights(blob), in_hw=hw)
    x = torch.from_numpy(np.random.default_rng(1).uniform(0, 1, (1, 3, *hw)).astype(np.float32))
    ref = dc.torch_darknet(cfg, blob, x)
    bufs, _ = torch_ref.run_graph(g, x, emulate_fp16_storage=False)
    assert len(ref) == len(heads)
    for r, hd in zip(ref, heads):
        got = bufs[hd.tid][:, hd.coff:hd.coff + hd.c]
        assert got.shape == r.shape
        np.testing.assert_allclose(got.numpy(), r.numpy(), rtol=2e-4, atol=2e-5)
