"""YOLO model descriptors (registry + class attributes of fastmot/models/yolo.py:11-58,154-299)
and their layer tables for the HIP conv engine.

The reference builds a TensorRT engine from an ONNX file and appends the YoloLayer_TRT plugin
(models/yolo.py:61-151); here `build_graph` emits the Darknet topology directly (semantics of
scripts/yolo2onnx.py:558-863: Conv SAME + BN eps 1e-5 + mish/leaky/linear, shortcut = Add,
route = concat with the most recent tensor first, maxpool SAME stride 1, nearest upsample) and
the decode runs in detect.hip.  MODEL_PATH / ENGINE_PATH are kept as attributes: MODEL_PATH is the reference's ONNX
file (read by models/onnx_reader.py; the Darknet .weights / .cfg of the same stem work as well); without a file
the descriptor raises, or runs with seeded random weights when that was opted into.
"""
import os
from pathlib import Path

import numpy as np

from . import scaled_yolov4
from .darknet import DarknetWeights, darknet_graph
from .graph import Graph, fold_bn, missing_weights


class YOLO:
    """Base class for YOLO models (attributes as fastmot/models/yolo.py:11-50).

    PLUGIN_PATH : kept for API compatibility (the decode plugin is part of libfastmot_hip.so).
    ENGINE_PATH / MODEL_PATH : cache / weight file locations.
    NUM_CLASSES, LETTERBOX, NEW_COORDS, INPUT_SHAPE (c, h, w), LAYER_FACTORS, SCALES, ANCHORS.
    """
    __registry = {}

    PLUGIN_PATH = Path(__file__).parents[1] / 'libfastmot_hip.so'
    ENGINE_PATH = None
    MODEL_PATH = None
    NUM_CLASSES = None
    LETTERBOX = False
    NEW_COORDS = False
    INPUT_SHAPE = None
    LAYER_FACTORS = None
    SCALES = None
    ANCHORS = None
    TOPOLOGY = 'yolov4'
    BUILTIN_CFG = None          # name of a generator in models/scaled_yolov4.py (used when no .cfg file is present)

    def __init_subclass__(cls, **kwargs):
        super().__init_subclass__(**kwargs)
        cls.__registry[cls.__name__] = cls

    @classmethod
    def get_model(cls, name):
        return cls.__registry[name]

    @classmethod
    def cfg_path(cls):
        """Darknet cfg next to the model file (the reference converts cfg + weights to ONNX offline,
        scripts/yolo2onnx.py; here a cfg is read directly when there is one)."""
        return cls.MODEL_PATH.with_suffix('.cfg') if cls.MODEL_PATH is not None else None

    @classmethod
    def weight_file(cls):
        """MODEL_PATH (the reference's .onnx, models/yolo.py:155-156) if it exists, else the Darknet .weights of the
        same stem it was converted from; None when neither is there."""
        if cls.MODEL_PATH is None:
            return None
        base = Path(cls.MODEL_PATH)
        for cand in (base, base.with_suffix('.weights')):
            if cand.is_file():
                return cand
        return None

    @classmethod
    def build_graph(cls, weights=None):
        """-> (Graph, [head output views])  heads in LAYER_FACTORS order.

        Weights: the explicit `weights` source, else the model file: the ONNX file scripts/yolo2onnx.py wrote
        (models/onnx_reader.py) or the Darknet .weights next to it; when neither exists this raises
        FileNotFoundError like the reference does, unless seeded random parameters were opted into
        (models.allow_random_weights(), benchmarks / tests).  Topology, in this order: the Darknet cfg next to the
        model file (any YOLOv3/v4/-tiny/Scaled-YOLOv4 cfg, models/darknet.py); the node list of the ONNX file itself
        (rebuilt into cfg sections, onnx_reader.darknet_cfg_from_onnx); the built-in yolov4.cfg table / Scaled-YOLOv4
        generators."""
        real = None
        onnx_model = None
        if weights is None:
            path = cls.weight_file()
            if path is None:
                weights = missing_weights(cls, seed=0)
            elif path.suffix == '.onnx':
                from .onnx_reader import OnnxDarknetWeights
                weights = real = OnnxDarknetWeights(path)
                onnx_model = weights.model
            else:
                weights = real = DarknetWeights(path)
        cfg = cls.cfg_path()
        text = None
        if cfg is not None and cfg.is_file():
            text, origin = cfg.read_text(), str(cfg)
        elif onnx_model is not None:
            from .onnx_reader import darknet_cfg_from_onnx
            text, origin = darknet_cfg_from_onnx(onnx_model, cls), f'the node list of {cls.MODEL_PATH}'
            shape = onnx_model.data_inputs[0][1]
            if tuple(shape[1:]) != tuple(cls.INPUT_SHAPE):      # (as models/yolo.py:120 of the reference asserts)
                raise ValueError(f'{cls.MODEL_PATH}: input {tuple(shape[1:])} != {cls.__name__}.INPUT_SHAPE {cls.INPUT_SHAPE}')
        if text is not None:
            g, heads, meta = darknet_graph(text, weights, in_hw=cls.INPUT_SHAPE[1:])
            if meta.get('classes') != cls.NUM_CLASSES or len(heads) != len(cls.LAYER_FACTORS) or \
                    meta.get('strides') != list(cls.LAYER_FACTORS) or meta.get('new_coords') != cls.NEW_COORDS:
                raise ValueError(f'{origin} does not describe {cls.__name__}: {meta}')
        elif cls.TOPOLOGY == 'yolov4':
            assert len(cls.LAYER_FACTORS) == len(cls.SCALES) == len(cls.ANCHORS)
            g, heads = yolov4_graph(cls, weights)
        elif cls.BUILTIN_CFG is not None:
            # Scaled-YOLOv4 topologies generated section by section (models/scaled_yolov4.py)
            text = getattr(scaled_yolov4, cls.BUILTIN_CFG)(cls.INPUT_SHAPE[2], cls.INPUT_SHAPE[1], cls.NUM_CLASSES)
            g, heads, meta = darknet_graph(text, weights, in_hw=cls.INPUT_SHAPE[1:])
            assert meta['strides'] == list(cls.LAYER_FACTORS) and meta['new_coords'] == cls.NEW_COORDS
        else:
            raise NotImplementedError(f'{cls.__name__}: no built-in layer table; put the Darknet cfg at {cfg}')
        if real is not None and real.remaining() != 0:
            raise ValueError(f'{cls.weight_file()}: {real.remaining()} '
                             f'{"layers" if onnx_model is not None else "bytes"} left after loading {cls.__name__}')
        return g, heads


def yolov4_graph(model, weights):
    """Darknet yolov4.cfg: CSPDarknet53 (mish) + SPP + PAN (leaky 0.1) + 3 heads
    (110 conv layers, 128.4 GFLOP @608^2 / 80 classes)."""
    _, H, W = model.INPUT_SHAPE
    assert H % 32 == 0 and W % 32 == 0
    g = Graph(weights, (H, W), 3)
    n = [0]

    def name():
        n[0] += 1
        return f'conv{n[0]}'

    def conv(x, cout, k=1, stride=1, act='mish', **kw):
        return g.conv(name(), x, cout, k, stride, act, **kw)

    merge_siblings = os.environ.get('FASTMOT_CSP_MERGE', '1') != '0'

    def csp(x, c_out, n_res, h, m):
        d = conv(x, c_out, 3, 2)
        cat = g.new(d.h, d.w, 2 * h)
        if merge_siblings:
            # the two 1x1 convs that read d (route branch A, second in the concat; residual branch) as ONE
            # conv with stacked weights: d is read once, one launch fewer.  Its output IS the concat tensor:
            # [b0 | A]; b0 is dead once the first residual unit has run and the stage's last 1x1 overwrites it.
            (wa, ba), (wb_, bb) = (fold_bn(g.wsrc.conv(name(), h, c_out, 1, bn=True)) for _ in range(2))
            conv(d, 2 * h, 1, dst=cat, wb=(np.concatenate([wb_, wa]), np.concatenate([bb, ba])))
            b = cat.slice(0, h)
        else:
            conv(d, h, 1, dst=cat.slice(h, h))        # route branch A (second in the concat)
            b = conv(d, h, 1)
        for _ in range(n_res):
            if g.use_resblock and g.resblock_supported(h, m) and g.resblock_pays(h, m, b.h, b.w):
                b = g.resblock(name(), name(), b, m)   # 1x1 + 3x3 + shortcut in one launch
                continue
            t = conv(b, m, 1)
            b = conv(t, h, 3, res=b)                   # shortcut (linear) : act(conv) + b
        conv(b, h, 1, dst=cat.slice(0, h))            # most recent tensor first
        return conv(cat, c_out, 1)

    x = conv(g.input, 32, 3)
    x = csp(x, 64, 1, 64, 32)
    x = csp(x, 128, 2, 64, 64)
    t3 = x = csp(x, 256, 8, 128, 128)
    t4 = x = csp(x, 512, 8, 256, 256)
    x = csp(x, 1024, 4, 512, 512)

    lk = dict(act='leaky')
    x = conv(x, 512, 1, **lk)
    x = conv(x, 1024, 3, **lk)
    spp = g.new(x.h, x.w, 2048)
    xs = conv(x, 512, 1, dst=spp.slice(1536, 512), **lk)
    g.spp(xs, spp.slice(0, 1536))
    x = conv(spp, 512, 1, **lk)
    x = conv(x, 1024, 3, **lk)
    # concat operands are written in place by their producers (no copy layers), and the two
    # [upsample] layers are folded into the 1x1 convs that feed them (conv ... up=2)
    d5 = g.new(x.h, x.w, 1024)
    p5 = conv(x, 512, 1, dst=d5.slice(512, 512), **lk)

    def five(x, c, dst=None):
        x = conv(x, c, 1, **lk)
        x = conv(x, 2 * c, 3, **lk)
        x = conv(x, c, 1, **lk)
        x = conv(x, 2 * c, 3, **lk)
        return conv(x, c, 1, dst=dst, **lk)

    u4 = g.new(t4.h, t4.w, 512)
    conv(p5, 256, 1, dst=u4.slice(256, 256), up=2, **lk)
    conv(t4, 256, 1, dst=u4.slice(0, 256), **lk)
    d4 = g.new(t4.h, t4.w, 512)
    n4 = five(u4, 256, dst=d4.slice(256, 256))

    u3 = g.new(t3.h, t3.w, 256)
    conv(n4, 128, 1, dst=u3.slice(128, 128), up=2, **lk)
    conv(t3, 128, 1, dst=u3.slice(0, 128), **lk)
    n3 = five(u3, 128)

    n_out = (5 + model.NUM_CLASSES) * (len(model.ANCHORS[0]) // 2)
    # new_coords models (Scaled-YOLOv4) apply the logistic activation in the conv before [yolo]
    head_act = 'logistic' if model.NEW_COORDS else 'linear'
    heads = []
    x = conv(n3, 256, 3, **lk)
    heads.append(conv(x, n_out, 1, act=head_act, bn=False, f32_out=True))

    conv(n3, 256, 3, 2, dst=d4.slice(0, 256), **lk)
    m4 = five(d4, 256)
    x = conv(m4, 512, 3, **lk)
    heads.append(conv(x, n_out, 1, act=head_act, bn=False, f32_out=True))

    conv(m4, 512, 3, 2, dst=d5.slice(0, 512), **lk)
    m5 = five(d5, 512)
    x = conv(m5, 1024, 3, **lk)
    heads.append(conv(x, n_out, 1, act=head_act, bn=False, f32_out=True))
    g.outputs = heads
    return g, heads


class YOLOv4(YOLO):
    ENGINE_PATH = Path(__file__).parent / 'yolov4_crowdhuman.hipnet'
    MODEL_PATH = Path(__file__).parent / 'yolov4_crowdhuman.onnx'
    NUM_CLASSES = 2
    INPUT_SHAPE = (3, 512, 512)
    LAYER_FACTORS = [8, 16, 32]
    SCALES = [1.2, 1.1, 1.05]
    ANCHORS = [[11, 22, 24, 60, 37, 116],
               [54, 186, 69, 268, 89, 369],
               [126, 491, 194, 314, 278, 520]]


class YOLOv4_608(YOLO):
    """BASELINE.json config[1]: Darknet yolov4.cfg at 608x608, 80 COCO classes (the configuration
    the 128.4 GFLOP figure refers to); anchors/scales of the public yolov4.cfg."""
    ENGINE_PATH = Path(__file__).parent / 'yolov4_608.hipnet'
    MODEL_PATH = Path(__file__).parent / 'yolov4.onnx'
    NUM_CLASSES = 80
    INPUT_SHAPE = (3, 608, 608)
    LAYER_FACTORS = [8, 16, 32]
    SCALES = [1.2, 1.1, 1.05]
    ANCHORS = [[12, 16, 19, 36, 40, 28],
               [36, 75, 76, 55, 72, 146],
               [142, 110, 192, 243, 459, 401]]


# The following descriptors are supported by the reference "but not provided" (models/yolo.py:166-299).
# Their metadata is identical; the topology comes from the Darknet cfg placed next to MODEL_PATH
# (models/darknet.py builds the layer table from it).
_COCO_ANCHORS = [[12, 16, 19, 36, 40, 28], [36, 75, 76, 55, 72, 146], [142, 110, 192, 243, 459, 401]]


def _scaled(name, input_shape, factors, anchors, scales=2.0, builtin=None):
    return type(name, (YOLO,), dict(
        BUILTIN_CFG=builtin,
        ENGINE_PATH=Path(__file__).parent / f'{_file_name(name)}.hipnet',
        MODEL_PATH=Path(__file__).parent / f'{_file_name(name)}.onnx',
        NUM_CLASSES=1, LETTERBOX=True, NEW_COORDS=True, INPUT_SHAPE=input_shape, LAYER_FACTORS=factors,
        SCALES=[scales] * len(factors), ANCHORS=anchors, TOPOLOGY='darknet-cfg', __module__=__name__))


def _file_name(name):
    return {'YOLOv4CSP': 'yolov4-csp', 'YOLOv4xMish': 'yolov4x-mish', 'YOLOv4CSPSwish': 'yolov4-csp-swish',
            'YOLOv4CSPxSwish': 'yolov4-csp-x-swish', 'YOLOv4P5': 'yolov4-p5', 'YOLOv4P6': 'yolov4-p6'}[name]


YOLOv4CSP = _scaled('YOLOv4CSP', (3, 640, 640), [8, 16, 32], _COCO_ANCHORS, builtin='yolov4_csp_cfg')
YOLOv4xMish = _scaled('YOLOv4xMish', (3, 640, 640), [8, 16, 32], _COCO_ANCHORS)
YOLOv4CSPSwish = _scaled('YOLOv4CSPSwish', (3, 640, 640), [8, 16, 32], _COCO_ANCHORS)
YOLOv4CSPxSwish = _scaled('YOLOv4CSPxSwish', (3, 640, 640), [8, 16, 32], _COCO_ANCHORS)
YOLOv4P5 = _scaled('YOLOv4P5', (3, 896, 896), [8, 16, 32],
                   [[13, 17, 31, 25, 24, 51, 61, 45], [48, 102, 119, 96, 97, 189, 217, 184],
                    [171, 384, 324, 451, 616, 618, 800, 800]])
YOLOv4P6 = _scaled('YOLOv4P6', (3, 1280, 1280), [8, 16, 32, 64],
                   [[13, 17, 31, 25, 24, 51, 61, 45], [61, 45, 48, 102, 119, 96, 97, 189],
                    [97, 189, 217, 184, 171, 384, 324, 451], [324, 451, 545, 357, 616, 618, 1024, 1024]],
                   builtin='yolov4_p6_cfg')


class YOLOv4CSP_640(YOLOv4CSP):
    """BASELINE.json config[2]: Scaled-YOLOv4 CSP at 640x640 with the 80 COCO classes (52.9 M parameters,
    121 GFLOP), seeded random weights unless yolov4-csp.weights is present."""
    NUM_CLASSES = 80


class YOLOv4P6_1280(YOLOv4P6):
    """BASELINE.json config[4]: Scaled-YOLOv4 P6 at 1280x1280, 80 classes, 4 heads x 4 anchors
    (127.5 M parameters, 722 GFLOP)."""
    NUM_CLASSES = 80


class YOLOv4Tiny(YOLO):
    ENGINE_PATH = Path(__file__).parent / 'yolov4-tiny.hipnet'
    MODEL_PATH = Path(__file__).parent / 'yolov4-tiny.onnx'
    NUM_CLASSES = 1
    INPUT_SHAPE = (3, 416, 416)
    LAYER_FACTORS = [32, 16]
    SCALES = [1.05, 1.05]
    ANCHORS = [[81, 82, 135, 169, 344, 319], [23, 27, 37, 58, 81, 82]]
    TOPOLOGY = 'darknet-cfg'


class YOLOv3(YOLO):
    ENGINE_PATH = Path(__file__).parent / 'yolov3.hipnet'
    MODEL_PATH = Path(__file__).parent / 'yolov3.onnx'
    NUM_CLASSES = 1
    INPUT_SHAPE = (3, 416, 416)
    LAYER_FACTORS = [32, 16, 8]
    SCALES = [1., 1.]                # as in the reference (models/yolo.py:272): two entries for three heads
    ANCHORS = [[116, 90, 156, 198, 373, 326], [30, 61, 62, 45, 59, 119], [10, 13, 16, 30, 33, 23]]
    TOPOLOGY = 'darknet-cfg'


class YOLOv3SPP(YOLOv3):
    ENGINE_PATH = Path(__file__).parent / 'yolov3-spp.hipnet'
    MODEL_PATH = Path(__file__).parent / 'yolov3-spp.onnx'
    INPUT_SHAPE = (3, 608, 608)


class YOLOv3Tiny(YOLO):
    ENGINE_PATH = Path(__file__).parent / 'yolov3-tiny.hipnet'
    MODEL_PATH = Path(__file__).parent / 'yolov3-tiny.onnx'
    NUM_CLASSES = 1
    INPUT_SHAPE = (3, 416, 416)
    LAYER_FACTORS = [32, 16]
    SCALES = [1., 1.]
    ANCHORS = [[81, 82, 135, 169, 344, 319], [10, 14, 23, 27, 37, 58]]
    TOPOLOGY = 'darknet-cfg'
