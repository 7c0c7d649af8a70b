"""YOLO model descriptors (registry + class attributes of fastmot/models/yolo.py:11-58,154-299)
and their layer tables for the HIP conv engine.

The reference builds a TensorRT engine from an ONNX file and appends the YoloLayer_TRT plugin
(models/yolo.py:61-151); here `build_graph` emits the Darknet topology directly (semantics of
scripts/yolo2onnx.py:558-863: Conv SAME + BN eps 1e-5 + mish/leaky/linear, shortcut = Add,
route = concat with the most recent tensor first, maxpool SAME stride 1, nearest upsample) and
the decode runs in detect.hip.  MODEL_PATH / ENGINE_PATH are kept as attributes: MODEL_PATH may
point to Darknet weights; without a file the network runs with seeded random weights.
"""
from pathlib import Path

from .graph import Graph, RandomWeights


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

    def __init_subclass__(cls, **kwargs):
        super().__init_subclass__(**kwargs)
        cls.__registry[cls.__name__] = cls

    @classmethod
    def get_model(cls, name):
        return cls.__registry[name]

    @classmethod
    def build_graph(cls, weights=None):
        """-> (Graph, [head output views])  heads in LAYER_FACTORS order."""
        if weights is None:
            weights = RandomWeights(seed=0)
        assert len(cls.LAYER_FACTORS) == len(cls.SCALES) == len(cls.ANCHORS) or cls.TOPOLOGY != 'yolov4'
        if cls.TOPOLOGY == 'yolov4':
            return yolov4_graph(cls, weights)
        raise NotImplementedError(f'topology {cls.TOPOLOGY} has no layer table yet')


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

    def csp(x, c_out, n_res, h, m):
        d = conv(x, c_out, 3, 2)
        cat = g.new(d.h, d.w, 2 * h)
        conv(d, h, 1, dst=cat.slice(h, h))            # route branch A (second in the concat)
        b = conv(d, h, 1)
        for _ in range(n_res):
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
    MODEL_PATH = Path(__file__).parent / 'yolov4_crowdhuman.weights'
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
    MODEL_PATH = Path(__file__).parent / 'yolov4.weights'
    NUM_CLASSES = 80
    INPUT_SHAPE = (3, 608, 608)
    LAYER_FACTORS = [8, 16, 32]
    SCALES = [1.2, 1.1, 1.05]
    ANCHORS = [[12, 16, 19, 36, 40, 28],
               [36, 75, 76, 55, 72, 146],
               [142, 110, 192, 243, 459, 401]]


# The following descriptors are supported by the reference "but not provided" (models/yolo.py:166-299);
# their metadata is kept so configs resolve, the layer tables are future work.
class YOLOv4CSP(YOLO):
    ENGINE_PATH = Path(__file__).parent / 'yolov4-csp.hipnet'
    MODEL_PATH = Path(__file__).parent / 'yolov4-csp.weights'
    NUM_CLASSES = 1
    LETTERBOX = True
    NEW_COORDS = True
    INPUT_SHAPE = (3, 640, 640)
    LAYER_FACTORS = [8, 16, 32]
    SCALES = [2.0, 2.0, 2.0]
    ANCHORS = [[12, 16, 19, 36, 40, 28],
               [36, 75, 76, 55, 72, 146],
               [142, 110, 192, 243, 459, 401]]
    TOPOLOGY = 'yolov4-csp'


class YOLOv4P6(YOLO):
    ENGINE_PATH = Path(__file__).parent / 'yolov4-p6.hipnet'
    MODEL_PATH = Path(__file__).parent / 'yolov4-p6.weights'
    NUM_CLASSES = 1
    LETTERBOX = True
    NEW_COORDS = True
    INPUT_SHAPE = (3, 1280, 1280)
    LAYER_FACTORS = [8, 16, 32, 64]
    SCALES = [2.0, 2.0, 2.0, 2.0]
    ANCHORS = [[13, 17, 31, 25, 24, 51, 61, 45],
               [61, 45, 48, 102, 119, 96, 97, 189],
               [97, 189, 217, 184, 171, 384, 324, 451],
               [324, 451, 545, 357, 616, 618, 1024, 1024]]
    TOPOLOGY = 'yolov4-p6'
