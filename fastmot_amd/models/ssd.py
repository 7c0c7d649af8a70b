"""SSD model descriptors (registry + class attributes of fastmot/models/ssd.py:9-49,99-107,197-205,294-301).

The reference builds these networks from TensorFlow frozen graphs through graphsurgeon + UFF + a TensorRT < 8
parser, with the box decode and NMS inside the engine (NMS_TRT plugin): the engine output already is a list of
TOPK detections per tile.  Neither a TensorFlow graph reader nor those .pb files exist here, so `build_graph`
is not provided; `SSDDetector` (detector.py) implements the tiling / normalisation / filtering / merging
stages of the reference around any inference callable that produces that output (its `backend` argument).
"""
from pathlib import Path


class SSD:
    """Base class: subclasses register themselves by name (fastmot/models/ssd.py:36-44)."""
    _registry = {}

    ENGINE_PATH = None
    MODEL_PATH = None
    NUM_CLASSES = None
    INPUT_SHAPE = None      # (channel, height, width)
    OUTPUT_NAME = None
    NMS_THRESH = None
    TOPK = None             # detections per tile in the network output

    def __init_subclass__(cls, **kwargs):
        super().__init_subclass__(**kwargs)
        SSD._registry[cls.__name__] = cls

    @classmethod
    def get_model(cls, name):
        return SSD._registry[name]

    @classmethod
    def build_graph(cls, weights=None):
        raise NotImplementedError(f'{cls.__name__}: building the network needs a TensorFlow frozen-graph reader '
                                  '(the reference uses graphsurgeon + UFF, TensorRT < 8); pass an inference '
                                  'callable to SSDDetector(backend=...) or use detector_type YOLO / PUBLIC')


class SSDMobileNetV1(SSD):
    ENGINE_PATH = Path(__file__).parent / 'ssd_mobilenet_v1_coco.hipnet'
    MODEL_PATH = Path(__file__).parent / 'ssd_mobilenet_v1_coco.pb'
    NUM_CLASSES = 91
    INPUT_SHAPE = (3, 300, 300)
    OUTPUT_NAME = 'NMS'
    NMS_THRESH = 0.5
    TOPK = 100


class SSDMobileNetV2(SSD):
    ENGINE_PATH = Path(__file__).parent / 'ssd_mobilenet_v2_coco.hipnet'
    MODEL_PATH = Path(__file__).parent / 'ssd_mobilenet_v2_coco.pb'
    NUM_CLASSES = 91
    INPUT_SHAPE = (3, 300, 300)
    OUTPUT_NAME = 'NMS'
    NMS_THRESH = 0.5
    TOPK = 100


class SSDInceptionV2(SSD):
    ENGINE_PATH = Path(__file__).parent / 'ssd_inception_v2_coco.hipnet'
    MODEL_PATH = Path(__file__).parent / 'ssd_inception_v2_coco.pb'
    NUM_CLASSES = 91
    INPUT_SHAPE = (3, 300, 300)
    OUTPUT_NAME = 'NMS'
    NMS_THRESH = 0.5
    TOPK = 100
