"""Label map (class id -> name); same accessors and default names as
fastmot/models/label.py:1-121: index 0 is 'head' (CrowdHuman), followed by the 90-entry COCO
paper category list, so that class id 1 is 'person' as cfg/mot.json's class_ids expects."""
from collections.abc import Sequence

_DEFAULT = ';'.join((
    'head;person;bicycle;car;motorcycle;airplane;bus;train;truck;boat;traffic light;fire hydrant',
    'street sign;stop sign;parking meter;bench;bird;cat;dog;horse;sheep;cow;elephant;bear;zebra',
    'giraffe;hat;backpack;umbrella;shoe;eye glasses;handbag;tie;suitcase;frisbee;skis;snowboard',
    'sports ball;kite;baseball bat;baseball glove;skateboard;surfboard;tennis racket;bottle;plate',
    'wine glass;cup;fork;knife;spoon;bowl;banana;apple;sandwich;orange;broccoli;carrot;hot dog',
    'pizza;donut;cake;chair;couch;potted plant;bed;mirror;dining table;window;desk;toilet;door;tv',
    'laptop;mouse;remote;keyboard;cell phone;microwave;oven;toaster;sink;refrigerator;blender',
    'book;clock;vase;scissors;teddy bear;hair drier;toothbrush'))

_label_map = tuple(_DEFAULT.split(';'))


def get_label_name(class_id):
    """Look up label name given a class ID."""
    return _label_map[class_id]


def set_label_map(label_map):
    """Set label name mapping from class IDs (index = class id)."""
    assert isinstance(label_map, Sequence)
    assert len(label_map) > 0
    global _label_map
    _label_map = tuple(label_map)
