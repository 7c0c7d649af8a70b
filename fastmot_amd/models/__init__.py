from .ssd import SSD
from .yolo import YOLO
from .reid import ReID
from .label import get_label_name, set_label_map
from .graph import allow_random_weights
