from .label import get_label_name, set_label_map
