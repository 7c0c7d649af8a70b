from .decoder import ConfigDecoder
from .profiler import Profiler
from .visualization import Visualizer
