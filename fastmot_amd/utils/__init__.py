from .decoder import ConfigDecoder
from .profiler import Profiler
