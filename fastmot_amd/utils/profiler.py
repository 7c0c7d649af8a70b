"""Named wall-clock accumulators; same observable API as fastmot/utils/profiler.py:5-33
(`with Profiler(name, aggregate)`, `.duration`, `Profiler.reset()`, `Profiler.get_avg_millis`)."""
import time
from collections import defaultdict


class Profiler:
    _calls = defaultdict(int)
    _elapsed = defaultdict(float)

    def __init__(self, name, aggregate=False):
        self.name = name
        self.start = self.end = self.duration = 0.
        if not aggregate:
            Profiler._calls[name] += 1

    def __enter__(self):
        self.start = time.perf_counter()
        return self

    def __exit__(self, exc_type, exc, tb):
        self.end = time.perf_counter()
        self.duration = self.end - self.start
        Profiler._elapsed[self.name] += self.duration

    @classmethod
    def reset(cls):
        cls._calls.clear()
        cls._elapsed.clear()

    @classmethod
    def get_avg_millis(cls, name):
        calls = cls._calls.get(name, 0)
        return cls._elapsed.get(name, 0.) * 1000 / calls if calls else 0.
