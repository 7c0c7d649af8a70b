"""fastmot_amd -- MI355X-native implementation of the FastMOT per-frame hot path.

Same public surface as the reference package (fastmot/__init__.py:1-7): MOT, MultiTracker,
KalmanFilter, Flow, FeatureExtractor, Track (+ models registries).  Importing the package does
not touch the GPU; the first object that needs the device creates the process-wide context and
raises if libfastmot_hip.so or a HIP device is missing (no CPU fallback).
"""
from . import models
from .videoio import VideoIO
from .tracker import MultiTracker
from .kalman_filter import KalmanFilter, MeasType
from .flow import Flow
from .track import Track

__all__ = ['VideoIO', 'MOT', 'FeatureExtractor', 'MultiTracker', 'KalmanFilter', 'MeasType', 'Flow', 'Track', 'models']


def __getattr__(name):
    # MOT / FeatureExtractor pull in the detector + network engine; resolved on first use
    if name == 'MOT':
        from .mot import MOT
        return MOT
    if name == 'FeatureExtractor':
        from .feature_extractor import FeatureExtractor
        return FeatureExtractor
    raise AttributeError(f'module {__name__!r} has no attribute {name!r}')
