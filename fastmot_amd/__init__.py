"""fastmot_amd -- MI355X-native implementation of the FastMOT per-frame hot path.

Same public surface as the reference package (fastmot/__init__.py:1-7): MOT, MultiTracker,
KalmanFilter, Flow, FeatureExtractor, Track (+ models registries).  Importing the package does
not touch the GPU; the first object that needs the device creates the process-wide context and
raises if libfastmot_hip.so or a HIP device is missing (no CPU fallback).
"""
from .tracker import MultiTracker
from .kalman_filter import KalmanFilter, MeasType
from .flow import Flow
from .track import Track

__all__ = ['MultiTracker', 'KalmanFilter', 'MeasType', 'Flow', 'Track']
