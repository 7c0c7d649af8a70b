"""MOT: the per-frame orchestrator (API of fastmot/mot.py:25-196).

Same stage schedule as the reference's MOT.step (mot.py:125-168) -- detector enqueue || KLT,
extractor enqueue || Kalman, then association -- but every stage is a set of kernels on its own HIP
stream of one shared device context, and the frame is uploaded once per step (or is already
resident: pass a detector.DeviceFrame)."""
from types import SimpleNamespace
from enum import Enum
import logging
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from .detector import SSDDetector, YOLODetector, PublicDetector, bind_frame
from .feature_extractor import FeatureExtractor
from .tracker import MultiTracker
from .flow import Flow
from .utils import Profiler
from .utils.visualization import Visualizer

LOGGER = logging.getLogger(__name__)
_NATIVE_FLOW = True      # tests flip this to compare the native prediction worker with the Python-thread path


class _NativeFlowJob:
    """Future-like handle of MultiTracker.predict_async (same interface as the thread-pool future it replaces)."""

    def __init__(self, tracker, frame):
        self._tracker = tracker
        self._job = tracker.predict_async(frame)

    def result(self):
        job, self._job = self._job, None
        if job is not None:
            with Profiler('track'):
                self._tracker.predict_finish(job)


class DetectorType(Enum):
    SSD = 0
    YOLO = 1
    PUBLIC = 2


class MOT:
    def __init__(self, size,
                 detector_type='YOLO',
                 detector_frame_skip=5,
                 class_ids=(1,),
                 ssd_detector_cfg=None,
                 yolo_detector_cfg=None,
                 public_detector_cfg=None,
                 feature_extractor_cfgs=None,
                 tracker_cfg=None,
                 visualizer_cfg=None,
                 draw=False):
        """Top level module that integrates detection, feature extraction and tracking
        (parameters: fastmot/mot.py:37-67).  `draw=True` renders the overlays of `visualizer_cfg` onto every
        frame handed to `step` as a host ndarray, in place, after tracking (mot.py:166-167,191-196)."""
        self.size = size
        self.detector_type = DetectorType[detector_type.upper()]
        assert detector_frame_skip >= 1
        self.detector_frame_skip = detector_frame_skip
        self.class_ids = tuple(np.unique(class_ids))
        self.draw = draw

        if ssd_detector_cfg is None:
            ssd_detector_cfg = SimpleNamespace()
        if yolo_detector_cfg is None:
            yolo_detector_cfg = SimpleNamespace()
        if public_detector_cfg is None:
            public_detector_cfg = SimpleNamespace()
        if feature_extractor_cfgs is None:
            feature_extractor_cfgs = (SimpleNamespace(),)
        if tracker_cfg is None:
            tracker_cfg = SimpleNamespace()
        if visualizer_cfg is None:
            visualizer_cfg = SimpleNamespace()
        if len(feature_extractor_cfgs) != len(class_ids):
            raise ValueError('Number of feature extractors must match length of class IDs')
        self.visualizer = Visualizer(**vars(visualizer_cfg))

        LOGGER.info('Loading detector model...')
        if self.detector_type == DetectorType.SSD:
            self.detector = SSDDetector(self.size, self.class_ids, **vars(ssd_detector_cfg))
        elif self.detector_type == DetectorType.YOLO:
            self.detector = YOLODetector(self.size, self.class_ids, **vars(yolo_detector_cfg))
        elif self.detector_type == DetectorType.PUBLIC:
            self.detector = PublicDetector(self.size, self.class_ids, self.detector_frame_skip,
                                           **vars(public_detector_cfg))

        LOGGER.info('Loading feature extractor models...')
        self.extractors = [FeatureExtractor(size=self.size, resident=(i == 0), **vars(cfg))
                           for i, cfg in enumerate(feature_extractor_cfgs)]
        self.tracker = MultiTracker(self.size, self.extractors[0].metric, **vars(tracker_cfg))
        self.frame_count = 0
        self._next_frame = None
        # KLT + Kalman run on a second host thread while this one drives detector -> ReID network (the
        # C-ABI calls release the GIL; the stages use separate HIP streams and share no state)
        self._flow_thread = ThreadPoolExecutor(max_workers=1, thread_name_prefix='fastmot-flow',
                                               initializer=self.tracker.ctx.bind_thread)

    def visible_tracks(self):
        """Confirmed and active tracks (iterator of Track)."""
        return (track for track in self.tracker.tracks.values()
                if track.confirmed and track.active)

    def reset(self, cap_dt):
        """Resets multiple object tracker. Must be called before `step`."""
        self.frame_count = 0
        self.tracker.reset(cap_dt)

    def step(self, frame, next_frame=None):
        """Runs multiple object tracker on the next frame (ndarray HxWx3 uint8 BGR, or a
        detector.DeviceFrame that is already resident on the GPU).

        next_frame (optional, not in the reference): the frame the following `step` will receive, when
        the caller already has it (file sources, a capture queue).  The detector network is then started
        on it right behind this frame's own pass, so that it overlaps this frame's KLT, ReID and association
        stages; results are unchanged (the detector is stateless), per-frame latency too."""
        ctx = self.tracker.ctx
        bind_frame(ctx, frame, self.size, begin_step=True)
        ctx.in_step = True
        self._next_frame = next_frame
        try:
            self._step(frame)
        finally:
            ctx.in_step = False
            self._next_frame = None
        if self.draw:
            self._draw(frame, self._last_detections)
        self.frame_count += 1

    def _prefetch_next(self):
        nxt = self._next_frame
        if nxt is not None and (self.frame_count + 1) % self.detector_frame_skip == 0:
            self._next_frame = None
            self.detector.prefetch(nxt)

    def _step(self, frame):
        self._last_detections = []          # what _draw shows: this frame's detections, none on skipped frames
        if self.frame_count == 0:
            detections = self._last_detections = self.detector(frame)
            self._prefetch_next()
            self.tracker.init(frame, detections)
        elif self.frame_count % self.detector_frame_skip == 0:
            with Profiler('preproc'):
                self.detector.detect_async(frame)

            # Same stages as mot.py:138-161.  The reference overlaps its CPU optical flow with the
            # asynchronous TensorRT detector and its Kalman step with the ReID network; here the whole
            # KLT + Kalman chain (device pyramid / keypoints / LK, host RANSAC, Kalman launch) runs on a
            # second host thread and its own HIP streams, so the critical path of a step is
            # detector -> ReID network -> association.  The stages are independent exactly as in the
            # reference, so the results are identical.
            # next_frame known: its upload and detector pass are queued right behind this frame's pass (the detector
            # stream never idles; results are collected in order, detect.hip) -- and BEFORE the KLT job is started: the
            # detector chain is the longest of a step, and with the worker thread's launches in front of it the copy
            # and the pass reached the GPU late and at varying times (779 +- 70 -> 872 +- 25 frames/s over 8 runs each)
            self._prefetch_next()
            native = _NATIVE_FLOW and type(self.tracker.flow) is Flow      # (tests script the flow with a fake)
            if native:
                # KLT + Kalman on the library's worker thread: marshalled here, scattered in predict_finish -- no second
                # Python thread competing for the interpreter lock (fastmot_hip.h: fm_track_predict_async)
                flow_done = _NativeFlowJob(self.tracker, frame)
            else:
                flow_done = self._flow_thread.submit(self._flow_and_kalman, frame)
            try:
                with Profiler('detect'):
                    detections = self._last_detections = self.detector.postprocess()

                with Profiler('extract'):
                    if len(self.extractors) == 1:
                        self.extractors[0].extract_async(frame, detections.tlbr)
                    else:
                        # one extractor per class id (mot.py:147-157); _split_bboxes_by_cls keeps the
                        # reference's bisect_right, quirk included
                        cls_bboxes = self._split_bboxes_by_cls(detections.tlbr, detections.label, self.class_ids)
                        for extractor, bboxes in zip(self.extractors, cls_bboxes):
                            extractor.extract_async(frame, bboxes)
                    self.tracker.prepare_detections(detections)
                    # the embedding-independent part of the association (track grouping, cost-matrix row order,
                    # packed launch arguments) runs while the ReID network is still busy: it only needs the Kalman
                    # step, i.e. the KLT thread, to have finished
                    flow_done.result()
                    # exactly one extractor holds all of this frame's boxes (always so with one class; with several,
                    # whenever _split_bboxes_by_cls hands every box to the first): the pairwise-cost kernel of the
                    # association is enqueued now, behind the ReID network on the device (tracker.update_begin)
                    busy = [e for e in self.extractors if getattr(e, 'last_num_features', 0)]
                    in_flight = len(busy) == 1 and busy[0].last_num_features == len(detections) and \
                        getattr(busy[0], '_pending', False)
                    pre = self.tracker.update_begin(detections, embeddings_in_flight=in_flight)
                    if len(self.extractors) == 1:
                        embeddings = self.extractors[0].postprocess()
                    else:
                        parts = [extractor.postprocess() for extractor in self.extractors]
                        filled = [p for p in parts if len(p)]
                        # (a single non-empty part is passed through as it is: the association kernels then
                        # read the device-resident copy; the reference's np.concatenate with its float64 empty
                        # parts only changes the dtype of the same values)
                        embeddings = filled[0] if len(filled) == 1 else np.concatenate(parts)
            finally:
                flow_done.result()

            with Profiler('assoc'):
                self.tracker.update(self.frame_count, detections, embeddings, pre=pre)
        else:
            self._prefetch_next()
            with Profiler('track'):
                self.tracker.track(frame)

    @staticmethod
    def _bisect_right(arr, val, left=0):
        """fastmot/utils/numba.py:43-53, verbatim semantics: the test is `arr[mid] >= val` (not `>`), so on
        the ascending label array every probe moves `left` -- the search returns len(arr) whenever
        val <= arr[left:] (always true for the first class id, the labels being restricted to class_ids)."""
        right = len(arr)
        while left < right:
            mid = left + (right - left) // 2
            if arr[mid] >= val:
                left = mid + 1
            else:
                right = mid
        return left

    @classmethod
    def _split_bboxes_by_cls(cls, bboxes, labels, class_ids):
        """mot.py:180-189.  With the reference's bisect_right every box goes to the FIRST class's extractor
        and the others receive empty slices; reproduced, not repaired, so that embeddings (and with them the
        association) equal the reference's."""
        cls_bboxes = []
        begin = 0
        labels = np.asarray(labels).tolist()
        for cls_id in class_ids:
            end = cls._bisect_right(labels, cls_id, begin)
            cls_bboxes.append(bboxes[begin:end])
            begin = end
        return cls_bboxes

    def _draw(self, frame, detections):
        if not isinstance(frame, np.ndarray):
            raise TypeError('draw=True needs host frames (ndarray): overlays are rendered on the CPU')
        visible = list(self.visible_tracks())
        self.visualizer.render(frame, visible, detections, self.tracker.klt_bboxes.values(),
                               self.tracker.flow.prev_bg_keypoints, self.tracker.flow.bg_keypoints,
                               caption=f'visible: {len(visible)}')

    def _flow_and_kalman(self, frame):
        with Profiler('track'):
            self.tracker.compute_flow(frame)
            self.tracker.apply_kalman()

    @staticmethod
    def print_timing_info():
        LOGGER.debug('=================Timing Stats=================')
        LOGGER.debug(f"{'track time:':<37}{Profiler.get_avg_millis('track'):>6.3f} ms")
        LOGGER.debug(f"{'preprocess time:':<37}{Profiler.get_avg_millis('preproc'):>6.3f} ms")
        LOGGER.debug(f"{'detect/flow time:':<37}{Profiler.get_avg_millis('detect'):>6.3f} ms")
        LOGGER.debug(f"{'feature extract/kalman filter time:':<37}"
                     f"{Profiler.get_avg_millis('extract'):>6.3f} ms")
        LOGGER.debug(f"{'association time:':<37}{Profiler.get_avg_millis('assoc'):>6.3f} ms")
