"""Detectors (API of fastmot/detector.py:18-431).

YOLODetector keeps the reference's two-phase protocol (`detect_async(frame)` / `postprocess()`)
but every stage between the frame and the final detections runs on the GPU (detect.hip): the
frame is uploaded once per step and shared with the ReID extractor and KLT; candidates never
leave the device.  PublicDetector (MOTChallenge det.txt) is kept as the detector-disabled path.
SSDDetector (TensorFlow UFF, TensorRT < 8 only) is out of scope.
"""
from collections import defaultdict
from pathlib import Path
import abc
import configparser

import numpy as np

from . import _lib, models
from .engine import HipNet, NET_DETECTOR
from .runtime import get_context

DET_DTYPE = _lib.DET_DTYPE


class DeviceFrame:
    """Handle of a frame that is already resident in the device frame ring (bench / pipelines that
    decode on the GPU).  MOT.step accepts it wherever an ndarray frame is accepted."""

    def __init__(self, index):
        self.index = index


def bind_frame(ctx, frame, size, begin_step=False):
    """Makes `frame` the current device frame.  MOT.step binds once per step (`begin_step=True`);
    the stages it calls with the same frame object then reuse the resident copy.  Stand-alone
    stage calls (outside MOT.step) always upload."""
    if getattr(ctx, 'frame_size', None) != tuple(size):
        ctx.frame_configure(size[0], size[1], getattr(ctx, 'ring_size', 0))
    if not begin_step and getattr(ctx, 'in_step', False) and getattr(ctx, 'bound_frame', None) is frame:
        return
    if getattr(ctx, 'next_frame', None) is frame:
        # the frame was prefetched (prefetch_frame): it is already on the device
        ctx.frame_promote_next()
        ctx.next_frame = None
        ctx.bound_frame = frame
        return
    ctx.next_frame = None
    if isinstance(frame, DeviceFrame):
        ctx.frame_ring_select(frame.index)
    else:
        ctx.frame_upload(frame)
    ctx.bound_frame = frame


def prefetch_frame(ctx, frame, size):
    """Puts the NEXT frame on the device (second upload slot / ring) without disturbing the current one;
    the next bind_frame(frame) promotes it instead of uploading again."""
    if getattr(ctx, 'frame_size', None) != tuple(size):
        ctx.frame_configure(size[0], size[1], getattr(ctx, 'ring_size', 0))
    if isinstance(frame, DeviceFrame):
        ctx.frame_ring_select_next(frame.index)
    else:
        ctx.frame_upload_next(frame)
    ctx.next_frame = frame


class Detector(abc.ABC):
    @abc.abstractmethod
    def __init__(self, size):
        self.size = size

    def __call__(self, frame):
        """Detect objects synchronously."""
        self.detect_async(frame)
        return self.postprocess()

    @abc.abstractmethod
    def detect_async(self, frame):
        raise NotImplementedError

    def prefetch(self, frame):
        """Optional: start detecting on the NEXT frame while the current one is still being tracked
        (MOT.step(frame, next_frame)).  The following detect_async(frame) is then a no-op."""

    @abc.abstractmethod
    def postprocess(self):
        raise NotImplementedError


class SSDDetector(Detector):
    def __init__(self, size, class_ids, **kwargs):
        raise NotImplementedError('SSD (TensorFlow UFF / TensorRT < 8) is out of scope of the MI355X '
                                  'hot path; use detector_type YOLO or PUBLIC')

    def detect_async(self, frame):
        raise NotImplementedError

    def postprocess(self):
        raise NotImplementedError


class YOLODetector(Detector):
    def __init__(self, size,
                 class_ids,
                 model='YOLOv4',
                 conf_thresh=0.25,
                 nms_thresh=0.5,
                 max_area=800000,
                 min_aspect_ratio=1.2,
                 weights=None,
                 max_candidates=8192,
                 reuse_buffers=True):
        """An object detector for YOLO models; parameters as fastmot/detector.py:221-253
        (`weights`: optional weight source for the layer table, default seeded random;
        `max_candidates`: capacity of the on-device candidate list)."""
        super().__init__(size)
        self.model = models.YOLO.get_model(model)
        assert 0 <= conf_thresh <= 1
        self.conf_thresh = conf_thresh
        assert 0 <= nms_thresh <= 1
        self.nms_thresh = nms_thresh
        assert max_area >= 0
        self.max_area = max_area
        assert min_aspect_ratio >= 0
        self.min_aspect_ratio = min_aspect_ratio

        self.label_mask = np.zeros(self.model.NUM_CLASSES, dtype=np.bool_)
        try:
            self.label_mask[tuple(class_ids),] = True
        except IndexError as err:
            raise ValueError('Unsupported class IDs') from err

        self.ctx = get_context()
        self._prefetched = None
        self.graph, self.heads = self.model.build_graph(weights)
        self.backend = HipNet(self.ctx, NET_DETECTOR, self.graph, 1, reuse_buffers=reuse_buffers)
        self.roi, self.upscaled_sz, self.bbox_offset = self._create_letterbox()
        self._configure(max_candidates)

    def _create_letterbox(self):
        """ROI of the network input that receives the resized frame, the frame size the boxes are
        scaled to and their offset (fastmot/detector.py:302-320)."""
        src_size = np.array(self.size)
        dst_size = np.array(self.model.INPUT_SHAPE[:0:-1])
        if self.model.LETTERBOX:
            scale_factor = min(dst_size / src_size)
            scaled_size = np.rint(src_size * scale_factor).astype(int)
            img_offset = ((dst_size - scaled_size) / 2).astype(int)
            roi = (int(img_offset[0]), int(img_offset[1]), int(scaled_size[0]), int(scaled_size[1]))
            upscaled_sz = np.rint(dst_size / scale_factor).astype(int)
            bbox_offset = (upscaled_sz - src_size) / 2
        else:
            roi = (0, 0, int(dst_size[0]), int(dst_size[1]))
            upscaled_sz = src_size
            bbox_offset = np.zeros(2)
        return roi, upscaled_sz, bbox_offset

    def _configure(self, max_candidates):
        m = self.model
        cfg = _lib.YoloCfg()
        cfg.in_w, cfg.in_h = m.INPUT_SHAPE[2], m.INPUT_SHAPE[1]
        cfg.roi_x, cfg.roi_y, cfg.roi_w, cfg.roi_h = self.roi
        cfg.input_tensor = self.graph.input.tid
        cfg.n_heads = len(self.heads)
        for i, head in enumerate(self.heads):
            cfg.head_tensor[i] = head.tid
            cfg.grid_w[i] = m.INPUT_SHAPE[2] // m.LAYER_FACTORS[i]
            cfg.grid_h[i] = m.INPUT_SHAPE[1] // m.LAYER_FACTORS[i]
            cfg.n_anchors[i] = len(m.ANCHORS[i]) // 2
            for j, a in enumerate(m.ANCHORS[i]):
                cfg.anchors[i][j] = a
            cfg.scale_xy[i] = m.SCALES[i]
        cfg.num_classes = m.NUM_CLASSES
        cfg.new_coords = int(m.NEW_COORDS)
        for i, v in enumerate(self.label_mask):
            cfg.label_mask[i] = int(v)
        cfg.conf_thresh, cfg.nms_thresh = self.conf_thresh, self.nms_thresh
        cfg.max_area, cfg.min_aspect_ratio = self.max_area, self.min_aspect_ratio
        cfg.size[0], cfg.size[1] = float(self.upscaled_sz[0]), float(self.upscaled_sz[1])
        cfg.offset[0], cfg.offset[1] = float(self.bbox_offset[0]), float(self.bbox_offset[1])
        cfg.max_candidates = max_candidates
        self._cfg = cfg
        self.ctx.detect_configure(cfg)

    def detect_async(self, frame):
        """Detects objects asynchronously (preprocess + network + decode + NMS enqueued)."""
        if self._prefetched is frame and frame is not None:
            self._prefetched = None              # already enqueued by prefetch()
            bind_frame(self.ctx, frame, self.size)
            return
        self._prefetched = None
        bind_frame(self.ctx, frame, self.size)
        self.ctx.detect_async()

    def prefetch(self, frame):
        prefetch_frame(self.ctx, frame, self.size)
        self.ctx.detect_async_next()
        self._prefetched = frame

    def postprocess(self):
        """Synchronizes and returns a record array of detections (DET_DTYPE), sorted in ascending
        order by class ID.  This API should be called after `detect_async`."""
        return self.ctx.detect_sync()


class PublicDetector(Detector):
    def __init__(self, size,
                 class_ids,
                 frame_skip,
                 sequence_path=None,
                 conf_thresh=0.5,
                 max_area=800000):
        """MOT Challenge public detections (fastmot/detector.py:368-431): reads
        <sequence_path>/det/det.txt and seqinfo.ini; boxes are rescaled to `size`."""
        super().__init__(size)
        assert tuple(class_ids) == (1,)
        self.frame_skip = frame_skip
        assert sequence_path is not None
        self.seq_root = Path(sequence_path)
        if not self.seq_root.is_absolute():
            self.seq_root = Path(__file__).parents[1] / sequence_path
        assert 0 <= conf_thresh <= 1
        self.conf_thresh = conf_thresh
        assert max_area >= 0
        self.max_area = max_area

        assert self.seq_root.exists()
        seqinfo = configparser.ConfigParser()
        seqinfo.read(self.seq_root / 'seqinfo.ini')
        self.seq_size = (int(seqinfo['Sequence']['imWidth']), int(seqinfo['Sequence']['imHeight']))

        self.detections = defaultdict(list)
        self.frame_id = 0

        rows = np.loadtxt(self.seq_root / 'det' / 'det.txt', delimiter=',', ndmin=2)
        scale = np.array(self.size, float) / np.array(self.seq_size, float)
        for row in rows:
            frame_id = int(row[0]) - 1
            x, y, w, h = (float(v) for v in row[2:6])
            tlbr = np.array([round(x), round(y), round(x + w - 1.), round(y + h - 1.)], float)
            conf, label = 1.0, 1      # the reference ignores det.txt's score column
            tlbr[:2] = tlbr[:2] / self.seq_size * self.size
            tlbr[2:] = tlbr[2:] / self.seq_size * self.size
            tlbr = np.rint(tlbr)
            bw, bh = tlbr[2] - tlbr[0] + 1, tlbr[3] - tlbr[1] + 1
            area = 0. if bw <= 0 or bh <= 0 else bw * bh
            if conf >= self.conf_thresh and area <= self.max_area:
                self.detections[frame_id].append((tlbr, label, conf))
        del scale

    def detect_async(self, frame):
        pass

    def postprocess(self):
        detections = np.array(self.detections[self.frame_id], DET_DTYPE).view(np.recarray)
        self.frame_id += self.frame_skip
        return detections
