"""Detectors (API of fastmot/detector.py:18-431).

YOLODetector keeps the reference's two-phase protocol (`detect_async(frame)` / `postprocess()`)
but every stage between the frame and the final detections runs on the GPU (detect.hip): the
frame is uploaded once per step and shared with the ReID extractor and KLT; candidates never
leave the device.  PublicDetector (MOTChallenge det.txt) is kept as the detector-disabled path.
SSDDetector keeps the reference's tiling / normalisation / filtering / cross-tile merging stages around
an inference callable (the SSD networks themselves need a TensorFlow graph reader, see models/ssd.py).
"""
from collections import defaultdict
from pathlib import Path
import abc
import configparser

import numpy as np

from . import _lib, models
from .engine import HipNet, NET_DETECTOR
from .runtime import get_context
from .utils.setorder import IntSet

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
    def __init__(self, size,
                 class_ids,
                 model='SSDInceptionV2',
                 tile_overlap=0.25,
                 tiling_grid=(4, 2),
                 conf_thresh=0.5,
                 merge_thresh=0.6,
                 max_area=120000,
                 backend=None):
        """Tiled SSD detector; parameters as fastmot/detector.py:46-75.  The frame is resized to the tiling
        region, cut into `tiling_grid` overlapping tiles of the network's input size, normalised to [-1, 1]
        (RGB, CHW) and handed to `backend(batch) -> det_out` as one batch; `det_out` is the output of the SSD
        engine with its NMS stage: TOPK rows (image id, label, conf, xmin, ymin, xmax, ymax in tile fractions)
        per tile, sorted by confidence.  `backend` (not in the reference: there it is the TensorRT engine built
        from the model's .pb file) is required because the SSD networks cannot be built here (models/ssd.py).
        These stages run on the host, as the reference's Numba versions do."""
        super().__init__(size)
        self.model = models.SSD.get_model(model)
        assert 0 <= tile_overlap <= 1
        self.tile_overlap = tile_overlap
        assert tiling_grid[0] >= 1 and tiling_grid[1] >= 1
        self.tiling_grid = tiling_grid
        assert 0 <= conf_thresh <= 1
        self.conf_thresh = conf_thresh
        assert 0 <= merge_thresh <= 1
        self.merge_thresh = merge_thresh
        assert max_area >= 0
        self.max_area = max_area

        self.label_mask = np.zeros(self.model.NUM_CLASSES, dtype=np.bool_)
        try:
            self.label_mask[tuple(class_ids),] = True
        except IndexError as err:
            raise ValueError('Unsupported class IDs') from err

        self.batch_size = int(np.prod(self.tiling_grid))
        self.tiles, self.tiling_region_sz = self._generate_tiles()
        self.scale_factor = tuple(np.array(self.size) / self.tiling_region_sz)
        if backend is None:
            self.model.build_graph()            # raises: explains what is missing
        self.backend = backend
        self.inp_handle = np.empty((self.batch_size, *self.model.INPUT_SHAPE), np.float32)
        self._det_out = None

    def detect_async(self, frame):
        """Preprocesses the frame and runs the inference callable on the batch of tiles."""
        from .videoio import resize_bgr
        if not isinstance(frame, np.ndarray):
            raise TypeError('SSDDetector works on host frames (ndarray)')
        self.normalize(resize_bgr(frame, self.tiling_region_sz), self.tiles, self.inp_handle)
        self._det_out = np.asarray(self.backend(self.inp_handle), np.float32).reshape(-1)

    def postprocess(self):
        """Returns a record array of detections (DET_DTYPE) sorted by class ID, duplicates across tiles merged."""
        assert self._det_out is not None, 'postprocess() without detect_async()'
        dets, tile_ids = self.filter_dets(self._det_out, self.tiles, self.model.TOPK, self.label_mask,
                                          self.max_area, self.conf_thresh, self.scale_factor)
        self._det_out = None
        return self.merge_dets(dets, tile_ids, self.batch_size, self.merge_thresh)

    def _generate_tiles(self):
        """Tile rectangles (tlbr, inclusive) in the tiling region and the region's size (detector.py:122-130)."""
        tile_wh = np.array(self.model.INPUT_SHAPE[:0:-1], float)
        grid = np.array(self.tiling_grid)
        step = (1 - self.tile_overlap) * tile_wh
        region = np.rint((grid - 1) * step + tile_wh).astype(int)
        tiles = []
        for row in range(grid[1]):
            for col in range(grid[0]):
                x, y = float(col * step[0]), float(row * step[1])
                tiles.append([round(x, 0), round(y, 0), round(x + tile_wh[0] - 1., 0), round(y + tile_wh[1] - 1., 0)])
        return np.array(tiles), tuple(region)

    @staticmethod
    def normalize(frame, tiles, out):
        """Tile crops (int-truncated, clamped at 0, inclusive corners) -> RGB, CHW, x * 2/255 - 1 (float32)."""
        t = np.maximum(tiles.astype(np.int_), 0)
        for i, (x0, y0, x1, y1) in enumerate(t):
            crop = frame[y0:y1 + 1, x0:x1 + 1, ::-1]
            out[i] = crop.transpose(2, 0, 1) * (2 / 255.) - 1.

    @staticmethod
    def filter_dets(det_out, tiles, topk, label_mask, max_area, thresh, scale_factor):
        """Engine rows -> frame coordinates (detector.py:161-185): per tile the rows up to the first one below
        `thresh`, classes of `label_mask`, tile fractions scaled by the tile size, shifted by the tile origin,
        scaled to the frame, rounded half-to-even; 0 < area <= max_area."""
        rows = np.asarray(det_out, np.float32).reshape(len(tiles), topk, 7)
        boxes, labels, confs, tile_ids = [], [], [], []
        for ti, tile in enumerate(tiles):
            conf = rows[ti, :, 2]
            below = np.flatnonzero(conf < thresh)
            n = int(below[0]) if len(below) else topk
            if n == 0:
                continue
            r = rows[ti, :n]
            label = r[:, 1].astype(int)
            w, h = tile[2] - tile[0] + 1, tile[3] - tile[1] + 1
            tlbr = np.rint(np.stack([(r[:, 3].astype(float) * w + tile[0]) * scale_factor[0],
                                     (r[:, 4].astype(float) * h + tile[1]) * scale_factor[1],
                                     (r[:, 5].astype(float) * w + tile[0]) * scale_factor[0],
                                     (r[:, 6].astype(float) * h + tile[1]) * scale_factor[1]], 1))
            bw, bh = tlbr[:, 2] - tlbr[:, 0] + 1, tlbr[:, 3] - tlbr[:, 1] + 1
            area = np.where((bw <= 0) | (bh <= 0), 0., bw * bh)
            ok = label_mask[label] & (area > 0) & (area <= max_area)
            boxes.append(tlbr[ok]); labels.append(label[ok]); confs.append(r[ok, 2]); tile_ids.append(np.full(ok.sum(), ti))
        dets = np.zeros(sum(len(b) for b in boxes), DET_DTYPE).view(np.recarray)
        if len(dets):
            dets.tlbr, dets.label, dets.conf = np.concatenate(boxes), np.concatenate(labels), np.concatenate(confs)
        return dets, (np.concatenate(tile_ids) if tile_ids else np.zeros(0, int))

    @staticmethod
    def merge_dets(dets, tile_ids, num_tile, thresh):
        """Merges the detections of one object seen by several tiles (detector.py:132-139,187-217).  A detection
        links to every detection of the same class in ANOTHER tile whose intersection-over-minimum is >= thresh
        and exceeds every earlier candidate of that tile (scan in index order: running maxima are all kept);
        connected groups collapse into their first member (enclosing box, maximum confidence); the survivors
        are returned in the iteration order of the reference's (Numba) set, then argsorted by class."""
        n = len(dets)
        if n == 0:
            return dets
        box = np.array(dets.tlbr, float)
        bw, bh = box[:, 2] - box[:, 0] + 1, box[:, 3] - box[:, 1] + 1
        area = np.where((bw <= 0) | (bh <= 0), 0., bw * bh)
        iw = np.minimum(box[:, None, 2], box[None, :, 2]) - np.maximum(box[:, None, 0], box[None, :, 0]) + 1
        ih = np.minimum(box[:, None, 3], box[None, :, 3]) - np.maximum(box[:, None, 1], box[None, :, 1]) + 1
        with np.errstate(divide='ignore', invalid='ignore'):
            iom = np.where((iw <= 0) | (ih <= 0), 0., iw * ih / np.minimum(area[:, None], area[None, :]))
        linkable = (tile_ids[:, None] != tile_ids[None, :]) & (dets.label[:, None] == dets.label[None, :]) & (iom >= thresh)
        links = []
        for i in range(n):
            best = np.zeros(num_tile)
            mine = []
            for j in np.flatnonzero(linkable[i]):
                if iom[i, j] > best[tile_ids[j]]:
                    best[tile_ids[j]] = iom[i, j]
                    mine.append(int(j))
            links.append(mine)
        seen = np.zeros(n, bool)
        keep = IntSet(n)            # `keep = set(range(len(dets)))` inside @njit: Numba's set, its iteration order
        out = dets.copy().view(np.recarray)
        for i in range(n):
            if not links[i] or seen[i]:
                continue
            seen[i] = True
            todo, group = [i], []
            while todo:
                for j in links[todo.pop()]:
                    if not seen[j]:
                        seen[j] = True
                        group.append(j)
                        todo.append(j)
            for k in group:
                out.tlbr[i] = np.concatenate([np.minimum(out.tlbr[i, :2], out.tlbr[k, :2]),
                                              np.maximum(out.tlbr[i, 2:], out.tlbr[k, 2:])])
                out.conf[i] = max(out.conf[i], out.conf[k])
                keep.discard(k)
        out = out[np.array(list(keep), np.int64)]
        return out[np.argsort(out.label)].view(np.recarray)


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
        if self._prefetched is not None:
            # the caller announced another frame than the one it passes now: collect and drop that pass, so that
            # postprocess() returns THIS frame's detections (passes are collected in the order they were enqueued)
            self._prefetched = None
            self.ctx.detect_sync()
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
