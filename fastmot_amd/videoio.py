"""VideoIO (API of fastmot/videoio.py:24-277) without OpenCV / GStreamer: threaded capture into a bounded
queue with the reference's semantics (file sources block when the buffer is full, live sources drop),
`cap_dt`, `read()`, `write()`, `release()`.

Sources this implementation can decode:
  * image sequences  'dir/%06d.jpg' (any format Pillow reads; the MOTChallenge layout)  -> Protocol.IMAGE
  * raw frame stacks '*.npy' ([N, H, W, 3] uint8 BGR, memory-mapped)                     -> Protocol.VIDEO
Video containers, cameras and network streams need a decoder this image does not have; they raise
NotImplementedError with the URI.  Outputs: image sequence ('out/%06d.png') or '*.npy'.

Frames are BGR uint8 like cv2's; a source whose size differs from `size` is resized with cv2.resize's
INTER_LINEAR arithmetic (imgproc/resize.cpp: 11-bit fixed-point coefficients, exact 2x decimation =
INTER_AREA) so that downstream results do not depend on which VideoIO produced the frame."""
from collections import deque
from enum import Enum
from pathlib import Path
from urllib.parse import urlparse
import logging
import threading

import numpy as np

LOGGER = logging.getLogger(__name__)


class Protocol(Enum):
    IMAGE = 0
    VIDEO = 1
    CSI = 2
    V4L2 = 3
    RTSP = 4
    HTTP = 5


def _lin_coef(dsize, ssize):
    scale = ssize / dsize
    fx = ((np.arange(dsize) + 0.5) * scale - 0.5).astype(np.float32)
    sx = np.floor(fx).astype(np.int32)
    fx = fx - sx.astype(np.float32)
    lo, hi = sx < 0, sx >= ssize - 1
    fx[lo | hi] = 0
    sx[lo] = 0
    sx[hi] = ssize - 1
    a1 = np.rint(fx * np.float32(2048)).astype(np.int32)
    a0 = np.rint((np.float32(1) - fx) * np.float32(2048)).astype(np.int32)
    return sx, np.minimum(sx + 1, ssize - 1), a0, a1


def resize_bgr(img, size):
    """cv2.resize(img, size) for uint8 images, INTER_LINEAR (fixed point, see module docstring)."""
    dw, dh = size
    sh, sw = img.shape[:2]
    if (sw, sh) == (dw, dh):
        return img
    src = img.astype(np.int32)
    if sw == 2 * dw and sh == 2 * dh:
        return ((src[0::2, 0::2] + src[0::2, 1::2] + src[1::2, 0::2] + src[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    x0, x1, ax0, ax1 = _lin_coef(dw, sw)
    y0, y1, ay0, ay1 = _lin_coef(dh, sh)
    rows = src[:, x0] * ax0[None, :, None] + src[:, x1] * ax1[None, :, None]
    s0, s1 = rows[y0] >> 4, rows[y1] >> 4
    out = (((ay0[:, None, None] * s0) >> 16) + ((ay1[:, None, None] * s1) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


class _ImageSequence:
    def __init__(self, pattern):
        from PIL import Image
        self._open = Image.open
        self.pattern = pattern
        self.index = 0 if Path(pattern % 0).exists() else 1
        if not Path(pattern % self.index).exists():
            raise RuntimeError('Unable to read video stream')

    def read(self):
        path = Path(self.pattern % self.index)
        if not path.exists():
            return None
        self.index += 1
        with self._open(path) as im:
            rgb = np.asarray(im.convert('RGB'))
        return np.ascontiguousarray(rgb[:, :, ::-1])


class _FrameStack:
    def __init__(self, path):
        self.frames = np.load(path, mmap_mode='r')
        if self.frames.ndim != 4 or self.frames.shape[3] != 3 or self.frames.dtype != np.uint8:
            raise RuntimeError('Unable to read video stream: expected a [N, H, W, 3] uint8 array')
        self.index = 0

    def read(self):
        if self.index >= len(self.frames):
            return None
        self.index += 1
        return np.array(self.frames[self.index - 1])


class VideoIO:
    def __init__(self, size, input_uri,
                 output_uri=None,
                 resolution=(1920, 1080),
                 frame_rate=30,
                 buffer_size=10,
                 proc_fps=30):
        """Parameters as fastmot/videoio.py:25-58."""
        self.size = tuple(size)
        self.input_uri = input_uri
        self.output_uri = output_uri
        self.resolution = resolution
        assert frame_rate > 0
        self.frame_rate = frame_rate
        assert buffer_size >= 1
        self.buffer_size = buffer_size
        assert proc_fps > 0
        self.proc_fps = proc_fps

        self.protocol = self._parse_uri(self.input_uri)
        self.is_live = self.protocol != Protocol.IMAGE and self.protocol != Protocol.VIDEO
        if self.protocol == Protocol.IMAGE:
            self.source = _ImageSequence(self.input_uri)
        elif self.protocol == Protocol.VIDEO and str(self.input_uri).endswith('.npy'):
            self.source = _FrameStack(self.input_uri)
        else:
            raise NotImplementedError(f'{self.input_uri}: {self.protocol.name} sources need a video decoder '
                                      '(supported here: image sequences and .npy frame stacks)')

        self.frame_queue = deque([], maxlen=self.buffer_size)
        self.cond = threading.Condition()
        self.exit_event = threading.Event()
        self.cap_thread = threading.Thread(target=self._capture_frames, daemon=True)

        frame = self.source.read()
        if frame is None:
            raise RuntimeError('Unable to read video stream')
        self.frame_queue.append(frame)

        height, width = frame.shape[:2]
        self.resolution = (width, height)
        self.cap_fps = self.frame_rate          # neither source kind carries a frame rate
        self.do_resize = (width, height) != self.size
        LOGGER.info('%dx%d stream @ %d FPS', width, height, self.cap_fps)

        self._written = 0
        self._stack = None
        if self.output_uri is not None:
            Path(self.output_uri).parent.mkdir(parents=True, exist_ok=True)
            if str(self.output_uri).endswith('.npy'):
                self._stack = []
            elif '%' not in str(self.output_uri):
                raise NotImplementedError(f'{self.output_uri}: video encoding needs an encoder '
                                          "(supported here: image sequences 'dir/%06d.png' and .npy)")

    @property
    def cap_dt(self):
        # limit capture interval at processing latency for live sources
        return 1 / min(self.cap_fps, self.proc_fps) if self.is_live else 1 / self.cap_fps

    def start_capture(self):
        """Start capturing from file or device."""
        if not self.cap_thread.is_alive():
            self.cap_thread.start()

    def stop_capture(self):
        """Stop capturing from file or device."""
        with self.cond:
            self.exit_event.set()
            self.cond.notify()
        self.frame_queue.clear()
        if self.cap_thread.is_alive():
            self.cap_thread.join()

    def read(self):
        """Reads the next video frame (None if there are no more frames)."""
        with self.cond:
            while len(self.frame_queue) == 0 and not self.exit_event.is_set():
                self.cond.wait()
            if len(self.frame_queue) == 0 and self.exit_event.is_set():
                return None
            frame = self.frame_queue.popleft()
            self.cond.notify()
        if self.do_resize:
            frame = resize_bgr(frame, self.size)
        return frame

    def write(self, frame):
        """Writes the next video frame."""
        assert self.output_uri is not None
        if self._stack is not None:
            self._stack.append(np.array(frame))
        else:
            from PIL import Image
            Image.fromarray(np.ascontiguousarray(frame[:, :, ::-1])).save(str(self.output_uri) % self._written)
        self._written += 1

    def release(self):
        """Cleans up input and output sources."""
        self.stop_capture()
        if self._stack is not None and self._stack:
            np.save(self.output_uri, np.stack(self._stack))

    def _capture_frames(self):
        while not self.exit_event.is_set():
            frame = self.source.read()
            with self.cond:
                if frame is None:
                    self.exit_event.set()
                    self.cond.notify()
                    break
                # keep unprocessed frames in the buffer for file
                if not self.is_live:
                    while (len(self.frame_queue) == self.buffer_size and
                           not self.exit_event.is_set()):
                        self.cond.wait()
                self.frame_queue.append(frame)
                self.cond.notify()

    @staticmethod
    def _parse_uri(uri):
        result = urlparse(str(uri))
        if result.scheme == 'csi':
            protocol = Protocol.CSI
        elif result.scheme == 'rtsp':
            protocol = Protocol.RTSP
        elif result.scheme == 'http':
            protocol = Protocol.HTTP
        else:
            if '/dev/video' in result.path:
                protocol = Protocol.V4L2
            elif '%' in result.path:
                protocol = Protocol.IMAGE
            else:
                protocol = Protocol.VIDEO
        return protocol
