"""ctypes binding of libfastmot_hip.so (include/fastmot_hip.h).

This is the ONLY native boundary of the package.  There is no CPU fallback: if the
shared library is missing or a call fails, an exception is raised (the reference raises
RuntimeError when its plugin/engine is missing, fastmot/utils/inference.py:50-63).
"""
import ctypes as C
import os
import sys
from pathlib import Path

import numpy as np
from types import SimpleNamespace

LIB_PATH = Path(os.environ.get('FASTMOT_LIB_PATH', Path(__file__).parent / 'libfastmot_hip.so'))

c_int_p = C.POINTER(C.c_int)
_lib = None


class FastMOTHipError(RuntimeError):
    pass


class KFParams(C.Structure):
    _fields_ = [('dt', C.c_double),
                ('std_factor_acc', C.c_double), ('std_offset_acc', C.c_double),
                ('std_factor_det', C.c_double * 2), ('std_factor_klt', C.c_double * 2),
                ('min_std_det', C.c_double * 2), ('min_std_klt', C.c_double * 2),
                ('init_pos_weight', C.c_double), ('init_vel_weight', C.c_double),
                ('vel_coupling', C.c_double), ('vel_half_life', C.c_double)]


def load():
    """Loads the shared library once; raises RuntimeError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise RuntimeError(f'{LIB_PATH} not found: build it with `python -m fastmot_amd.build` '
                           '(hipcc --offload-arch=gfx950); there is no CPU fallback')
    try:
        lib = C.CDLL(str(LIB_PATH), mode=os.RTLD_NOW | os.RTLD_LOCAL)
    except OSError as err:
        raise RuntimeError(f'Unable to load {LIB_PATH}') from err
    lib.fm_last_error.restype = C.c_char_p
    _lib = lib
    return lib


def _ptr(arr, ctype=None):
    """Raw data pointer of a C-contiguous array that the CALLER keeps alive for the duration of the call
    (`arr.ctypes.data_as` would hold a reference but costs 2.4 us per argument; ~60 arguments per frame)."""
    if arr is None:
        return None
    return C.c_void_p(arr.__array_interface__['data'][0])


def _as(arr, dtype, shape=None):
    out = np.ascontiguousarray(arr, dtype=dtype)
    if shape is not None:
        out = out.reshape(shape)
    return out


def check(rc):
    if rc != 0:
        raise FastMOTHipError(f'libfastmot_hip error {rc}: {load().fm_last_error().decode()}')


def device_count():
    n = load().fm_device_count()
    if n < 0:
        raise FastMOTHipError(load().fm_last_error().decode())
    return n


def device_pci_bus_id(device):
    buf = C.create_string_buffer(64)
    check(load().fm_device_pci_bus_id(C.c_int(device), buf, C.c_int(64)))
    return buf.value.decode().lower()


class HipContext:
    """One context = one GPU = one video stream (fm_ctx)."""

    def __init__(self, device=0):
        self.lib = load()
        self._ctx = C.c_void_p()
        check(self.lib.fm_ctx_create(C.c_int(device), C.byref(self._ctx)))
        self.device = device
        self.feat_dim = 512

    def close(self):
        if getattr(self, '_ctx', None) is not None and self._ctx:
            self.lib.fm_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        # at interpreter shutdown the HIP runtime may already be gone: the orderly path is the atexit hook
        # registered by runtime.get_context(), which runs while the runtime is still alive
        if sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._ctx

    def synchronize(self):
        check(self.lib.fm_ctx_synchronize(self._ctx))

    def set_option(self, key, value):
        """Tunables of the context: 'zero_copy_tracks', 'host_lap_elems' (include/fastmot_hip.h)."""
        check(self.lib.fm_ctx_set_option(self._ctx, key.encode(), C.c_int(int(value))))

    def trace_start(self, cap=20000):
        """Arms the library's stage-boundary event trace; -> host CLOCK_MONOTONIC ns of the trace's zero."""
        t0 = C.c_int64(0)
        check(self.lib.fm_trace_start(self._ctx, C.c_int(cap), C.byref(t0)))
        self._trace_cap = cap
        return t0.value

    def trace_read(self):
        """-> (tags int32[n], ms float32[n]): GPU time of every recorded stage boundary since the trace's zero."""
        cap = self._trace_cap
        tags = np.zeros(cap, np.int32)
        ms = np.zeros(cap, np.float32)
        n = C.c_int(0)
        check(self.lib.fm_trace_read(self._ctx, C.c_int(cap), _ptr(tags), _ptr(ms), C.byref(n)))
        return tags[:n.value], ms[:n.value]

    def bind_thread(self):
        """Called once by every additional host thread that drives this context."""
        check(self.lib.fm_ctx_bind_thread(self._ctx))

    def device_info(self):
        buf = C.create_string_buffer(256)
        check(self.lib.fm_device_info(self._ctx, buf, 256))
        # gcnArchName itself contains ':' (gfx950:sramecc+:xnack-) -> parse from the right
        parts = buf.value.decode().split(':')
        return {'name': parts[0], 'arch': ':'.join(parts[1:-3]), 'cus': int(parts[-3]),
                'clock_mhz': int(parts[-2]), 'hbm_bytes': int(parts[-1])}

    # ------------------------------------------------------------------ Kalman
    def kf_configure(self, dt, std_factor_acc, std_offset_acc, std_factor_det, std_factor_klt,
                     min_std_det, min_std_klt, init_pos_weight, init_vel_weight, vel_coupling,
                     vel_half_life):
        p = KFParams(dt, std_factor_acc, std_offset_acc, (C.c_double * 2)(*std_factor_det),
                     (C.c_double * 2)(*std_factor_klt), (C.c_double * 2)(*min_std_det),
                     (C.c_double * 2)(*min_std_klt), init_pos_weight, init_vel_weight, vel_coupling,
                     vel_half_life)
        check(self.lib.fm_kf_configure(self._ctx, C.byref(p)))

    def set_frame_rect(self, tlbr):
        a = _as(tlbr, np.float64, (4,))
        check(self.lib.fm_set_frame_rect(self._ctx, _ptr(a)))

    def trk_create(self, slots, det_tlbr):
        s = _as(slots, np.int32)
        b = _as(det_tlbr, np.float64, (-1, 4))
        assert len(s) == len(b)
        check(self.lib.fm_trk_create(self._ctx, C.c_int(len(s)), _ptr(s), _ptr(b)))

    def trk_step(self, slots, H, klt_tlbr, has_klt, mult):
        s = _as(slots, np.int32)
        n = len(s)
        Hm = _as(H, np.float64, (9,))
        k = _as(klt_tlbr, np.float64, (-1, 4)) if n else np.zeros((0, 4))
        hk = _as(has_klt, np.uint8)
        mu = _as(mult, np.float64)
        assert len(k) == n and len(hk) == n and len(mu) == n
        tlbr = np.empty((n, 4), np.float64)
        lost = np.empty(n, np.uint8)
        check(self.lib.fm_trk_step(self._ctx, C.c_int(n), _ptr(s), _ptr(Hm), _ptr(k), _ptr(hk),
                                   _ptr(mu), _ptr(tlbr), _ptr(lost)))
        return tlbr, lost.astype(bool)

    def trk_step_ops(self, ops, slots, H, klt_tlbr, has_klt, mult):
        s = _as(slots, np.int32)
        n = len(s)
        Hm = _as(H, np.float64, (9,))
        k = _as(klt_tlbr, np.float64, (-1, 4)) if n else np.zeros((0, 4))
        hk = _as(has_klt, np.uint8)
        mu = _as(mult, np.float64)
        tlbr = np.empty((n, 4), np.float64)
        lost = np.empty(n, np.uint8)
        check(self.lib.fm_trk_step_ops(self._ctx, C.c_int(ops), C.c_int(n), _ptr(s), _ptr(Hm), _ptr(k),
                                       _ptr(hk), _ptr(mu), _ptr(tlbr), _ptr(lost)))
        return tlbr, lost.astype(bool)

    def trk_update_det(self, slots, det_tlbr):
        s = _as(slots, np.int32)
        n = len(s)
        b = _as(det_tlbr, np.float64, (-1, 4)) if n else np.zeros((0, 4))
        assert len(b) == n
        tlbr = np.empty((n, 4), np.float64)
        lost = np.empty(n, np.uint8)
        check(self.lib.fm_trk_update_det(self._ctx, C.c_int(n), _ptr(s), _ptr(b), _ptr(tlbr), _ptr(lost)))
        return tlbr, lost.astype(bool)

    def trk_get_state(self, slots):
        s = _as(slots, np.int32)
        mean = np.empty((len(s), 8), np.float64)
        cov = np.empty((len(s), 8, 8), np.float64)
        check(self.lib.fm_trk_get_state(self._ctx, C.c_int(len(s)), _ptr(s), _ptr(mean), _ptr(cov)))
        return mean, cov

    def trk_set_state(self, slots, mean, cov):
        s = _as(slots, np.int32)
        m = _as(mean, np.float64, (-1, 8))
        c = _as(cov, np.float64, (-1, 8, 8))
        assert len(m) == len(s) and len(c) == len(s)
        check(self.lib.fm_trk_set_state(self._ctx, C.c_int(len(s)), _ptr(s), _ptr(m), _ptr(c)))

    def trk_copy_state(self, dst, src):
        check(self.lib.fm_trk_copy_state(self._ctx, C.c_int(dst), C.c_int(src)))

    # ------------------------------------------------------------------ features
    def feat_configure(self, dim):
        check(self.lib.fm_feat_configure(self._ctx, C.c_int(dim)))
        self.feat_dim = dim

    def emb_upload(self, emb):
        if emb is None:
            raise ValueError('emb is None; use emb_use_device(n)')
        e = _as(emb, np.float32)
        if e.ndim != 2 or (len(e) and e.shape[1] != self.feat_dim):
            raise ValueError('embeddings must be [n][feat_dim]')
        check(self.lib.fm_emb_upload(self._ctx, C.c_int(len(e)), _ptr(e)))

    def emb_use_device(self, n):
        check(self.lib.fm_emb_upload(self._ctx, C.c_int(n), None))

    def feat_update(self, slots, emb_rows):
        s = _as(slots, np.int32)
        r = _as(emb_rows, np.int32)
        assert len(s) == len(r)
        check(self.lib.fm_feat_update(self._ctx, C.c_int(len(s)), _ptr(s), _ptr(r)))

    def feat_merge(self, dst, src):
        check(self.lib.fm_feat_merge(self._ctx, C.c_int(dst), C.c_int(src)))

    def feat_reset(self, slots):
        s = _as(slots, np.int32)
        check(self.lib.fm_feat_reset(self._ctx, C.c_int(len(s)), _ptr(s)))

    def feat_get(self, slot):
        fsum = np.empty(self.feat_dim, np.float32)
        avg = np.empty(self.feat_dim, np.float32)
        cnt = C.c_int32(0)
        check(self.lib.fm_feat_get(self._ctx, C.c_int(slot), _ptr(fsum), _ptr(avg), C.byref(cnt)))
        return fsum, avg, cnt.value

    def feat_read(self, slots):
        s = _as(slots, np.int32)
        avg = np.empty((len(s), self.feat_dim), np.float32)
        cnt = np.zeros(len(s), np.int32)
        check(self.lib.fm_feat_read(self._ctx, C.c_int(len(s)), _ptr(s), _ptr(avg), _ptr(cnt)))
        return avg, cnt

    def feat_write(self, slots, avg, count):
        s = _as(slots, np.int32)
        a = _as(avg, np.float32).reshape(len(s), self.feat_dim)
        c = _as(count, np.int32)
        check(self.lib.fm_feat_write(self._ctx, C.c_int(len(s)), _ptr(s), _ptr(a), _ptr(c)))

    # ------------------------------------------------------------------ association
    def find_occluded(self, tlbr, thresh):
        b = _as(tlbr, np.float64, (-1, 4)) if len(tlbr) else np.zeros((0, 4))
        out = np.zeros(len(b), np.uint8)
        check(self.lib.fm_find_occluded(self._ctx, C.c_int(len(b)), _ptr(b), C.c_double(thresh), _ptr(out)))
        return out.astype(bool)

    def iou_dist(self, a, b):
        a = _as(a, np.float64, (-1, 4))
        b = _as(b, np.float64, (-1, 4))
        out = np.empty((len(a), len(b)), np.float64)
        check(self.lib.fm_iou_dist(self._ctx, C.c_int(len(a)), _ptr(a), C.c_int(len(b)), _ptr(b), _ptr(out)))
        return out

    def assoc_prepare(self, metric, slots, trk_tlbr, trk_label, det_tlbr, det_label, det_occluded,
                      trk_feat_f32=None, after_extractor=False):
        """All pairwise terms of this frame in one launch (fm_assoc_prepare2).  after_extractor: the embeddings are the
        batch extract_async enqueued last and may still be in flight (ordered on the device).  Returns True when the
        terms also reach page-locked host memory, i.e. `assoc_cascade` can run the whole cascade in one call."""
        s = _as(slots, np.int32)
        nT = len(s)
        tb = _as(trk_tlbr, np.float64).reshape(nT, 4)
        tl = _as(trk_label, np.int64)
        db = _as(det_tlbr, np.float64).reshape(-1, 4)
        nD = len(db)
        dl = _as(det_label, np.int64)
        do = _as(det_occluded, np.uint8)
        f32 = np.zeros(nT, np.uint8) if trk_feat_f32 is None else _as(trk_feat_f32, np.uint8)
        assert len(tl) == nT and len(dl) == nD and len(do) == nD and len(f32) == nT
        host = C.c_int(0)
        check(self.lib.fm_assoc_prepare2(self._ctx, C.c_int(metric), C.c_int(nT), _ptr(s), _ptr(tb),
                                         _ptr(tl), C.c_int(nD), _ptr(db), _ptr(dl), _ptr(do), _ptr(f32),
                                         C.c_int(1 if after_extractor else 0), C.byref(host)))
        return bool(host.value)

    def cascade_pack(self, group_sizes, conf_rows, conf_active, unconf_rows, hist_rows, hist_labels, det_conf,
                     motion_weight, max_assoc_cost, fill_val, max_iou_cost, conf_thresh, max_reid_cost):
        """Packs the arguments of `assoc_cascade` (fm_cascade_in); everything here is known before the embeddings are."""
        off = np.zeros(len(group_sizes) + 1, np.int32)
        np.cumsum(group_sizes, out=off[1:])
        arrs = (off, _as(conf_rows, np.int32), _as(conf_active, np.uint8), _as(unconf_rows, np.int32),
                _as(hist_rows, np.int32), _as(hist_labels, np.int64), _as(det_conf, np.float64))
        cin = CascadeIn(len(group_sizes), len(arrs[3]), len(arrs[4]), 0,
                        *(a.__array_interface__['data'][0] for a in arrs),
                        motion_weight, max_assoc_cost, fill_val, max_iou_cost, conf_thresh, max_reid_cost)
        n_trk = len(arrs[1]) + len(arrs[3])
        out = np.empty(CASCADE_HEADER + 3 * n_trk + 3 * len(arrs[6]), np.int32)
        return cin, arrs, out

    def assoc_cascade(self, pack):
        """The association cascade in one call (fm_assoc_cascade); returns the output words as a list."""
        cin, _, out = pack
        check(self.lib.fm_assoc_cascade(self._ctx, C.byref(cin), _ptr(out), C.c_int(len(out))))
        return out[:out[9]].tolist()

    def assoc_get_pairwise(self, nT, nD):
        feat = np.empty((nT, nD), np.float64)
        maha = np.empty((nT, nD), np.float64)
        iou = np.empty((nT, nD), np.float64)
        check(self.lib.fm_assoc_get_pairwise(self._ctx, _ptr(feat), _ptr(maha), _ptr(iou)))
        return feat, maha, iou

    def assoc_stage(self, stage, solver, rows, cols, motion_weight=0., max_cost=0., fill_val=1.,
                    row_labels=None, want_cost=False):
        r = _as(rows, np.int32)
        c = _as(cols, np.int32)
        nr, nc = len(r), len(c)
        mn = min(nr, nc)
        m_rows = np.empty(mn, np.int32)
        m_cols = np.empty(mn, np.int32)
        gated = np.zeros(mn, np.uint8)
        n_match = C.c_int(0)
        cost = np.empty((nr, nc), np.float64) if want_cost else None
        rl = _as(row_labels, np.int64) if row_labels is not None else None
        check(self.lib.fm_assoc_stage(self._ctx, C.c_int(stage), C.c_int(solver), C.c_int(nr), _ptr(r),
                                      C.c_int(nc), _ptr(c), C.c_double(motion_weight),
                                      C.c_double(max_cost), C.c_double(fill_val), _ptr(rl), _ptr(m_rows),
                                      _ptr(m_cols), _ptr(gated), C.byref(n_match), _ptr(cost)))
        k = n_match.value
        return m_rows[:k], m_cols[:k], gated[:k].astype(bool), cost

    def lap(self, cost):
        cm = _as(cost, np.float64)
        nr, nc = cm.shape
        mn = min(nr, nc)
        m_rows = np.empty(mn, np.int32)
        m_cols = np.empty(mn, np.int32)
        n_match = C.c_int(0)
        check(self.lib.fm_lap(self._ctx, _ptr(cm), C.c_int(nr), C.c_int(nc), _ptr(m_rows), _ptr(m_cols),
                              C.byref(n_match)))
        return m_rows[:n_match.value], m_cols[:n_match.value]

    def greedy(self, cost, max_cost):
        cm = _as(cost, np.float64)
        nr, nc = cm.shape
        mn = min(nr, nc)
        m_rows = np.empty(mn, np.int32)
        m_cols = np.empty(mn, np.int32)
        n_match = C.c_int(0)
        check(self.lib.fm_greedy(self._ctx, _ptr(cm), C.c_int(nr), C.c_int(nc), C.c_double(max_cost),
                                 _ptr(m_rows), _ptr(m_cols), C.byref(n_match)))
        return m_rows[:n_match.value], m_cols[:n_match.value]


CASCADE_HEADER = 16


class CascadeIn(C.Structure):          # fm_cascade_in
    _fields_ = [('n_groups', C.c_int32), ('n_unconf', C.c_int32), ('n_hist', C.c_int32), ('reserved', C.c_int32),
                ('group_off', C.c_void_p), ('conf_rows', C.c_void_p), ('conf_active', C.c_void_p),
                ('unconf_rows', C.c_void_p), ('hist_rows', C.c_void_p), ('hist_labels', C.c_void_p),
                ('det_conf', C.c_void_p),
                ('motion_weight', C.c_double), ('max_assoc_cost', C.c_double), ('fill_val', C.c_double),
                ('max_iou_cost', C.c_double), ('conf_thresh', C.c_double), ('max_reid_cost', C.c_double)]


METRIC_EUCLIDEAN = 0
METRIC_COSINE = 1
STAGE_MATCHING = 0
STAGE_IOU = 1
STAGE_REID = 2
SOLVER_LAP = 0
SOLVER_GREEDY = 1


# ---------------------------------------------------------------------- detector / extractor
FM_MAX_HEADS, FM_MAX_ANCHORS = 4, 6


class YoloCfg(C.Structure):
    _fields_ = [('in_w', C.c_int32), ('in_h', C.c_int32),
                ('roi_x', C.c_int32), ('roi_y', C.c_int32), ('roi_w', C.c_int32), ('roi_h', C.c_int32),
                ('input_tensor', C.c_int32), ('n_heads', C.c_int32),
                ('head_tensor', C.c_int32 * FM_MAX_HEADS),
                ('grid_w', C.c_int32 * FM_MAX_HEADS), ('grid_h', C.c_int32 * FM_MAX_HEADS),
                ('n_anchors', C.c_int32 * FM_MAX_HEADS),
                ('anchors', (C.c_float * (2 * FM_MAX_ANCHORS)) * FM_MAX_HEADS),
                ('scale_xy', C.c_float * FM_MAX_HEADS),
                ('num_classes', C.c_int32), ('new_coords', C.c_int32),
                ('label_mask', C.c_uint8 * 128),
                ('conf_thresh', C.c_double), ('nms_thresh', C.c_double), ('max_area', C.c_double),
                ('min_aspect_ratio', C.c_double),
                ('size', C.c_double * 2), ('offset', C.c_double * 2),
                ('max_candidates', C.c_int32)]


DET_DTYPE = np.dtype([('tlbr', float, 4), ('label', int), ('conf', float)], align=True)
assert DET_DTYPE.itemsize == 48


class _PinnedBlock:
    def __init__(self, lib, nbytes):
        self._lib = lib
        p = C.c_void_p()
        check(lib.fm_host_alloc(C.c_size_t(nbytes), C.byref(p)))
        self.ptr = p.value

    def __del__(self):
        if getattr(self, 'ptr', None):
            try:
                self._lib.fm_host_free(C.c_void_p(self.ptr))
            except Exception:
                pass
            self.ptr = None


def pinned_empty(lib, shape, dtype):
    """ndarray in page-locked host memory obtained from fm_host_alloc (freed with the array)."""
    dtype = np.dtype(dtype)
    nbytes = int(np.prod(shape)) * dtype.itemsize
    block = _PinnedBlock(lib, nbytes)
    buf = (C.c_uint8 * nbytes).from_address(block.ptr)
    buf._pinned_block = block              # keeps the allocation alive as long as any view exists
    return np.frombuffer(buf, dtype=dtype).reshape(shape)


def _bind_device_io(cls):
    def frame_configure(self, width, height, ring_size=0):
        check(self.lib.fm_frame_configure(self._ctx, C.c_int(width), C.c_int(height), C.c_int(ring_size)))
        self.frame_size = (width, height)
        self.ring_size = ring_size

    def frame_upload(self, frame):
        w, h = self.frame_size
        if frame.shape != (h, w, 3) or frame.dtype != np.uint8:
            raise ValueError(f'frame must be uint8 {h}x{w}x3')
        f = np.ascontiguousarray(frame)
        check(self.lib.fm_frame_upload(self._ctx, _ptr(f)))

    def pinned_frames(self, n):
        """n frames (n, H, W, 3) uint8 in page-locked host memory (fm_host_alloc): frames stored here are
        uploaded without a staging copy.  The buffer lives as long as the returned array's base object."""
        w, h = self.frame_size
        return pinned_empty(self.lib, (n, h, w, 3), np.uint8)

    def frame_ring_store(self, index, frame):
        f = np.ascontiguousarray(frame, np.uint8)
        check(self.lib.fm_frame_ring_store(self._ctx, C.c_int(index), _ptr(f)))

    def frame_ring_select(self, index):
        check(self.lib.fm_frame_ring_select(self._ctx, C.c_int(index)))

    def frame_upload_next(self, frame):
        w, h = self.frame_size
        if frame.shape != (h, w, 3) or frame.dtype != np.uint8:
            raise ValueError(f'frame must be uint8 {h}x{w}x3')
        f = np.ascontiguousarray(frame)
        check(self.lib.fm_frame_upload_next(self._ctx, _ptr(f)))

    def frame_ring_select_next(self, index):
        check(self.lib.fm_frame_ring_select_next(self._ctx, C.c_int(index)))

    def frame_promote_next(self):
        check(self.lib.fm_frame_promote_next(self._ctx))

    def detect_async_next(self):
        check(self.lib.fm_detect_async_next(self._ctx))

    def frame_read(self):
        w, h = self.frame_size
        out = np.empty((h, w, 3), np.uint8)
        check(self.lib.fm_frame_read(self._ctx, _ptr(out)))
        return out

    def detect_configure(self, cfg):
        check(self.lib.fm_detect_configure(self._ctx, C.byref(cfg)))

    def detect_async(self):
        check(self.lib.fm_detect_async(self._ctx))

    def detect_net_ms(self):
        """HIP-event time of the detector network of the pass collected last, or None when that pass was not timed
        (option 'net_timing' = N: every N-th pass carries the event pair; 0, the default: none)."""
        ms = C.c_float(0)
        check(self.lib.fm_detect_net_ms(self._ctx, C.byref(ms)))
        return ms.value if ms.value >= 0 else None

    def detect_preprocess_only(self):
        check(self.lib.fm_detect_preprocess_only(self._ctx))

    def detect_sync(self, cap=4096):
        # (a staging array kept across calls: a fresh 196 KB np.zeros per frame cost ~10 us on the step's critical chain)
        out = getattr(self, '_det_stage', None)
        if out is None or len(out) < cap:
            out = self._det_stage = np.zeros(cap, DET_DTYPE)
        n = C.c_int(0)
        check(self.lib.fm_detect_sync(self._ctx, _ptr(out), C.c_int(cap), C.byref(n)))
        return out[:n.value].copy().view(np.recarray)

    def filter_dets(self, rows, cap=8192):
        r = _as(rows, np.float32).reshape(-1, 7)
        out = np.zeros(cap, DET_DTYPE)
        n = C.c_int(0)
        check(self.lib.fm_filter_dets(self._ctx, _ptr(r), C.c_int(len(r)), _ptr(out), C.c_int(cap), C.byref(n)))
        return out[:n.value].view(np.recarray)

    def detect_last_counts(self):
        nc, nd = C.c_int(0), C.c_int(0)
        check(self.lib.fm_detect_last_counts(self._ctx, C.byref(nc), C.byref(nd)))
        return nc.value, nd.value

    def detect_raw_candidates(self, cap=65536):
        rows = np.empty((cap, 8), np.float32)
        n = C.c_int(0)
        check(self.lib.fm_detect_raw_candidates(self._ctx, _ptr(rows), C.c_int(cap), C.byref(n)))
        return rows[:n.value]

    def extract_configure(self, input_tensor, in_w, in_h):
        check(self.lib.fm_extract_configure(self._ctx, C.c_int(input_tensor), C.c_int(in_w), C.c_int(in_h)))

    def extract_async(self, tlbrs):
        b = _as(tlbrs, np.float64).reshape(-1, 4)
        check(self.lib.fm_extract_async(self._ctx, C.c_int(len(b)), _ptr(b)))
        return len(b)

    def extract_sync(self, n):
        out = np.empty((n, self.feat_dim), np.float32)
        check(self.lib.fm_extract_sync(self._ctx, C.c_int(n), _ptr(out)))
        return out

    def extract_read_input(self, n, in_w, in_h):
        out = np.empty((n, in_h, in_w, 3), np.float32)
        check(self.lib.fm_extract_read_input(self._ctx, C.c_int(n), _ptr(out)))
        return out

    for fn in (frame_configure, frame_upload, pinned_frames, frame_ring_store, frame_ring_select, frame_read, frame_upload_next,
               frame_ring_select_next, frame_promote_next, detect_async_next,
               detect_configure, detect_async, detect_net_ms, detect_preprocess_only, detect_sync, filter_dets,
               detect_raw_candidates, detect_last_counts, extract_configure, extract_async, extract_sync, extract_read_input):
        setattr(cls, fn.__name__, fn)


_bind_device_io(HipContext)


# ---------------------------------------------------------------------- optical flow
class FlowPredictParams(C.Structure):
    _fields_ = [('feat_density', C.c_double), ('feat_dist_factor', C.c_double),
                ('opt_scale', C.c_float * 2), ('bg_scale', C.c_float * 2),
                ('max_error', C.c_double), ('ransac_max_iter', C.c_int32), ('ransac_conf', C.c_double),
                ('inlier_thresh', C.c_int32), ('frame_w', C.c_int32), ('frame_h', C.c_int32)]


FLOW_OK, FLOW_NO_BACKGROUND, FLOW_NO_HOMOGRAPHY = 0, 1, 2


class FlowCfg(C.Structure):
    _fields_ = [('small_w', C.c_int32), ('small_h', C.c_int32), ('bg_w', C.c_int32), ('bg_h', C.c_int32),
                ('win_size', C.c_int32), ('max_level', C.c_int32), ('max_count', C.c_int32),
                ('epsilon', C.c_double), ('fast_thresh', C.c_int32), ('max_corners', C.c_int32),
                ('block_size', C.c_int32), ('quality_level', C.c_double), ('gray_coeff_bits', C.c_int32)]


def _bind_flow(cls):
    def flow_configure(self, cfg):
        check(self.lib.fm_flow_configure(self._ctx, C.byref(cfg)))

    def flow_init(self):
        check(self.lib.fm_flow_init(self._ctx))

    def flow_begin(self):
        check(self.lib.fm_flow_begin(self._ctx))

    def flow_swap(self):
        check(self.lib.fm_flow_swap(self._ctx))

    def flow_targets(self, inside_tlbr, kps, kp_off):
        r = _as(inside_tlbr, np.float64).reshape(-1, 4)
        nT = len(r)
        off = _as(kp_off, np.int32)
        k = _as(kps, np.float32).reshape(-1, 2)
        assert len(off) == nT + 1 and off[-1] == len(k)
        area = np.zeros(nT, np.int32)
        keep = np.zeros(len(k), np.uint8)
        check(self.lib.fm_flow_targets(self._ctx, C.c_int(nT), _ptr(r), _ptr(k), _ptr(off), _ptr(area), _ptr(keep)))
        return area, keep.astype(bool)

    def flow_prepare(self, inside_tlbr, full_tlbr, kps, kp_off, feat_density, feat_dist_factor,
                     pts_cap=65536, bg_cap=8192):
        r = _as(inside_tlbr, np.float64).reshape(-1, 4)
        nT = len(r)
        fb = _as(full_tlbr, np.float64).reshape(nT, 4)
        off = _as(kp_off, np.int32)
        k = _as(kps, np.float32).reshape(-1, 2)
        assert len(off) == nT + 1 and off[-1] == len(k)
        area = np.zeros(nT, np.int32)
        keep = np.zeros(len(k), np.uint8)
        needy = np.zeros(nT, np.uint8)
        if not hasattr(self, '_flow_bufs') or self._flow_bufs[0].shape[0] < pts_cap or self._flow_bufs[1].shape[0] < bg_cap:
            self._flow_bufs = (np.empty((pts_cap, 2), np.float32), np.empty((bg_cap, 2), np.float32))
        new_pts, bg = self._flow_bufs
        new_off = np.zeros(nT, np.int32)
        new_cnt = np.zeros(nT, np.int32)
        n_new, n_bg = C.c_int(0), C.c_int(0)
        check(self.lib.fm_flow_prepare(self._ctx, C.c_int(nT), _ptr(r), _ptr(fb), _ptr(k), _ptr(off),
                                       C.c_double(feat_density), C.c_double(feat_dist_factor), _ptr(area), _ptr(keep),
                                       _ptr(needy), C.c_int(pts_cap), _ptr(new_pts), _ptr(new_off), _ptr(new_cnt),
                                       C.byref(n_new), C.c_int(bg_cap), _ptr(bg), C.byref(n_bg)))
        return area, keep.astype(bool), needy.astype(bool), new_pts, new_off, new_cnt, bg[:n_bg.value].copy()

    def flow_predict(self, inside_tlbr, full_tlbr, kps, kp_off, params, pts_cap=65536):
        """fm_flow_predict: -> (status, H, result, est_tlbr, n_matched, prev_pts, cur_pts, trk_off, bg_range);
        prev/cur are the compacted RANSAC-inlier keypoints (views of per-call arrays)."""
        r = _as(inside_tlbr, np.float64).reshape(-1, 4)
        nT = len(r)
        fb = _as(full_tlbr, np.float64).reshape(nT, 4)
        off = _as(kp_off, np.int32)
        k = _as(kps, np.float32).reshape(-1, 2)
        assert len(off) == nT + 1 and off[-1] == len(k)
        prev = np.empty((pts_cap, 2), np.float32)
        cur = np.empty((pts_cap, 2), np.float32)
        trk_off = np.zeros(nT + 1, np.int32)
        bg_range = np.zeros(2, np.int32)
        H = np.zeros((3, 3))
        status = C.c_int(0)
        result = np.zeros(nT, np.int32)
        est = np.zeros((nT, 4))
        n_matched = np.zeros(nT, np.int32)
        check(self.lib.fm_flow_predict(self._ctx, C.c_int(nT), _ptr(r), _ptr(fb), _ptr(k), _ptr(off), C.byref(params),
                                       C.c_int(pts_cap), _ptr(prev), _ptr(cur), _ptr(trk_off), _ptr(bg_range),
                                       _ptr(H), C.byref(status), _ptr(result), _ptr(est), _ptr(n_matched)))
        return status.value, H, result, est, n_matched, prev, cur, trk_off, bg_range

    def track_predict_async(self, inside_tlbr, full_tlbr, kps, kp_off, params, slots, ages, sorted_idx, age_penalty,
                            pts_cap=65536):
        """fm_track_predict_async: fm_flow_predict + the Kalman step on the library's worker thread.  Returns a job
        (it owns every buffer the worker reads or writes) for track_predict_wait."""
        job = SimpleNamespace()
        job.r = _as(inside_tlbr, np.float64).reshape(-1, 4)
        nT = len(job.r)
        job.fb = _as(full_tlbr, np.float64).reshape(nT, 4)
        job.off = _as(kp_off, np.int32)
        job.k = _as(kps, np.float32).reshape(-1, 2)
        assert len(job.off) == nT + 1 and job.off[-1] == len(job.k)
        job.prev = np.empty((pts_cap, 2), np.float32)
        job.cur = np.empty((pts_cap, 2), np.float32)
        job.trk_off = np.zeros(nT + 1, np.int32)
        job.bg_range = np.zeros(2, np.int32)
        job.H = np.zeros((3, 3))
        job.result = np.zeros(nT, np.int32)
        job.est = np.zeros((nT, 4))
        job.n_matched = np.zeros(nT, np.int32)
        job.slots = _as(slots, np.int32)
        nK = len(job.slots)
        job.ages = _as(ages, np.int32)
        job.sorted_idx = _as(sorted_idx, np.int32)
        assert len(job.ages) == nK and len(job.sorted_idx) == nK
        job.tlbr = np.empty((nK, 4), np.float64)
        job.lost = np.zeros(nK, np.uint8)
        job.params = params
        check(self.lib.fm_track_predict_async(
            self._ctx, C.c_int(nT), _ptr(job.r), _ptr(job.fb), _ptr(job.k), _ptr(job.off), C.byref(params),
            C.c_int(pts_cap), _ptr(job.prev), _ptr(job.cur), _ptr(job.trk_off), _ptr(job.bg_range), _ptr(job.H),
            _ptr(job.result), _ptr(job.est), _ptr(job.n_matched), C.c_int(nK), _ptr(job.slots), _ptr(job.ages),
            _ptr(job.sorted_idx), C.c_double(float(age_penalty)), _ptr(job.tlbr), _ptr(job.lost)))
        return job

    def track_predict_wait(self, job):
        """-> (the tuple ctx.flow_predict returns, next_tlbrs, lost) ; the last two are None when no Kalman step ran."""
        status, kalman = C.c_int(0), C.c_int(0)
        check(self.lib.fm_track_predict_wait(self._ctx, C.byref(status), C.byref(kalman)))
        pred = (status.value, job.H, job.result, job.est, job.n_matched, job.prev, job.cur, job.trk_off, job.bg_range)
        if kalman.value:
            return pred, job.tlbr, job.lost.astype(bool)
        return pred, None, None

    def flow_detect(self, track_idx, track_tlbr, min_dist, cap=1000):
        idx = _as(track_idx, np.int32)
        n = len(idx)
        tb = _as(track_tlbr, np.float64).reshape(n, 4)
        md = _as(min_dist, np.int32)
        pts = np.empty((n, cap, 2), np.float32)
        cnt = np.zeros(n, np.int32)
        check(self.lib.fm_flow_detect(self._ctx, C.c_int(n), _ptr(idx), _ptr(tb), _ptr(md), C.c_int(cap),
                                      _ptr(pts), _ptr(cnt)))
        return pts, cnt

    def flow_background(self, cap=8192):
        pts = np.empty((cap, 2), np.float32)
        n = C.c_int(0)
        check(self.lib.fm_flow_background(self._ctx, C.c_int(cap), _ptr(pts), C.byref(n)))
        return pts[:n.value]

    def flow_lk(self, prev_pts):
        p = _as(prev_pts, np.float32).reshape(-1, 2)
        n = len(p)
        nxt = np.empty((n, 2), np.float32)
        status = np.zeros(n, np.uint8)
        err = np.zeros(n, np.float32)
        check(self.lib.fm_flow_lk(self._ctx, C.c_int(n), _ptr(p), _ptr(nxt), _ptr(status), _ptr(err)))
        return nxt, status, err

    def gallery_unique_id(self):
        buf = C.create_string_buffer(128)
        check(self.lib.fm_gallery_unique_id(buf))
        return buf.raw

    def gallery_init(self, channel, world, rank, unique_id, row_bytes):
        assert len(unique_id) == 128
        check(self.lib.fm_gallery_init(self._ctx, C.c_int(channel), C.c_int(world), C.c_int(rank), C.c_char_p(unique_id),
                                       C.c_size_t(row_bytes)))

    def gallery_allgather_async(self, channel, row):
        row = np.ascontiguousarray(row, np.uint8)
        check(self.lib.fm_gallery_allgather_async(self._ctx, C.c_int(channel), _ptr(row)))

    def gallery_allgather_wait(self, channel, world, row_bytes):
        out = np.empty(world * row_bytes, np.uint8)
        ms = C.c_float(0)
        check(self.lib.fm_gallery_allgather_wait(self._ctx, C.c_int(channel), _ptr(out), C.byref(ms)))
        return out, float(ms.value)

    def gallery_destroy(self, channel=0):
        check(self.lib.fm_gallery_destroy(self._ctx, C.c_int(channel)))

    def flow_estimate(self, prev_pts, cur_pts, status, begins, ends, bg_begin, bg_end, track_tlbr, size,
                      ransac_max_iter, ransac_conf, inlier_thresh):
        p = _as(prev_pts, np.float32).reshape(-1, 2)
        c = _as(cur_pts, np.float32).reshape(-1, 2)
        st = _as(status, np.uint8)
        n = len(p)
        b, e = _as(begins, np.int32), _as(ends, np.int32)
        nT = len(b)
        tb = _as(track_tlbr, np.float64).reshape(nT, 4)
        H = np.zeros((3, 3))
        ok = C.c_int(0)
        result = np.zeros(nT, np.int32)
        est = np.zeros((nT, 4))
        n_matched = np.zeros(nT, np.int32)
        inl = np.zeros(n, np.uint8)
        check(self.lib.fm_flow_estimate(self._ctx, C.c_int(n), _ptr(p), _ptr(c), _ptr(st), C.c_int(nT), _ptr(b),
                                        _ptr(e), C.c_int(bg_begin), C.c_int(bg_end), _ptr(tb), C.c_int(size[0]),
                                        C.c_int(size[1]), C.c_int(ransac_max_iter), C.c_double(ransac_conf),
                                        C.c_int(inlier_thresh), _ptr(H), C.byref(ok), _ptr(result), _ptr(est),
                                        _ptr(n_matched), _ptr(inl)))
        return (H if ok.value else None), result, est, n_matched, inl.astype(bool)

    def flow_read_image(self, which):
        w, h = C.c_int(0), C.c_int(0)
        buf = np.empty(self.frame_size[0] * self.frame_size[1] * 4, np.uint8)
        check(self.lib.fm_flow_read_image(self._ctx, C.c_int(which), _ptr(buf), C.byref(w), C.byref(h)))
        return buf[:w.value * h.value].reshape(h.value, w.value).copy()

    for fn in (flow_configure, flow_init, flow_begin, track_predict_async,
               track_predict_wait, flow_swap, flow_targets, flow_prepare, flow_predict, flow_detect,
               flow_background,
               flow_lk, gallery_unique_id, gallery_init,
               gallery_allgather_async, gallery_allgather_wait, gallery_destroy, flow_estimate, flow_read_image):
        setattr(cls, fn.__name__, fn)


_bind_flow(HipContext)
