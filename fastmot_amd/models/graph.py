"""Layer-table builder for the HIP conv engine (include/fastmot_hip.h: fm_tensor / fm_layer).

Replaces the ONNX -> TensorRT engine build of the reference (fastmot/models/yolo.py:106-151,
fastmot/models/reid.py:48-92).  A Graph is a list of NHWC fp16 tensors (channels padded to 8) and
layers; concat / route is expressed by writing producers into channel slices of a shared tensor,
BatchNorm is folded into the conv weights (eps 1e-5, scripts/yolo2onnx.py:419-421) and packed in
the MFMA kernel's [cout_pad32][K_pad64] fp16 layout.
"""
import ctypes as C
import logging
import os

import numpy as np

LOGGER = logging.getLogger(__name__)

(OP_CONV, OP_DWCONV3, OP_MAXPOOL, OP_AVGPOOL, OP_UPSAMPLE2, OP_COPY, OP_GATE, OP_GATE_SUM, OP_HEAD,
 OP_LITECONV, OP_SPP, OP_GATED_SUM, OP_STEMCONV, OP_ADD, OP_RESBLOCK, OP_CONVS, OP_LITECHAIN, OP_CONVD, OP_STEM2, OP_PAIR11,
 OP_OSTAIL) = range(21)
CONV_OPS = (OP_CONV, OP_CONVS, OP_CONVD)      # the three kernels behind Graph.conv (same fields, different weight layouts)
SPP_MAX_HW = 2048
ACT = {'linear': 0, 'leaky': 1, 'mish': 2, 'relu': 3, 'logistic': 4, 'swish': 5}
RES_NONE, RES_AFTER_ACT, RES_BEFORE_ACT = 0, 1, 2


class fm_tensor(C.Structure):
    _fields_ = [('h', C.c_int32), ('w', C.c_int32), ('c', C.c_int32), ('f32', C.c_int32), ('offset', C.c_int64)]


class fm_layer(C.Structure):
    _fields_ = [('op', C.c_int32), ('n_in', C.c_int32),
                ('in_', C.c_int32 * 4), ('in_coff', C.c_int32 * 4),
                ('out', C.c_int32), ('out_coff', C.c_int32),
                ('res', C.c_int32), ('res_coff', C.c_int32), ('res_mode', C.c_int32),
                ('cin', C.c_int32), ('cout', C.c_int32), ('k', C.c_int32), ('stride', C.c_int32),
                ('pad', C.c_int32), ('act', C.c_int32), ('hid', C.c_int32), ('up', C.c_int32),
                ('gate', C.c_int32 * 4),
                ('w_off', C.c_int64), ('b_off', C.c_int64), ('w2_off', C.c_int64), ('b2_off', C.c_int64)]


def ceil_to(x, m):
    return (x + m - 1) // m * m


class View:
    """A channel slice [coff, coff + c) of tensor `tid` (c = logical channels)."""

    def __init__(self, tid, coff, c, h, w):
        self.tid, self.coff, self.c, self.h, self.w = tid, coff, c, h, w

    @property
    def cpad(self):
        return ceil_to(self.c, 8)

    def slice(self, coff, c):
        assert coff % 8 == 0
        return View(self.tid, self.coff + coff, c, self.h, self.w)


_ALLOW_RANDOM = [os.environ.get('FASTMOT_RANDOM_WEIGHTS', '0') == '1']


def allow_random_weights(flag=True):
    """Benchmarks / tests without weight files: let model descriptors whose MODEL_PATH is missing fall back to
    seeded random parameters (also FASTMOT_RANDOM_WEIGHTS=1).  Off by default: like the reference, a missing
    model file is an error -- tracks computed from random weights look plausible and mean nothing."""
    _ALLOW_RANDOM[0] = bool(flag)


def missing_weights(model, seed):
    """Policy for a descriptor whose weight file does not exist (called by YOLO/ReID.build_graph)."""
    if not _ALLOW_RANDOM[0]:
        raise FileNotFoundError(
            f'{model.__name__}: weight file {model.MODEL_PATH} not found.  Put the file there, pass weights=..., '
            'or opt in to seeded random weights with fastmot_amd.models.allow_random_weights() / '
            'FASTMOT_RANDOM_WEIGHTS=1 (benchmarks and tests only)')
    LOGGER.warning('%s: %s not found -- running with SEEDED RANDOM weights (outputs are meaningless)',
                   model.__name__, model.MODEL_PATH)
    return RandomWeights(seed=seed)


class RandomWeights:
    """Seeded random parameters with variance-preserving scaling (no trained weights are available
    offline; real Darknet/torchreid checkpoints plug in through the same interface)."""

    def __init__(self, seed=0):
        self.rng = np.random.default_rng(seed)

    def conv(self, name, cout, cin, k, bn=True, gain=1.0, groups=1):
        fan_in = cin // groups * k * k
        w = self.rng.normal(0, gain * np.sqrt(1.0 / fan_in), (cout, cin // groups, k, k)).astype(np.float32)
        if bn:
            gamma = self.rng.uniform(0.8, 1.2, cout).astype(np.float32)
            beta = self.rng.normal(0, 0.1, cout).astype(np.float32)
            mean = self.rng.normal(0, 0.1, cout).astype(np.float32)
            var = self.rng.uniform(0.8, 1.2, cout).astype(np.float32)
            return dict(w=w, gamma=gamma, beta=beta, mean=mean, var=var)
        return dict(w=w, bias=self.rng.normal(0, 0.1, cout).astype(np.float32))

    def linear(self, name, cout, cin, bn=False):
        p = self.conv(name, cout, cin, 1, bn=bn)
        p['w'] = p['w'].reshape(cout, cin)
        return p


def fold_bn(p, eps=1e-5):
    """-> (weight fp32, bias fp32) with BatchNorm folded (yolo2onnx.py:419-421 eps)."""
    w = p['w'].astype(np.float32)
    if 'gamma' in p:
        scale = p['gamma'] / np.sqrt(p['var'] + eps)
        w = w * scale.reshape(-1, *([1] * (w.ndim - 1)))
        b = p['beta'] - p['mean'] * scale
        if 'bias' in p:                  # Linear/Conv with its own bias followed by BN (OSNet fc head)
            b = b + p['bias'] * scale
    else:
        b = p.get('bias', np.zeros(w.shape[0], np.float32))
    return w.astype(np.float32), b.astype(np.float32)


class Graph:
    def __init__(self, weights, in_hw, in_c=3):
        self.wsrc = weights
        self.tensors = []      # (h, w, cpad, f32)
        self.layers = []       # dicts
        self.blob = bytearray()
        self.n_gates = 0
        self.gate_c = 8
        self.use_stem = True   # small-Cin first layers go to the LDS-patch stem kernel
        # darknet residual units (1x1, 3x3, shortcut) as one fused launch (resblock.hip)
        self.use_resblock = os.environ.get('FASTMOT_RESBLOCK', '1') != '0'
        # the four LightConv streams of an OSNet block as one launch (litechain.hip) instead of one per depth
        self.use_lightchain = os.environ.get('FASTMOT_LITECHAIN', '1') != '0'
        # convs with at most this many output pixels per sample and a long reduction take the streamed
        # kernel (K split inside the workgroup, convs.hip) instead of the LDS-tiled one + split-K reduce
        self.convs_max_pixels = int(os.environ.get('FASTMOT_CONVS_MAXP', '1444'))
        # DMA-fed multi-accumulator kernel (convd.hip) for 1x1 / 3x3 layers with cin % 64 == 0: 0 = never; 1 = where it
        # measured faster (profiles/r05_convd_sweep_*.txt): instead of the LDS-tiled kernel everywhere, instead of the
        # streamed kernel only for the stride-2 3x3 convs into a map of >= 1024 pixels (the streamed kernel keeps the
        # 38 x 38 / 19 x 19 levels: 13.5 vs 14.1 us and 15.1 vs 19.4 us on their 3x3 layers); 2 = wherever it applies
        self.convd_level = int(os.environ.get('FASTMOT_CONVD', '1'))
        self.convd_min_cin1 = 16   # smallest cin of a 1x1 layer on it (64: profiles/r05_osnet_pointwise_on_convd_ab.txt)
        # the stem (3 -> 32, 3x3 s1) and the stride-2 3x3 conv behind it as one launch (stem2.hip, FM_OP_STEM2)
        self.use_stem2 = os.environ.get('FASTMOT_STEM2', '2') != '0'
        self.use_stem3 = os.environ.get('FASTMOT_STEM2', '2') == '2'     # ... and the pointwise conv behind the pair as its third stage
        # a 64 -> 64 pointwise conv into the first half of a concat + the 128 -> 64 / 128 pointwise conv over that concat as one
        # launch (pair11.hip, FM_OP_PAIR11: the tail of the first two CSP stages)
        self.use_pair11 = os.environ.get('FASTMOT_PAIR11', '1') != '0'
        # OSNet x0.25's 16 x 8 stage + conv5 + head as one launch, one workgroup per sample (ostail.hip, FM_OP_OSTAIL)
        self.use_ostail = os.environ.get('FASTMOT_OSTAIL', '1') != '0'
        self.conv_params = []  # (layer index, folded fp16-rounded weight fp32, bias) for the test oracle
        h, w = in_hw
        self.input = self.new(h, w, in_c)
        self.outputs = []

    # ---------------------------------------------------------------- tensors / blob
    def new(self, h, w, c, f32=False):
        self.tensors.append((h, w, ceil_to(c, 8), int(f32)))
        return View(len(self.tensors) - 1, 0, c, h, w)

    def _push(self, arr):
        while len(self.blob) % 16:
            self.blob.append(0)
        off = len(self.blob)
        self.blob += np.ascontiguousarray(arr).tobytes()
        return off

    def _layer(self, **kw):
        d = dict(op=0, ins=[], out=None, res=None, res_mode=RES_NONE, cin=0, cout=0, k=1, stride=1, pad=0,
                 act=0, hid=0, up=1, gates=[], w_off=0, b_off=0, w2_off=0, b2_off=0)
        d.update(kw)
        self.layers.append(d)
        return d

    # ---------------------------------------------------------------- ops
    def conv(self, name, x, cout, k=1, stride=1, act='linear', bn=True, dst=None, res=None,
             res_mode=RES_AFTER_ACT, f32_out=False, pad=None, bias=True, up=1, wb=None):
        cin_pad = x.cpad
        pad = k // 2 if pad is None else pad
        ho = (x.h + 2 * pad - k) // stride + 1
        wo = (x.w + 2 * pad - k) // stride + 1
        if dst is None:
            dst = self.new(ho * up, wo * up, cout, f32=f32_out)
        assert dst.h == ho * up and dst.w == wo * up and dst.c == cout, (name, dst.h, ho, dst.c, cout)
        assert up in (1, 2) and not (up == 2 and f32_out)
        if wb is not None:      # pre-folded (weight [cout, x.c, k, k], bias [cout]) supplied by the caller
            w, b = (np.asarray(a, np.float32) for a in wb)
            assert w.shape == (cout, x.c, k, k) and b.shape == (cout,)
        else:
            w, b = fold_bn(self.wsrc.conv(name, cout, x.c, k, bn=bn))
        if not bias:
            b = np.zeros_like(b)
        w16 = w.astype(np.float16)
        prev = self._pair11_applies(x, cout, k, stride, pad, res, f32_out, up)
        if prev is not None:
            # `x` is a 128-channel concat whose first 64 channels the previous layer -- a 64 -> 64 pointwise conv -- has just
            # written: both as ONE launch (pair11.hip); that half of the concat is never stored (check_fusions: nobody else
            # may read it afterwards).  Same K order as the unfused conv over the concat: the results are bit-identical.
            self.layers.pop()
            _, w1ref, b1ref = self.conv_params.pop()
            w1p = np.zeros((64, prev['ins'][0].c, 1, 1), np.float16)
            w1p[:] = w1ref.astype(np.float16)
            wp = np.zeros((ceil_to(cout, 32), x.c, 1, 1), np.float16)
            wp[:cout] = w16
            bias2 = np.zeros(wp.shape[0], np.float32)
            bias2[:cout] = b
            self._pair11_dropped = getattr(self, '_pair11_dropped', []) + [(len(self.layers), x.tid, x.coff, 64)]
            self._layer(op=OP_PAIR11, ins=[prev['ins'][0], x.slice(64, 64)], out=dst, cin=64, cout=cout, k=1, stride=1, pad=0,
                        act=ACT[act], hid=64, gates=[prev['act']], w_off=self._push(self._pack_frag(w1p)),
                        b_off=self._push(np.asarray(b1ref, np.float32)), w2_off=self._push(self._pack_frag(wp)),
                        b2_off=self._push(bias2), name=name,
                        pair_ref=(w1ref, b1ref, prev['act'], w16.astype(np.float32), b))
            return dst
        if self._stem3_applies(x, cout, k, stride, pad, res, f32_out, up):
            # a pointwise conv over the whole output of the stem pair (the first CSP stage's merged 1x1 conv): third stage of the
            # same launch (stem2.hip, C3) -- the stride-2 conv's output is never stored either
            pair = self.layers.pop()
            w1ref, b1ref, a1, w2ref, b2ref = pair['stem2_ref']
            w2p = np.zeros((64, 32, 3, 3), np.float16)
            w2p[:] = np.asarray(w2ref, np.float16)
            w3p = np.zeros((ceil_to(cout, 32), 64, 1, 1), np.float16)
            w3p[:cout] = w16
            b23 = np.zeros(64 + w3p.shape[0], np.float32)
            b23[:64] = b2ref
            b23[64:64 + cout] = b
            blob2 = np.concatenate([self._pack_frag(w2p).reshape(-1), self._pack_frag(w3p).reshape(-1)])
            self._stem2_dropped = [self._stem2_dropped, x.tid] if not isinstance(self._stem2_dropped, list) else self._stem2_dropped + [x.tid]
            self._layer(op=OP_STEM2, ins=pair['ins'], out=dst, cin=pair['cin'], cout=cout, k=3, stride=2, pad=1, act=ACT[act],
                        hid=32, gates=[a1, 64, pair['act']], w_off=pair['w_off'], b_off=pair['b_off'],
                        w2_off=self._push(blob2), b2_off=self._push(b23), name=name,
                        stem2_ref=(w1ref, b1ref, a1, w2ref, b2ref), stem3_ref=(pair['act'], w16.astype(np.float32), b))
            return dst
        if self._stem2_applies(x, cout, k, stride, pad, res, f32_out, up):
            # second layer of the network, a 3x3 stride-2 conv over the whole output of the stem layer: both as ONE launch
            # (stem2.hip) -- the 32-channel full-resolution tensor between them (27 MB at 608 x 608) is never stored.  The
            # stem's layer entry is replaced; finish() checks that nothing else wanted its output.
            stem = self.layers.pop()
            _, w1ref, b1ref = self.conv_params.pop()
            wp = np.zeros((ceil_to(cout, 32), x.c, 3, 3), np.float16)
            wp[:cout] = w16
            bias2 = np.zeros(wp.shape[0], np.float32)
            bias2[:cout] = b
            self._stem2_dropped = x.tid
            self._layer(op=OP_STEM2, ins=stem['ins'], out=dst, cin=stem['cin'], cout=cout, k=3, stride=2, pad=1, act=ACT[act],
                        hid=32, gates=[stem['act']], w_off=stem['w_off'], b_off=stem['b_off'],
                        w2_off=self._push(self._pack_frag(wp)), b2_off=self._push(bias2), name=name,
                        stem2_ref=(w1ref, b1ref, stem['act'], w16.astype(np.float32), b))
            return dst
        if (self.use_stem and x.c <= 4 and x.coff == 0 and cout <= 32 and (k, stride) in ((3, 1), (3, 2), (7, 2))
                and res is None and not f32_out and up == 1):
            # stem layer: input patch staged in LDS instead of 16 B gathers per tap (stemconv.hip)
            kp = ceil_to(k * k * 4, 16)
            wk = np.zeros((32, k, k, 4), np.float16)
            wk[:cout, :, :, :x.c] = w16.transpose(0, 2, 3, 1)
            packed = np.zeros((32, kp), np.float16)
            packed[:, :k * k * 4] = wk.reshape(32, -1)
            bias32 = np.zeros(32, np.float32)
            bias32[:cout] = b
            self._layer(op=OP_STEMCONV, ins=[x], out=dst, cin=cin_pad, cout=cout, k=k, stride=stride, pad=pad,
                        act=ACT[act], w_off=self._push(packed), b_off=self._push(bias32), name=name)
            self.conv_params.append((len(self.layers) - 1, w16.astype(np.float32), b))
            return dst
        streamed = x.c == cin_pad and cin_pad % 64 == 0 and k * k * cin_pad >= 512 and ho * wo <= self.convs_max_pixels
        beats_streamed = k == 3 and stride == 2 and ho * wo >= 1024
        # (1x1: any cin % 8 == 0 -- the K range is zero-padded to whole 64-deep steps; OSNet's 16 .. 96-channel pointwise convs)
        if ((k == 3 and cin_pad % 64 == 0) or (k == 1 and pad == 0 and cin_pad >= self.convd_min_cin1)) and \
                self.convd_level >= (1 if not streamed or beats_streamed else 2):
            wk4 = np.zeros((ceil_to(cout, 32), k, k, cin_pad), np.float16)
            wk4[:cout, :, :, :x.c] = w16.transpose(0, 2, 3, 1)
            wk = np.zeros((wk4.shape[0], ceil_to(k * k * cin_pad, 64)), np.float16)
            wk[:, :k * k * cin_pad] = wk4.reshape(wk4.shape[0], -1)
            bias = np.zeros(wk.shape[0], np.float32)
            bias[:cout] = b
            self._layer(op=OP_CONVD, ins=[x], out=dst, cin=cin_pad, cout=cout, k=k, stride=stride, pad=pad,
                        act=ACT[act], up=up, w_off=self._push(self._pack_tile64(wk)),
                        b_off=self._push(bias), res=res, res_mode=res_mode if res is not None else RES_NONE, name=name)
            self.conv_params.append((len(self.layers) - 1, w16.astype(np.float32), b))
            return dst
        if streamed:
            wp = np.zeros((ceil_to(cout, 32), x.c, k, k), np.float16)
            wp[:cout] = w16
            bias = np.zeros(wp.shape[0], np.float32)
            bias[:cout] = b
            self._layer(op=OP_CONVS, ins=[x], out=dst, cin=cin_pad, cout=cout, k=k, stride=stride, pad=pad,
                        act=ACT[act], up=up, w_off=self._push(self._pack_frag(wp)), b_off=self._push(bias),
                        res=res, res_mode=res_mode if res is not None else RES_NONE, name=name)
            self.conv_params.append((len(self.layers) - 1, w16.astype(np.float32), b))
            return dst
        # pack [cout_pad32][Kpad64], K order (kh, kw, cin_pad)
        K = k * k * cin_pad
        kpad = ceil_to(K, 64)
        cpad = ceil_to(cout, 32)
        wk = np.zeros((cpad, k, k, cin_pad), np.float16)
        wk[:cout, :, :, :x.c] = w16.transpose(0, 2, 3, 1)
        packed = np.zeros((cpad, kpad), np.float16)
        packed[:, :K] = wk.reshape(cpad, K)
        bias = np.zeros(cpad, np.float32)
        bias[:cout] = b
        lay = self._layer(op=OP_CONV, ins=[x], out=dst, cin=cin_pad, cout=cout, k=k, stride=stride, pad=pad,
                          act=ACT[act], up=up, w_off=self._push(packed), b_off=self._push(bias),
                          res=res, res_mode=res_mode if res is not None else RES_NONE, name=name)
        self.conv_params.append((len(self.layers) - 1, w16.astype(np.float32), b))
        return dst

    def _stem2_applies(self, x, cout, k, stride, pad, res, f32_out, up):
        if not (self.use_stem2 and len(self.layers) == 1 and k == 3 and stride == 2 and pad == 1 and res is None and
                not f32_out and up == 1 and cout in (64, 128)):
            return False
        stem = self.layers[0]
        return (stem['op'] == OP_STEMCONV and stem['k'] == 3 and stem['stride'] == 1 and stem['pad'] == 1 and
                stem['cout'] == 32 and stem['ins'][0].tid == self.input.tid and stem['out'].tid == x.tid and x.coff == 0 and
                x.c == 32 and x.tid != self.input.tid and x.tid not in [v.tid for v in self.outputs])

    def _pair11_applies(self, x, cout, k, stride, pad, res, f32_out, up):
        """-> the previous layer's dict when it and this conv form a fusable pair, else None."""
        if not (self.use_pair11 and self.layers and k == 1 and stride == 1 and pad == 0 and res is None and not f32_out and
                up == 1 and cout in (64, 128) and x.c == 128 and x.cpad == 128 and x.coff % 8 == 0):
            return None
        prev = self.layers[-1]
        if not (prev['op'] in CONV_OPS and prev['k'] == 1 and prev['stride'] == 1 and prev['cout'] == 64 and prev['res'] is None and
                prev['up'] == 1 and prev['out'].tid == x.tid and prev['out'].coff == x.coff and prev['out'].c == 64 and
                prev['ins'][0].c == 64 and prev['ins'][0].cpad % 8 == 0 and prev['ins'][0].tid != x.tid and
                not self.tensors[x.tid][3] and self.conv_params and self.conv_params[-1][0] == len(self.layers) - 1):
            return None
        if x.h * x.w < 8192:        # (the small maps' pointwise layers are launch bound, not traffic bound: measured on the large ones only)
            return None
        return prev

    def _stem3_applies(self, x, cout, k, stride, pad, res, f32_out, up):
        if not (self.use_stem2 and self.use_stem3 and len(self.layers) == 1 and k == 1 and stride == 1 and pad == 0 and res is None and
                not f32_out and up == 1 and cout in (64, 128)):
            return False
        pair = self.layers[0]
        return (pair['op'] == OP_STEM2 and len(pair['gates']) == 1 and pair['cout'] == 64 and pair['out'].tid == x.tid and
                x.coff == 0 and x.c == 64 and pair['out'].coff == 0 and x.tid not in [v.tid for v in self.outputs])

    def check_fusions(self):
        """A fused stem pair dropped the stem's output tensor: no later layer may read it (a cfg that routes from layer 0
        has to be built with FASTMOT_STEM2=0 / use_stem2 = False)."""
        for li, tid, coff, c in getattr(self, '_pair11_dropped', []):
            for d in self.layers[li + 1:]:
                for v in d['ins'] + ([d['res']] if d['res'] is not None else []):
                    if v.tid == tid and v.coff < coff + c and coff < v.coff + v.c:
                        raise ValueError('a fused 1x1 pair dropped a concat slice that a later layer reads: build with use_pair11 = False')
        t = getattr(self, '_stem2_dropped', None)
        if t is None:
            return
        dropped = t if isinstance(t, list) else [t]
        for d in self.layers:
            used = [v.tid for v in d['ins']] + ([d['res'].tid] if d['res'] is not None else [])
            if any(t in used or t in [v.tid for v in self.outputs] for t in dropped):
                raise ValueError('the stem\'s output is read by another layer: build this network with use_stem2 = False')

    def fuse_ostail(self, first):
        """Collapses layers[first:] -- built by models/reid.py osnet_graph as: transition AvgPool2d(2, 2) over a 32 x 16 x 96 map,
        OSBlock 96 -> 128 (conv1, four-stream chain, gate, conv3 + downsample over the concat [x2 | x]), OSBlock 128 -> 128
        (conv1, chain, gate, conv3 + identity), conv5, head 128 -> 512 -- into ONE FM_OP_OSTAIL layer (ostail.hip: one
        workgroup per sample keeps the 128-pixel maps in LDS from the pool to the embedding).  Returns False, leaving the
        table as it is, when the layers are not that sequence (other widths, per-depth LightConv launches)."""
        sub = self.layers[first:]
        if not self.use_ostail or len(sub) != 11:
            return False
        pool, c1a, cha, ga, c3a, c1b, chb, gb, c3b, c5, head = sub
        relu = ACT['relu']

        def conv(d, cin, cout, res=False):
            return (d['op'] in CONV_OPS and d['k'] == 1 and d['stride'] == 1 and d['ins'][0].c == cin and d['cout'] == cout and
                    d['act'] == relu and (d['res'] is not None) == res and d['up'] == 1)

        def streams(ch, g, x1):
            return (ch['op'] == OP_LITECHAIN and ch['cout'] == 32 and ch['act'] == relu and ch['ins'][0].tid == x1['out'].tid and
                    g['op'] == OP_GATED_SUM and g['hid'] == 2 and len(g['ins']) == 4 and all(v.tid == ch['out'].tid for v in g['ins']))
        x = pool['ins'][0]
        ok = (pool['op'] == OP_AVGPOOL and (pool['k'], pool['stride'], pool['pad']) == (2, 2, 0) and (x.h, x.w, x.c) == (32, 16, 96) and
              x.coff % 8 == 0 and conv(c1a, 96, 32) and c1a['ins'][0].tid == pool['out'].tid and streams(cha, ga, c1a) and
              conv(c3a, 128, 128) and c3a['ins'][0].tid == pool['out'].tid == ga['out'].tid and
              (ga['out'].coff, pool['out'].coff) == (c3a['ins'][0].coff, c3a['ins'][0].coff + 32) and
              conv(c1b, 128, 32) and c1b['ins'][0].tid == c3a['out'].tid and streams(chb, gb, c1b) and
              conv(c3b, 32, 128, res=True) and c3b['res_mode'] == RES_BEFORE_ACT and c3b['res'].tid == c3a['out'].tid and
              c3b['ins'][0].tid == gb['out'].tid and conv(c5, 128, 128) and c5['ins'][0].tid == c3b['out'].tid and
              head['op'] == OP_HEAD and head['cout'] == 512 and head['ins'][0].tid == c5['out'].tid and
              (head['ins'][0].h, head['ins'][0].w) == (16, 8))
        if not ok:
            return False
        cp = {i: (w, b) for i, w, b in self.conv_params}

        def frag(w):
            w = np.asarray(w, np.float32)
            return self._pack_frag(w.reshape(w.shape[0], -1, 1, 1).astype(np.float16)).reshape(-1)

        def chain(ch):
            refs = ch['lite_ref']
            return (np.concatenate([frag(pw) for pw, _, _ in refs]),
                    np.concatenate([np.asarray(wd, np.float16).reshape(32, 9).T.reshape(-1) for _, wd, _ in refs]),
                    np.concatenate([np.asarray(bd, np.float32) for _, _, bd in refs]))

        def gate(g):
            w1, b1, w2, b2 = (np.asarray(a, np.float32) for a in g['gate_ref'])
            return np.concatenate([w1.reshape(-1), b1, np.zeros(2, np.float32), w2.reshape(-1), b2])
        (w1a, b1a), (w3a, b3a), (w1b, b1b), (w3b, b3b), (w5, b5) = (cp[first + i] for i in (1, 4, 5, 8, 9))
        pwa, dwa, bca = chain(cha)
        pwb, dwb, bcb = chain(chb)
        wfc, bfc = head['head_ref']
        halfs = np.concatenate([frag(w1a), pwa, dwa, frag(w3a), frag(w1b), pwb, dwb, frag(w3b), frag(w5),
                                np.asarray(wfc, np.float16).reshape(-1)])
        floats = np.concatenate([np.asarray(a, np.float32).reshape(-1) for a in
                                 (b1a, bca, gate(ga), b3a, b1b, bcb, gate(gb), b3b, b5, bfc)])
        assert halfs.dtype == np.float16 and floats.dtype == np.float32 and (len(halfs), len(floats)) == (135808, 1928)
        del self.layers[first:]
        self.conv_params = [(i, w, b) for i, w, b in self.conv_params if i < first]
        # (out: as FM_OP_HEAD, a view of the input -- the result goes to the context's embedding buffer)
        self._layer(op=OP_OSTAIL, ins=[x], out=View(x.tid, x.coff, x.c, x.h, x.w), cin=96, hid=32, k=128, cout=512, stride=len(halfs), pad=len(floats),
                    w_off=self._push(halfs), b_off=self._push(floats), name='ostail',
                    sub=[(d, cp.get(first + i)) for i, d in enumerate(sub)])
        return True

    @staticmethod
    def resblock_supported(c, mid):
        return c in (64, 128, 256) and mid in (c, c // 2)

    def resblock_pays(self, c, mid, h, w):
        """The fused residual unit (resblock.hip) against its two convs on the DMA kernel (convd.hip, shortcut in the
        3x3's epilogue): measured per shape in round 5 (profiles/r05_layers_YOLOv4P6_1280_convd*.txt and the sweeps r05_convd_sweep*.txt) -- at 80 x 80 x 256 (YOLOv4-P6) the
        fused unit takes 41.5 us, the pair 6.7 + 18.5 us; at 160 x 160 x 128 25.4 against 7 + 16.5; on the maps of
        YOLOv4 @ 608 / -CSP @ 640 (<= 0.8 M activations per unit) the fused unit wins (8.9 against ~13 us at
        76 x 76 x 128).  Without the DMA kernel the fused unit always pays."""
        return not (self.convd_level >= 1 and c >= 128 and mid % 64 == 0 and h * w * c >= 1500000)

    @staticmethod
    def _pack_frag(w16):
        """[cout, cin, k, k] -> MFMA A-fragment order [cout/32][K/16][lane][8], K order (kh, kw, cin),
        lane = (k / 8 % 2) * 32 + cout % 32 (cout % 32 == 0, K % 16 == 0)."""
        cout, cin, k, _ = w16.shape
        K = k * k * cin
        assert cout % 32 == 0 and K % 16 == 0
        rows = w16.transpose(0, 2, 3, 1).reshape(cout // 32, 32, K // 16, 2, 8)
        return np.ascontiguousarray(rows.transpose(0, 2, 3, 1, 4))

    @staticmethod
    def _pack_tile64(wmat):
        """[cout (% 32 == 0), K (% 64 == 0)] fp16, K order (kh, kw, cin) -> the LDS tile images of convd.hip:
        [cout/32][K/64][32 rows][8 slots][8 halfs], slot s of row r = K chunk s ^ ((r / 2) % 8) of the row's K step
        (the XOR swizzle that makes the kernel's ds_read_b128 fragment reads conflict-free; the DMA writes LDS
        lane-linear, so the permutation has to be in the source)."""
        cout, K = wmat.shape
        assert cout % 32 == 0 and K % 64 == 0
        rows = wmat.reshape(cout // 32, 32, K // 64, 8, 8).transpose(0, 2, 1, 3, 4)     # [blk][step][row][chunk][8]
        r = np.arange(32)[:, None]
        src = np.arange(8)[None, :] ^ ((r >> 1) & 7)                                    # chunk held by (row, slot)
        return np.ascontiguousarray(rows[:, :, r, src, :])

    def resblock(self, name1, name2, x, mid, act='mish', dst=None, wb1=None, wb2=None, bn1=True, bn2=True):
        """Darknet residual unit in one launch (resblock.hip): x + act(conv3x3(act(conv1x1(x)))).
        Parameters are drawn in layer order (1x1 then 3x3), like the unfused layers."""
        c = x.c
        assert self.resblock_supported(c, mid) and x.cpad == c
        if dst is None:
            dst = self.new(x.h, x.w, c)
        w1, b1 = wb1 if wb1 is not None else fold_bn(self.wsrc.conv(name1, mid, c, 1, bn=bn1))
        w2, b2 = wb2 if wb2 is not None else fold_bn(self.wsrc.conv(name2, c, mid, 3, bn=bn2))
        w1, b1, w2, b2 = (np.asarray(a, np.float32) for a in (w1, b1, w2, b2))
        w1h, w2h = w1.astype(np.float16), w2.astype(np.float16)
        self._layer(op=OP_RESBLOCK, ins=[x], out=dst, cin=c, cout=c, k=3, stride=1, pad=1, act=ACT[act], hid=mid,
                    w_off=self._push(self._pack_frag(w1h)), b_off=self._push(b1),
                    w2_off=self._push(self._pack_frag(w2h)), b2_off=self._push(b2), name=name2,
                    res_ref=(w1h.astype(np.float32), b1, w2h.astype(np.float32), b2))
        return dst

    def dwconv3(self, name, x, act='relu', dst=None):
        c = x.c
        if dst is None:
            dst = self.new(x.h, x.w, c)
        p = self.wsrc.conv(name, c, c, 3, bn=True, groups=c)
        w, b = fold_bn(p)                       # [c,1,3,3]
        w16 = w.astype(np.float16)
        wk = np.zeros((9, x.cpad), np.float16)
        wk[:, :c] = w16.reshape(c, 9).T
        bias = np.zeros(x.cpad, np.float32)
        bias[:c] = b
        self._layer(op=OP_DWCONV3, ins=[x], out=dst, cin=x.cpad, cout=c, k=3, stride=1, pad=1, act=ACT[act],
                    w_off=self._push(wk), b_off=self._push(bias), name=name)
        self.conv_params.append((len(self.layers) - 1, w16.astype(np.float32), b))
        return dst

    def lightconv_params(self, name, c):
        """Draws + packs the parameters of one torchreid LightConv3x3 (1x1 linear bias=False ->
        depthwise 3x3 bias=False -> BN): pointwise first, then depthwise, as the unfused layers do."""
        cp = ceil_to(c, 8)
        pw = self.wsrc.conv(name + '.pw', c, c, 1, bn=False)['w'].astype(np.float16)     # [c, c, 1, 1]
        packed = np.zeros((ceil_to(c, 32), ceil_to(cp, 64)), np.float16)
        packed[:c, :c] = pw.reshape(c, c)
        wd, bd = fold_bn(self.wsrc.conv(name + '.dw', c, c, 3, bn=True, groups=c))
        wd16 = wd.astype(np.float16)
        wk = np.zeros((9, cp), np.float16)
        wk[:, :c] = wd16.reshape(c, 9).T
        bias = np.zeros(cp, np.float32)
        bias[:c] = bd
        return dict(pw=packed, dw=wk, bias=bias, ref=(pw.astype(np.float32), wd16.astype(np.float32), bd))

    def lightconv_group(self, name, xs, params, act='relu', dst=None, gap_slot=False):
        """G = len(xs) <= 4 independent fused LightConv3x3 of equal geometry in ONE launch
        (FM_OP_LITECONV, blockIdx.y = group): group i reads view xs[i] and writes channels
        [i*c, (i+1)*c) of the returned view.  c == xs[i].c <= 128 and c % 8 == 0 when G > 1."""
        G, c = len(xs), xs[0].c
        assert 1 <= G <= 4 and c <= 128 and all(x.c == c and x.coff % 8 == 0 and (x.h, x.w) == (xs[0].h, xs[0].w) for x in xs)
        assert G == 1 or c % 8 == 0
        if dst is None:
            dst = self.new(xs[0].h, xs[0].w, G * c)
        assert dst.c == G * c
        gates = []
        if gap_slot:       # group 0's per-tile channel sums go to a gate slot (input of gated_sum(parts=...))
            gates = [self.n_gates]
            self.n_gates += 1
            self.gate_c = max(self.gate_c, xs[0].cpad)
        self.last_gap_slot = gates[0] if gates else None
        self._layer(op=OP_LITECONV, ins=list(xs), out=dst, cin=xs[0].cpad, cout=c, k=3, stride=1, pad=1, act=ACT[act],
                    gates=gates,
                    w_off=self._push(np.stack([p['pw'] for p in params])),
                    w2_off=self._push(np.stack([p['dw'] for p in params])),
                    b_off=self._push(np.stack([p['bias'] for p in params])), name=name,
                    lite_ref=[p['ref'] for p in params])
        return dst

    @staticmethod
    def lightchain_fits(c, h, w):
        """The chain kernel keeps two halo tiles of a block in LDS (litechain.hip: litechain_lds_bytes)."""
        if c % 8 or not 8 <= c <= 128:
            return False
        nt = (c + 31) // 32
        th, tw = (16, 16 if w > 8 else 8) if nt == 1 else ((8, 16) if w > 8 else (16, 8))
        s = c + (0 if (c >> 3) & 1 else 8)
        ys = max((th + 8) * (tw + 8) * s, (256 // (c // 8)) * c * 2)
        return (ys + (th + 6) * (tw + 6) * s + 4 * 9 * 32 * nt + 4 * 2 * 32 * nt) * 2 <= 64 * 1024

    def lightchain(self, name, x, params, act='relu', dst=None):
        """The four streams of an OSNet block -- chains of 1, 2, 3, 4 LightConv3x3 over x -- in ONE launch
        (FM_OP_LITECHAIN).  params: the 10 lightconv_params() sets in (stream, level) order.  Returns the
        4*c-channel view (stream s at [s*c, (s+1)*c)); self.last_gap_slots = the streams' GAP partial slots."""
        c = x.c
        assert len(params) == 10 and x.coff % 8 == 0 and self.lightchain_fits(c, x.h, x.w)
        if dst is None:
            dst = self.new(x.h, x.w, 4 * c)
        assert dst.c == 4 * c
        gates = list(range(self.n_gates, self.n_gates + 4))
        self.n_gates += 4
        self.gate_c = max(self.gate_c, x.cpad)
        self.last_gap_slots = gates
        self._layer(op=OP_LITECHAIN, ins=[x], out=dst, cin=x.cpad, cout=c, k=3, stride=1, pad=1, act=ACT[act],
                    gates=gates, w_off=self._push(np.stack([p['pw'] for p in params])),
                    w2_off=self._push(np.stack([p['dw'] for p in params])),
                    b_off=self._push(np.stack([p['bias'] for p in params])), name=name,
                    lite_ref=[p['ref'] for p in params])
        return dst

    def lightconv(self, name, x, cout, act='relu', fuse=True):
        """torchreid LightConv3x3: 1x1 conv (linear, bias=False) -> depthwise 3x3 (bias=False) -> BN -> act.
        Fused into one FM_OP_LITECONV launch when cin == cout <= 128, else two layers; both forms
        draw the same parameters in the same order."""
        if not (fuse and x.c == cout and cout <= 128 and x.coff % 8 == 0):
            y = self.conv(name + '.pw', x, cout, 1, 1, 'linear', bn=False, bias=False)
            return self.dwconv3(name + '.dw', y, act)
        return self.lightconv_group(name, [x], [self.lightconv_params(name, cout)], act)

    def pool(self, x, k, stride, pad, avg=False, dst=None, pad_end=None):
        """pad_end: padding at the bottom/right when it differs from `pad` (ONNX SAME_UPPER of darknet
        [maxpool], yolo2onnx.py:838-863); windows are clipped at the border either way."""
        pad_end = pad if pad_end is None else pad_end
        assert avg is False or pad_end == pad
        ho = (x.h + pad + pad_end - k) // stride + 1
        wo = (x.w + pad + pad_end - k) // stride + 1
        if dst is None:
            dst = self.new(ho, wo, x.c)
        assert dst.h == ho and dst.w == wo
        self._layer(op=OP_AVGPOOL if avg else OP_MAXPOOL, ins=[x], out=dst, cin=x.cpad, cout=x.c, k=k,
                    stride=stride, pad=pad, pad_end=pad_end)
        return dst

    def spp(self, x, dst):
        """Stride-1 max pools k = 13, 9, 5 of x into dst channels [0,c), [c,2c), [2c,3c) (dst: a view of
        3*x.c channels).  One fused launch when the map fits LDS, else three pool layers."""
        c = x.c
        assert dst.c == 3 * c and dst.h == x.h and dst.w == x.w and c % 8 == 0
        if x.h * x.w > SPP_MAX_HW:
            for i, k in enumerate((13, 9, 5)):
                self.pool(x, k, 1, k // 2, dst=dst.slice(i * c, c))
            return dst
        self._layer(op=OP_SPP, ins=[x], out=dst, cin=x.cpad, cout=c, k=5, stride=1, pad=2)
        return dst

    def upsample2(self, x, dst=None):
        if dst is None:
            dst = self.new(2 * x.h, 2 * x.w, x.c)
        self._layer(op=OP_UPSAMPLE2, ins=[x], out=dst, cin=x.cpad, cout=x.c)
        return dst

    def add(self, a, b, dst=None):
        assert a.c == b.c and (a.h, a.w) == (b.h, b.w)
        if dst is None:
            dst = self.new(a.h, a.w, a.c)
        self._layer(op=OP_ADD, ins=[a, b], out=dst, cin=a.cpad, cout=a.c)
        return dst

    def copy(self, x, dst):
        self._layer(op=OP_COPY, ins=[x], out=dst, cin=x.cpad, cout=x.c)
        return dst

    def gate(self, name, x, hid, gate_params=None):
        """OSNet ChannelGate weights are shared by the four streams of a block: pass the dict
        returned by the first call as gate_params to reuse the blob offsets."""
        c = x.c
        if gate_params is None:
            p1 = self.wsrc.conv(name + '.fc1', hid, c, 1, bn=False)
            p2 = self.wsrc.conv(name + '.fc2', c, hid, 1, bn=False)
            w1 = np.zeros((hid, x.cpad), np.float16)
            w1[:, :c] = p1['w'].reshape(hid, c).astype(np.float16)
            w2 = np.zeros((x.cpad, hid), np.float16)
            w2[:c] = p2['w'].reshape(c, hid).astype(np.float16)
            b2 = np.zeros(x.cpad, np.float32)
            b2[:c] = p2['bias']
            gate_params = dict(w_off=self._push(w1), b_off=self._push(p1['bias'].astype(np.float32)),
                               w2_off=self._push(w2), b2_off=self._push(b2),
                               ref=(w1[:, :c].astype(np.float32), p1['bias'], w2[:c].astype(np.float32), p2['bias']))
        gid = self.n_gates
        self.n_gates += 1
        self.gate_c = max(self.gate_c, x.cpad)
        self._layer(op=OP_GATE, ins=[x], out=x, cin=x.cpad, cout=c, hid=hid, gates=[gid], name=name,
                    gate_ref=gate_params['ref'],
                    **{k: gate_params[k] for k in ('w_off', 'b_off', 'w2_off', 'b2_off')})
        return gid, gate_params

    def gated_sum(self, name, xs, hid, dst=None, parts=None):
        """OSNet unified aggregation gate fused: out = sum_i xs[i] * ChannelGate(xs[i]) with one shared
        gate (fc1: c -> hid, ReLU, fc2: hid -> c, sigmoid) -- FM_OP_GATED_SUM, one launch."""
        x = xs[0]
        c = x.c
        assert 1 <= len(xs) <= 4 and all(v.c == c and v.coff % 8 == 0 for v in xs)
        p1 = self.wsrc.conv(name + '.fc1', hid, c, 1, bn=False)
        p2 = self.wsrc.conv(name + '.fc2', c, hid, 1, bn=False)
        w1 = np.zeros((hid, x.cpad), np.float16)
        w1[:, :c] = p1['w'].reshape(hid, c).astype(np.float16)
        w2 = np.zeros((x.cpad, hid), np.float16)
        w2[:c] = p2['w'].reshape(c, hid).astype(np.float16)
        b2 = np.zeros(x.cpad, np.float32)
        b2[:c] = p2['bias']
        if dst is None:
            dst = self.new(x.h, x.w, c)
        assert parts is None or len(parts) == len(xs)    # gate slots holding the producers' channel sums
        self._layer(op=OP_GATED_SUM, ins=list(xs), out=dst, cin=x.cpad, cout=c, hid=hid, name=name,
                    gates=list(parts) if parts is not None else [],
                    w_off=self._push(w1), b_off=self._push(p1['bias'].astype(np.float32)),
                    w2_off=self._push(w2), b2_off=self._push(b2),
                    gate_ref=(w1[:, :c].astype(np.float32), p1['bias'], w2[:c].astype(np.float32), p2['bias']))
        return dst

    def gate_sum(self, xs, gids, dst=None):
        x = xs[0]
        if dst is None:
            dst = self.new(x.h, x.w, x.c)
        self._layer(op=OP_GATE_SUM, ins=list(xs), out=dst, cin=x.cpad, cout=x.c, gates=list(gids))
        return dst

    def head(self, name, x, dim):
        p = self.wsrc.linear(name, dim, x.c, bn=True)
        w, b = fold_bn(p)
        w16 = np.zeros((dim, x.cpad), np.float16)
        w16[:, :x.c] = w.astype(np.float16)
        dst = View(x.tid, x.coff, x.c, x.h, x.w)   # output goes to the ctx embedding buffer
        self._layer(op=OP_HEAD, ins=[x], out=dst, cin=x.cpad, cout=dim, w_off=self._push(w16),
                    b_off=self._push(b.astype(np.float32)), name=name,
                    head_ref=(w16[:, :x.c].astype(np.float32), b))
        return dst

    # ---------------------------------------------------------------- C tables
    def plan_arena(self, max_batch, reuse):
        """Byte offsets of the tensors in one activation arena.  With `reuse`, tensors whose live
        ranges [first writer/reader layer, last layer touching them] do not overlap share bytes
        (greedy first-fit by decreasing size).  The network input and the graph outputs are read /
        written outside the layer sequence and therefore stay live for the whole run."""
        n = len(self.tensors)
        size = [ceil_to(max_batch * h * w * c * (4 if f32 else 2), 256) for (h, w, c, f32) in self.tensors]
        first, last = [10**9] * n, [-1] * n
        for li, d in enumerate(self.layers):
            touched = [v.tid for v in d['ins']] + [d['out'].tid] + ([d['res'].tid] if d['res'] is not None else [])
            for t in touched:
                first[t], last[t] = min(first[t], li), max(last[t], li)
        persistent = {self.input.tid} | {v.tid for v in self.outputs}
        for t in range(n):
            if t in persistent or not reuse or last[t] < 0:
                first[t], last[t] = -1, 10**9

        def disjoint(t, u):
            """all accesses of t are over before u is touched, or the other way round"""
            if first[t] < 0 or first[u] < 0:
                return False
            return last[t] < first[u] or last[u] < first[t]
        offsets = [0] * n
        placed = []                      # (offset, size, tensor)
        for t in sorted(range(n), key=lambda i: -size[i]):
            cands = sorted((o, s) for (o, s, u) in placed if not disjoint(t, u))
            off = 0
            for o, s in cands:
                if off + size[t] <= o:
                    break
                off = max(off, o + s)
            offsets[t] = off
            placed.append((off, size[t], t))
        total = max((o + s for (o, s, _) in placed), default=0)
        return offsets, total

    def tables(self, max_batch=1, reuse=False):
        self.check_fusions()
        offsets, arena = self.plan_arena(max_batch, reuse)
        self.arena_bytes = arena
        ts = (fm_tensor * len(self.tensors))()
        for i, (h, w, c, f32) in enumerate(self.tensors):
            ts[i] = fm_tensor(h, w, c, f32, offsets[i])
        ls = (fm_layer * len(self.layers))()
        for i, d in enumerate(self.layers):
            L = fm_layer()
            L.op = d['op']
            L.n_in = len(d['ins'])
            for j, v in enumerate(d['ins']):
                L.in_[j] = v.tid
                L.in_coff[j] = v.coff
            L.out, L.out_coff = d['out'].tid, d['out'].coff
            if d['res'] is not None:
                L.res, L.res_coff = d['res'].tid, d['res'].coff
            else:
                L.res, L.res_coff = -1, 0
            L.res_mode = d['res_mode']
            for key in ('cin', 'cout', 'k', 'stride', 'pad', 'act', 'hid', 'up', 'w_off', 'b_off', 'w2_off', 'b2_off'):
                setattr(L, key, d[key])
            for j in range(4):
                L.gate[j] = d['gates'][j] if j < len(d['gates']) else -1
            ls[i] = L
        blob = bytes(self.blob) + b'\0' * 64
        return ts, ls, blob

    def conv_flops(self, batch=1):
        """2*MAC over conv layers (the 'conv roofline' numerator)."""
        total = 0
        for d in self.layers:
            if d['op'] in CONV_OPS + (OP_STEMCONV,):
                o = d['out']
                total += 2 * d['k'] * d['k'] * d['ins'][0].c * d['cout'] * o.h * o.w * batch
            elif d['op'] == OP_RESBLOCK:
                o = d['out']
                total += 2 * 10 * d['cin'] * d['hid'] * o.h * o.w * batch
            elif d['op'] == OP_PAIR11:
                o = d['out']
                total += (2 * d['cin'] * d['hid'] + 2 * (d['hid'] + d['cin']) * d['cout']) * o.h * o.w * batch
            elif d['op'] == OP_STEM2:
                o, i = d['out'], d['ins'][0]
                c2 = d['gates'][1] if len(d['gates']) > 1 else d['cout']
                total += (2 * 9 * i.c * d['hid'] * i.h * i.w + 2 * 9 * d['hid'] * c2 * o.h * o.w) * batch
                if len(d['gates']) > 1:
                    total += 2 * c2 * d['cout'] * o.h * o.w * batch
        return total
