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
 OP_LITECONV, OP_SPP, OP_GATED_SUM, OP_STEMCONV, OP_ADD, OP_RESBLOCK, OP_CONVS, OP_LITECHAIN, OP_CSPSTAGE,
 OP_GATEDCONV) = range(19)
SPP_MAX_HW = 2048
ACT = {'linear': 0, 'leaky': 1, 'mish': 2, 'relu': 3, 'logistic': 4, 'swish': 5}
RES_NONE, RES_AFTER_ACT, RES_BEFORE_ACT, RES_CONCAT = 0, 1, 2, 3


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
                ('w_off', C.c_int64), ('b_off', C.c_int64), ('w2_off', C.c_int64), ('b2_off', C.c_int64),
                ('branch', C.c_int32), ('wait_for', C.c_int32), ('signal', C.c_int32), ('cin2', C.c_int32)]


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
        # the first CSP stage of CSPDarknet53 (4 launches on the 304 x 304 map) as one launch (cspstage.hip): bit-identical,
        # measured SLOWER (73 us against 59 us, profiles/r04_cspstage_ab.txt) -> off unless FASTMOT_CSPSTAGE=1
        self.use_cspstage = os.environ.get('FASTMOT_CSPSTAGE', '0') == '1'
        # darknet residual units (1x1, 3x3, shortcut) as one fused launch (resblock.hip)
        self.use_resblock = os.environ.get('FASTMOT_RESBLOCK', '1') != '0'
        # the four LightConv streams of an OSNet block as one launch (litechain.hip) instead of one per depth
        self.use_lightchain = os.environ.get('FASTMOT_LITECHAIN', '1') != '0'
        # tail of an OSNet block (gate, gated sum, conv3 + shortcut) as one launch (gatedconv.hip): measured slower than
        # the two launches it replaces (profiles/r04_gatedconv_ab.txt) -- off unless FASTMOT_GATEDCONV=1
        self.use_gatedconv = os.environ.get('FASTMOT_GATEDCONV', '0') == '1'
        # convs with at most this many output pixels per sample and a long reduction take the streamed
        # kernel (K split inside the workgroup, convs.hip) instead of the LDS-tiled one + split-K reduce
        self.convs_max_pixels = int(os.environ.get('FASTMOT_CONVS_MAXP', '1444'))
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
                 act=0, hid=0, up=1, gates=[], w_off=0, b_off=0, w2_off=0, b2_off=0, cin2=0)
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
        if x.c == cin_pad and cin_pad % 64 == 0 and k * k * cin_pad >= 512 and ho * wo <= self.convs_max_pixels:
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

    @staticmethod
    def resblock_supported(c, mid):
        return c in (64, 128, 256) and mid in (c, c // 2)

    @staticmethod
    def _pack_frag(w16):
        """[cout, cin, k, k] -> MFMA A-fragment order [cout/32][K/16][lane][8], K order (kh, kw, cin),
        lane = (k / 8 % 2) * 32 + cout % 32 (cout % 32 == 0, K % 16 == 0)."""
        cout, cin, k, _ = w16.shape
        K = k * k * cin
        assert cout % 32 == 0 and K % 16 == 0
        rows = w16.transpose(0, 2, 3, 1).reshape(cout // 32, 32, K // 16, 2, 8)
        return np.ascontiguousarray(rows.transpose(0, 2, 3, 1, 4))

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

    @staticmethod
    def cspstage_supported(c, mid):
        return c == 64 and mid == 32

    def cspstage(self, names, d, mid, cout, act='mish', dst=None):
        """First CSP stage of CSPDarknet53 behind its stride-2 conv `d`, in one launch (cspstage.hip):
        [b | A] = act(1x1 d) ; b' = b + act(3x3 act(1x1 b)) ; c = act(1x1 b') ; out = act(1x1 [c | A]).
        names: the six Darknet layers in cfg order (route branch A, residual branch b, bottleneck 1x1, 3x3, post 1x1,
        stage output 1x1) -- parameters are drawn in that order, like the unfused layers."""
        h = d.c
        assert self.cspstage_supported(h, mid) and d.cpad == h and cout == h
        if dst is None:
            dst = self.new(d.h, d.w, cout)
        (wa, ba), (wb, bb) = (fold_bn(self.wsrc.conv(n, h, h, 1, bn=True)) for n in names[:2])
        w3a, b3a = fold_bn(self.wsrc.conv(names[2], mid, h, 1, bn=True))
        w3b, b3b = fold_bn(self.wsrc.conv(names[3], h, mid, 3, bn=True))
        w4, b4 = fold_bn(self.wsrc.conv(names[4], h, h, 1, bn=True))
        w5, b5 = fold_bn(self.wsrc.conv(names[5], cout, 2 * h, 1, bn=True))
        w2, b2 = np.concatenate([wb, wa]), np.concatenate([bb, ba])          # [b | A], as the merged sibling conv writes them
        mats = [np.asarray(w, np.float32).astype(np.float16) for w in (w2, w3a, w3b, w4, w5)]
        blob = np.concatenate([self._pack_frag(w).reshape(-1) for w in mats])
        bias = np.concatenate([np.asarray(b, np.float32) for b in (b2, b3a, b3b, b4, b5)])
        self._layer(op=OP_CSPSTAGE, ins=[d], out=dst, cin=h, cout=cout, k=3, stride=1, pad=1, act=ACT[act], hid=mid,
                    w_off=self._push(blob), b_off=self._push(bias), name=names[5],
                    csp_ref=tuple((w.astype(np.float32), np.asarray(b, np.float32))
                                  for w, b in zip(mats, (b2, b3a, b3b, b4, b5))))
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

    def gated_conv(self, name, conv_name, xs, hid, parts, cout, act='relu', x2=None, res=None, wb=None, dst=None):
        """Tail of an OSNet block in one launch (FM_OP_GATEDCONV, gatedconv.hip): act(conv1x1([gated_sum(xs) | x2]) (+ res)).
        xs: the four stream views with their tile-sum slots `parts`; x2: second K segment (a stage's first block: the
        block input, `wb` = ([W3 | Wd], b3 + bd)); res: identity shortcut, added before the activation."""
        x = xs[0]
        c = x.c
        assert len(xs) == 4 and len(parts) == 4 and all(v.c == c and v.coff % 8 == 0 for v in xs) and c % 8 == 0
        assert x2 is None or res is None
        c2 = x2.c if x2 is not None else 0
        assert c2 % 8 == 0 and cout % 8 == 0
        p1 = self.wsrc.conv(name + '.fc1', hid, c, 1, bn=False)
        p2 = self.wsrc.conv(name + '.fc2', c, hid, 1, bn=False)
        w1 = p1['w'].reshape(hid, c).astype(np.float16)
        w2 = p2['w'].reshape(c, hid).astype(np.float16)
        b1, b2 = p1['bias'].astype(np.float32), p2['bias'].astype(np.float32)
        blob = bytearray()
        for a in (w1, b1, w2, b2):                        # sections 16 B aligned (fastmot_hip.h: FM_OP_GATEDCONV)
            blob += np.ascontiguousarray(a).tobytes()
            blob += b'\0' * (-len(blob) % 16)
        if wb is not None:      # (a callable: evaluated here, after the gate's parameters, like the two-launch tail reads them)
            w, b = (np.asarray(a, np.float32) for a in (wb() if callable(wb) else wb))
        else:
            w, b = fold_bn(self.wsrc.conv(conv_name, cout, c, 1, bn=True))
        K = c + c2
        assert w.shape == (cout, K, 1, 1) and b.shape == (cout,)
        w16 = w.astype(np.float16)
        cpad = ceil_to(cout, 32)
        c16 = ceil_to(c, 16)                              # two K segments of whole 16-channel steps (gatedconv.hip)
        packed = np.zeros((cpad, ceil_to(c16 + ceil_to(c2, 16), 64)), np.float16)
        packed[:cout, :c] = w16.reshape(cout, K)[:, :c]
        packed[:cout, c16:c16 + c2] = w16.reshape(cout, K)[:, c:]
        bias = np.zeros(cpad, np.float32)
        bias[:cout] = b
        if dst is None:
            dst = self.new(x.h, x.w, cout)
        self._layer(op=OP_GATEDCONV, ins=list(xs), out=dst, cin=c, cout=cout, hid=hid, act=ACT[act], cin2=c2,
                    gates=list(parts), res=x2 if x2 is not None else res,
                    res_mode=RES_CONCAT if x2 is not None else (RES_BEFORE_ACT if res is not None else RES_NONE),
                    w_off=self._push(packed), b_off=self._push(bias),
                    w2_off=self._push(np.frombuffer(bytes(blob), np.uint8)), name=name,
                    gate_ref=(w1.astype(np.float32), p1['bias'], w2.astype(np.float32), p2['bias']),
                    conv_ref=(w16.astype(np.float32), b))
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
    def plan_arena(self, max_batch, reuse, before=None):
        """Byte offsets of the tensors in one activation arena.  With `reuse`, tensors whose live
        ranges [first writer/reader layer, last layer touching them] do not overlap share bytes
        (greedy first-fit by decreasing size).  The network input and the graph outputs are read /
        written outside the layer sequence and therefore stay live for the whole run.
        before (two-branch schedules, plan_branches): before[b] = set of layers that are complete when layer b starts;
        two tensors then share bytes only if every access of one happens before every access of the other (table order
        alone does not say so any more)."""
        n = len(self.tensors)
        size = [ceil_to(max_batch * h * w * c * (4 if f32 else 2), 256) for (h, w, c, f32) in self.tensors]
        first, last = [10**9] * n, [-1] * n
        acc = [[] for _ in range(n)]
        for li, d in enumerate(self.layers):
            touched = [v.tid for v in d['ins']] + [d['out'].tid] + ([d['res'].tid] if d['res'] is not None else [])
            for t in touched:
                first[t], last[t] = min(first[t], li), max(last[t], li)
                if not acc[t] or acc[t][-1] != li:
                    acc[t].append(li)
        persistent = {self.input.tid} | {v.tid for v in self.outputs}
        for t in range(n):
            if t in persistent or not reuse or last[t] < 0:
                first[t], last[t] = -1, 10**9

        def disjoint(t, u):
            """all accesses of t are over before u is touched, or the other way round"""
            if first[t] < 0 or first[u] < 0:
                return False
            if before is None:
                return last[t] < first[u] or last[u] < first[t]
            return all(a in before[b] for a in acc[t] for b in acc[u]) or all(b in before[a] for a in acc[t] for b in acc[u])
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

    @staticmethod
    def happens_before(plan):
        """before[b] = layers guaranteed complete when layer b starts under a two-branch plan: the earlier layers of its
        own branch, the layer it waits for, and everything before those."""
        before = []
        prev = [-1, -1]
        for i, (br, wait, _) in enumerate(plan):
            s = set()
            for p in (prev[br], wait):
                if p >= 0:
                    s |= before[p]
                    s.add(p)
            before.append(s)
            prev[br] = i
        return before

    def plan_branches(self, max_batch, offsets=None):
        """Two-stream schedule of the layer sequence: -> [(branch, wait_for, signal)] per layer (fm_layer).

        The table order is one valid serial order; batch-1 networks leave most of the GPU idle in their small layers, and
        some of them do not depend on each other: a YOLO head's 3x3 + 1x1 and the PAN path that continues from the same
        tensor, the two 1x1 convs that fill the halves of a concat.  Dependencies are derived from MEMORY, not from the
        table's tensor ids alone: layer b (later) depends on layer a if one writes what the other reads or writes --
        same tensor and intersecting channel ranges, or (with the shared arena) different tensors whose byte ranges
        intersect.  Layers are then list-scheduled in table order onto two branches with a rough duration model
        (launch floor + FLOPs + bytes); every branch keeps table order, so hazards inside a branch are ordered by its
        stream, and a cross-branch dependency becomes one event wait.  Results are bit-identical to the chain: the same
        kernels on the same data."""
        n = len(self.layers)
        size = [max_batch * h * w * c * (4 if f32 else 2) for (h, w, c, f32) in self.tensors]

        def foot(d):
            reads = [(v.tid, v.coff, v.coff + v.cpad) for v in d['ins']]
            if d['res'] is not None:
                reads.append((d['res'].tid, d['res'].coff, d['res'].coff + d['res'].cpad))
            o = d['out']
            return reads, [(o.tid, o.coff, o.coff + max(o.cpad, ceil_to(d.get('cout', 0) or o.c, 8)))]

        def clash(x, y):
            if x[0] == y[0]:
                return x[1] < y[2] and y[1] < x[2]
            if offsets is None:
                return False
            ax, ay = offsets[x[0]], offsets[y[0]]
            return ax < ay + size[y[0]] and ay < ax + size[x[0]]
        feet = [foot(d) for d in self.layers]
        gate_users = {}
        deps = [set() for _ in range(n)]
        for b in range(n):
            rb, wb = feet[b]
            for a in range(b):
                ra, wa = feet[a]
                if any(clash(x, y) for x in wb for y in ra + wa) or any(clash(x, y) for x in rb for y in wa):
                    deps[b].add(a)
            for gslot in self.layers[b]['gates']:                # gate / pool slots are shared scratch: keep their users ordered
                if gslot >= 0:
                    deps[b].update(gate_users.get(gslot, ()))
                    gate_users.setdefault(gslot, []).append(b)
        flops = self.layer_flops(max_batch) if hasattr(self, 'layer_flops') else None

        def est(i):
            d = self.layers[i]
            r, w = feet[i]
            by = sum((c1 - c0) * self.tensors[t][0] * self.tensors[t][1] * 2 for t, c0, c1 in r + w) * max_batch
            k = d.get('k', 1) or 1
            fl = 2.0 * k * k * (d.get('cin', 0) or 0) * (d.get('cout', 0) or 0) * d['out'].h * d['out'].w * max_batch
            return 4e-6 + fl / 200e12 + by / 2e12
        finish, branch = [0.0] * n, [0] * n
        free = [0.0, 0.0]
        for i in range(n):
            best = None
            for s in (0, 1):
                start = free[s]
                for j in deps[i]:
                    start = max(start, finish[j] + (1.5e-6 if branch[j] != s else 0.0))
                if best is None or start < best[0] - 2e-6:       # the side branch has to win by more than a launch gap
                    best = (start, s)
            branch[i] = best[1]
            finish[i] = best[0] + est(i)
            free[best[1]] = finish[i]
        if n:
            branch[n - 1] = 0 if all(b == 0 for b in branch[:-1]) else branch[n - 1]
        wait, signal = [-1] * n, [0] * n
        waited = [-1, -1]                                        # per branch: newest layer of the other branch already waited for
        for i in range(n):
            s = branch[i]
            other = [j for j in deps[i] if branch[j] != s]
            if other and max(other) > waited[s]:
                wait[i] = max(other)
                signal[wait[i]] = 1
                waited[s] = wait[i]
        return list(zip(branch, wait, signal))

    def tables(self, max_batch=1, reuse=False, branches=False):
        plan = self.plan_branches(max_batch) if branches else None
        if plan is not None and not any(b for b, _, _ in plan):
            plan = None                                          # nothing to run side by side: one chain
        self.branch_plan = plan
        offsets, arena = self.plan_arena(max_batch, reuse, self.happens_before(plan) if plan is not None else None)
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
            for key in ('cin', 'cout', 'k', 'stride', 'pad', 'act', 'hid', 'up', 'w_off', 'b_off', 'w2_off', 'b2_off', 'cin2'):
                setattr(L, key, d[key])
            for j in range(4):
                L.gate[j] = d['gates'][j] if j < len(d['gates']) else -1
            L.branch, L.wait_for, L.signal = plan[i] if plan is not None else (0, -1, 0)
            ls[i] = L
        blob = bytes(self.blob) + b'\0' * 64
        return ts, ls, blob

    def conv_flops(self, batch=1):
        """2*MAC over conv layers (the 'conv roofline' numerator)."""
        total = 0
        for d in self.layers:
            if d['op'] in (OP_CONV, OP_STEMCONV, OP_CONVS):
                o = d['out']
                total += 2 * d['k'] * d['k'] * d['ins'][0].c * d['cout'] * o.h * o.w * batch
            elif d['op'] == OP_GATEDCONV:
                o = d['out']
                total += 2 * (d['cin'] + d['cin2']) * d['cout'] * o.h * o.w * batch
            elif d['op'] == OP_RESBLOCK:
                o = d['out']
                total += 2 * 10 * d['cin'] * d['hid'] * o.h * o.w * batch
            elif d['op'] == OP_CSPSTAGE:
                o, c, m = d['out'], d['cin'], d['hid']
                total += 2 * (2 * c * c + c * m + 9 * m * c + c * c + 2 * c * d['cout']) * o.h * o.w * batch
        return total
