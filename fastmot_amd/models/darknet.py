"""Darknet `.cfg` / `.weights` -> layer table of the HIP conv engine.

Replaces scripts/yolo2onnx.py + the TensorRT engine build of the reference (DarkNetParser
yolo2onnx.py:86-205, WeightLoader :283-400, GraphBuilderONNX :403-870): any cfg made of
[convolutional] / [maxpool] / [route] / [shortcut] / [upsample] / [yolo] sections (YOLOv3/v4, -tiny,
Scaled-YOLOv4 CSP / P5 / P6) becomes a `Graph`, and the `[yolo]` sections become the descriptor
attributes YOLODetector needs (anchors, strides, scale_x_y, new_coords, classes).

Engine-specific lowering (each keeps the yolo2onnx semantics, tests/test_darknet.py checks it against an
independent PyTorch interpretation of the same cfg):
  * [shortcut] folds into the epilogue of the conv right before it when nothing else reads that conv,
    otherwise FM_OP_ADD;
  * [route] with one source is an alias (`groups` / `group_id`: a channel-slice view); with several
    sources the producers write their slices of one concat tensor in place when they can, else FM_OP_COPY;
  * [upsample] (stride 2) folds into the conv that feeds it (conv `up=2`) when nothing else reads it;
  * three stride-1 [maxpool] 5/9/13 of one tensor concatenated in the order 13, 9, 5 (the SPP block)
    become one FM_OP_SPP launch;
  * the conv before a [yolo] section is an fp32 head.
"""
import io
import os
import re
from pathlib import Path

import numpy as np

from .graph import Graph, SPP_MAX_HW, fold_bn

SUPPORTED = ('net', 'convolutional', 'maxpool', 'shortcut', 'route', 'upsample', 'yolo')


def parse_cfg(text):
    """-> list of dicts (one per section, 'type' key), values int / float / str / list like
    DarkNetParser._parse_params (yolo2onnx.py:178-205).  [yolo] parameters are kept (the reference
    drops them and hard-codes the anchors per model class instead)."""
    layers = []
    cur = None
    for raw in text.splitlines():
        line = raw.split('#', 1)[0].strip()
        if not line:
            continue
        m = re.fullmatch(r'\[(\w+)\]', line)
        if m:
            if m.group(1) not in SUPPORTED:
                raise ValueError(f'{m.group(1)} layer not supported!')
            cur = dict(type=m.group(1))
            layers.append(cur)
            continue
        if cur is None or '=' not in line:
            raise ValueError(f'cannot parse cfg line: {raw!r}')
        key, val = (s.strip() for s in line.split('=', 1))
        cur[key] = _value(val, as_list=key in ('layers', 'mask', 'anchors', 'steps', 'scales'))
    if not layers or layers[0]['type'] != 'net':
        raise ValueError('cfg must start with a [net] section')
    return layers


def _value(val, as_list=False):
    def one(s):
        s = s.strip()
        try:
            return int(s)
        except ValueError:
            try:
                return float(s)
            except ValueError:
                return s
    if as_list or ',' in val:
        return [one(s) for s in val.split(',') if s.strip()]
    return one(val)


class DarknetWeights:
    """Sequential reader of a Darknet weights file (yolo2onnx.py:341-400): 5 x int32 header, then per
    [convolutional] section in cfg order: batch-norm beta, gamma, mean, var (or the conv bias), then the
    weights [filters, cin / groups, k, k], all float32."""

    def __init__(self, source):
        data = Path(source).read_bytes() if isinstance(source, (str, Path)) else bytes(source)
        self.buf = io.BytesIO(data)
        self.size = len(data)
        self.header = np.frombuffer(self.buf.read(20), np.int32)
        if len(self.header) != 5:
            raise ValueError('truncated Darknet weights header')

    def _read(self, n):
        raw = self.buf.read(4 * n)
        if len(raw) != 4 * n:
            raise ValueError('Darknet weights file ends before the cfg does')
        return np.frombuffer(raw, np.float32).copy()

    def conv(self, name, cout, cin, k, bn=True, gain=1.0, groups=1):
        if bn:
            beta, gamma, mean, var = (self._read(cout) for _ in range(4))
            p = dict(gamma=gamma, beta=beta, mean=mean, var=var)
        else:
            p = dict(bias=self._read(cout))
        p['w'] = self._read(cout * (cin // groups) * k * k).reshape(cout, cin // groups, k, k)
        return p

    def remaining(self):
        return self.size - self.buf.tell()


def _resolve(i, ref):
    """cfg layer index (0 = first section after [net]) of a relative / absolute reference."""
    return i + ref if ref < 0 else ref


def darknet_graph(cfg, weights, in_hw=None):
    """cfg: parse_cfg() output (or cfg text); weights: DarknetWeights or any object with the
    RandomWeights.conv interface.  -> (Graph, [head views in cfg order], meta) with
    meta = dict(input_shape, classes, anchors, strides, scales, new_coords)."""
    if isinstance(cfg, str):
        cfg = parse_cfg(cfg)
    net, layers = cfg[0], cfg[1:]
    H, W = in_hw if in_hw is not None else (int(net['height']), int(net['width']))
    cin0 = int(net.get('channels', 3))
    n = len(layers)

    # ---- pass 1: shapes (c, h, w) and readers of every layer
    shape = [None] * n
    readers = [set() for _ in range(n)]

    def src_shape(i):
        return (cin0, H, W) if i < 0 else shape[i]

    for i, L in enumerate(layers):
        t = L['type']
        if t == 'convolutional':
            c, h, w = src_shape(i - 1)
            if i > 0:
                readers[i - 1].add(i)
            k, s = int(L.get('size', 1)), int(L.get('stride', 1))
            if int(L.get('groups', 1)) != 1:
                raise NotImplementedError('grouped [convolutional] layers')
            p = k // 2 if int(L.get('pad', 0)) else int(L.get('padding', 0))
            shape[i] = (int(L['filters']), (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1)
        elif t == 'maxpool':
            c, h, w = src_shape(i - 1)
            readers[i - 1].add(i)
            s = int(L.get('stride', 1))
            shape[i] = (c, -(-h // s), -(-w // s))               # auto_pad SAME_UPPER
        elif t == 'upsample':
            c, h, w = src_shape(i - 1)
            readers[i - 1].add(i)
            s = int(L.get('stride', 2))
            shape[i] = (c, h * s, w * s)
        elif t == 'shortcut':
            assert L.get('activation', 'linear') == 'linear'
            j = _resolve(i, int(L['from'] if not isinstance(L['from'], list) else L['from'][0]))
            readers[i - 1].add(i)
            readers[j].add(i)
            assert shape[i - 1] == shape[j], f'shortcut {i}: {shape[i - 1]} vs {shape[j]}'
            shape[i] = shape[i - 1]
        elif t == 'route':
            srcs = [_resolve(i, int(r)) for r in L['layers']]
            for j in srcs:
                readers[j].add(i)
            if len(srcs) == 1:
                c, h, w = shape[srcs[0]]
                if 'groups' in L:
                    assert c % int(L['groups']) == 0 and int(L['group_id']) < int(L['groups'])
                    c //= int(L['groups'])
                shape[i] = (c, h, w)
            else:
                assert 'groups' not in L, 'groups not implemented for multiple-input route layer!'
                hw = {shape[j][1:] for j in srcs}
                assert len(hw) == 1, f'route {i}: operand sizes differ {hw}'
                shape[i] = (sum(shape[j][0] for j in srcs),) + shape[srcs[0]][1:]
        elif t == 'yolo':
            readers[i - 1].add(i)
            shape[i] = shape[i - 1]

    # ---- pass 2: concat planning.  A multi-source route owns a tensor; operand j is written in place by
    # its producer when the producer is a layer that materialises a fresh tensor and is not already an
    # operand of another concat.  `alias_of`: single-source routes without groups.
    def base(j):
        """Follow single-source non-group routes down to the layer that owns the data."""
        while layers[j]['type'] == 'route' and len(layers[j]['layers']) == 1 and 'groups' not in layers[j]:
            j = _resolve(j, int(layers[j]['layers'][0]))
        return j

    def folded_upsample(j):
        """Upsample layer j folds into the conv j-1 (only reader) -> the conv produces j's tensor."""
        return (layers[j]['type'] == 'upsample' and int(layers[j].get('stride', 2)) == 2 and j > 0 and
                layers[j - 1]['type'] == 'convolutional' and readers[j - 1] == {j} and
                not (j + 1 < n and layers[j + 1]['type'] == 'yolo'))

    def folded_shortcut(j):
        return (layers[j]['type'] == 'shortcut' and layers[j - 1]['type'] == 'convolutional' and
                readers[j - 1] == {j})

    def conv_attrs(j):
        L = layers[j]
        k = int(L.get('size', 1))
        return (k, int(L.get('stride', 1)), k // 2 if int(L.get('pad', 0)) else int(L.get('padding', 0)),
                L.get('activation', 'linear'), int(L.get('batch_normalize', 0)) == 1)

    def fused_resblock(j):
        """Shortcut j closes a residual unit conv1x1 (j-2) -> conv3x3 (j-1) -> + input of j-2 that the
        fused kernel covers (resblock.hip): one launch instead of three."""
        if not (use_resblock and j >= 3 and folded_shortcut(j) and layers[j - 2]['type'] == 'convolutional' and
                readers[j - 2] == {j - 1} and layers[j].get('activation', 'linear') == 'linear'):
            return False
        frm = layers[j]['from']
        if _resolve(j, int(frm[0] if isinstance(frm, list) else frm)) != j - 3:
            return False
        a, b = conv_attrs(j - 2), conv_attrs(j - 1)
        return (a[:3] == (1, 1, 0) and b[:3] == (3, 1, 1) and a[3] == b[3] and shape[j - 3][0] == shape[j][0] and
                Graph.resblock_supported(shape[j][0], shape[j - 2][0]) and
                g.resblock_pays(shape[j][0], shape[j - 2][0], shape[j][1], shape[j][2]))

    def producer(j):
        """Layer index whose emitted op writes the tensor of layer j, or None if j is a view."""
        j = base(j)
        t = layers[j]['type']
        if t in ('convolutional', 'maxpool'):
            return j
        if t == 'upsample':
            return j - 1 if folded_upsample(j) else j
        if t == 'shortcut':
            return j - 1 if folded_shortcut(j) else j
        return None if t == 'route' and len(layers[j]['layers']) == 1 else j    # group slice: view

    placed = {}          # producer layer -> (concat layer, channel offset)
    spp_groups = {}      # last pool layer of an SPP triple -> (source layer, concat layer, offset of k13)
    spp_members = set()
    for i, L in enumerate(layers):
        if L['type'] != 'route' or len(L['layers']) < 2:
            continue
        srcs = [_resolve(i, int(r)) for r in L['layers']]
        off = 0
        offs = []
        for j in srcs:
            offs.append(off)
            off += shape[j][0]
        # SPP: operands q, q+1, q+2 are stride-1 maxpools 13, 9, 5 of the same tensor, read by nothing else
        for q in range(len(srcs) - 2):
            trio = [base(j) for j in srcs[q:q + 3]]
            if not all(layers[j]['type'] == 'maxpool' and int(layers[j].get('stride', 1)) == 1 and
                       readers[j] == {i} and j not in placed and j not in spp_members for j in trio):
                continue
            if [int(layers[j]['size']) for j in trio] != [13, 9, 5] or len({base(j - 1) for j in trio}) != 1:
                continue
            if shape[trio[0]][0] % 8 or offs[q] % 8 or shape[trio[0]][1] * shape[trio[0]][2] > SPP_MAX_HW:
                continue
            spp_members.update(trio)
            spp_groups[max(trio)] = (base(trio[0] - 1), i, offs[q])
        for j, o in zip(srcs, offs):
            pj = producer(j)
            if pj is None or pj in placed or base(j) in spp_members or o % 8 or shape[j][0] % 8:
                continue
            if pj < i:
                placed[pj] = (i, o)

    # ---- pass 2b: sibling 1x1 convs of a CSP stage.  conv a (-> concat C at offset o) and conv b = a + 2 read
    # the same tensor (route -2 in between); when the concat slot right below a's is later filled by a plain
    # 1x1 conv P that runs after every reader of b, both run as ONE conv with stacked weights writing
    # [b | a] into C: the input is read once, one launch fewer; b's slot is dead by the time P overwrites it.
    sibling_of, deferred = {}, set()
    if os.environ.get('FASTMOT_CSP_MERGE', '1') != '0':
        slot_owner = {v: k for k, v in placed.items()}
        for a, (ci, oa) in sorted(placed.items()):
            b = a + 2
            if not (layers[a]['type'] == 'convolutional' and a >= 1 and b < n and
                    layers[b]['type'] == 'convolutional' and layers[a + 1]['type'] == 'route' and
                    len(layers[a + 1]['layers']) == 1 and 'groups' not in layers[a + 1] and
                    _resolve(a + 1, int(layers[a + 1]['layers'][0])) == a - 1):
                continue
            h = shape[a][0]
            pj = slot_owner.get((ci, oa - h))
            if (conv_attrs(a) != conv_attrs(b) or conv_attrs(a)[:3] != (1, 1, 0) or shape[b][0] != h or h % 8 or
                    b in placed or pj is None or layers[pj]['type'] != 'convolutional' or shape[pj][0] != h or
                    conv_attrs(pj)[:3] != (1, 1, 0) or base(pj - 1) == b or not all(r < pj for r in readers[b]) or
                    (pj + 1 < n and (folded_shortcut(pj + 1) or folded_upsample(pj + 1))) or
                    (b + 1 < n and (folded_shortcut(b + 1) or folded_upsample(b + 1) or layers[b + 1]['type'] == 'yolo')) or
                    (a + 1 < n and layers[a + 1]['type'] == 'yolo')):
                continue
            sibling_of[b] = a
            deferred.add(a)

    # ---- pass 3: emit
    g = Graph(weights, (H, W), cin0)
    use_resblock = g.use_resblock
    stash = {}
    out = [None] * n                      # View of every layer's output
    cat = {}                              # concat layer -> tensor view (allocated by its first producer)
    heads, meta_yolo = [], []

    def cat_view(ci):
        if ci not in cat:
            c, h, w = shape[ci]
            cat[ci] = g.new(h, w, c)
        return cat[ci]

    def dst_of(pj, c):
        if pj in placed:
            ci, o = placed[pj]
            return cat_view(ci).slice(o, c)
        return None

    def src(i):
        return g.input if i < 0 else out[i]

    for i, L in enumerate(layers):
        t = L['type']
        c, h, w = shape[i]
        if t == 'convolutional':
            k, s = int(L.get('size', 1)), int(L.get('stride', 1))
            p = k // 2 if int(L.get('pad', 0)) else int(L.get('padding', 0))
            act = L.get('activation', 'linear')
            bn = int(L.get('batch_normalize', 0)) == 1
            is_head = i + 1 < n and layers[i + 1]['type'] == 'yolo'
            res = None
            up = 1
            if i in deferred:               # sibling merge: parameters are read in file order, emitted with conv i + 2
                stash[i] = fold_bn(g.wsrc.conv(f'{i:03d}_convolutional', c, shape[i - 1][0] if i else cin0, k, bn=bn))
                ci, oa = placed[i]
                out[i] = cat_view(ci).slice(oa, c)
                continue
            if i in sibling_of:
                a = sibling_of[i]
                ci, oa = placed[a]
                wa, ba = stash.pop(a)
                wb_, bb = fold_bn(g.wsrc.conv(f'{i:03d}_convolutional', c, src(i - 1).c, k, bn=bn))
                g.conv(f'{i:03d}_convolutional', src(i - 1), 2 * c, k, s, act, pad=p, dst=cat_view(ci).slice(oa - c, 2 * c),
                       wb=(np.concatenate([wb_, wa]), np.concatenate([bb, ba])))
                out[i] = cat_view(ci).slice(oa - c, c)
                continue
            if i + 2 < n and layers[i + 2]['type'] == 'shortcut' and fused_resblock(i + 2):
                continue                    # the 1x1 of a fused residual unit: emitted with its 3x3
            if i + 1 < n and layers[i + 1]['type'] == 'shortcut' and fused_resblock(i + 1):
                out[i] = g.resblock(f'{i - 1:03d}_convolutional', f'{i:03d}_convolutional', src(i - 2), shape[i - 1][0],
                                    act, dst=dst_of(i, c), bn1=conv_attrs(i - 1)[4], bn2=bn)
                continue
            if i + 1 < n and folded_shortcut(i + 1):
                res = out[_resolve(i + 1, int(layers[i + 1]['from'] if not isinstance(layers[i + 1]['from'], list)
                                              else layers[i + 1]['from'][0]))]
            elif i + 1 < n and folded_upsample(i + 1):
                up = 2
            out[i] = g.conv(f'{i:03d}_convolutional', src(i - 1), c, k, s, act, bn=bn, pad=p, res=res, up=up,
                            f32_out=is_head, dst=dst_of(i, c))
            if is_head:
                heads.append(out[i])
        elif t == 'shortcut':
            if folded_shortcut(i):
                out[i] = out[i - 1]
            else:
                j = _resolve(i, int(L['from'] if not isinstance(L['from'], list) else L['from'][0]))
                out[i] = g.add(out[i - 1], out[j], dst=dst_of(i, c))
        elif t == 'upsample':
            if folded_upsample(i):
                out[i] = out[i - 1]
            else:
                assert int(L.get('stride', 2)) == 2, 'only x2 [upsample]'
                out[i] = g.upsample2(out[i - 1], dst=dst_of(i, c))
        elif t == 'maxpool':
            k, s = int(L['size']), int(L.get('stride', 1))
            if i in spp_members:               # the last pool of the triple emits the fused launch
                if i in spp_groups:
                    sj, ci, o = spp_groups[i]
                    g.spp(out[sj], cat_view(ci).slice(o, 3 * c))
                continue                        # readable only through the concat (checked in pass 2)
            total = (h - 1) * s + k - src(i - 1).h       # SAME_UPPER: extra padding at the end
            total = max(total, 0)
            out[i] = g.pool(src(i - 1), k, s, total // 2, dst=dst_of(i, c), pad_end=total - total // 2)
        elif t == 'route':
            srcs = [_resolve(i, int(r)) for r in L['layers']]
            if len(srcs) == 1:
                v = out[srcs[0]]
                if 'groups' in L:
                    v = v.slice(int(L['group_id']) * c, c)
                out[i] = v
            else:
                tensor = cat_view(i)
                off = 0
                for j in srcs:
                    cj = shape[j][0]
                    pj = producer(j)
                    in_place = (pj is not None and placed.get(pj) == (i, off)) or base(j) in spp_members
                    if not in_place:
                        g.copy(out[j], tensor.slice(off, cj))
                    off += cj
                out[i] = tensor
        elif t == 'yolo':
            out[i] = out[i - 1]
            meta_yolo.append(L)
    # SPP members are only reachable through their concat; the checks above guarantee nothing else reads them
    g.outputs = heads
    meta = _yolo_meta(meta_yolo, heads, (cin0, H, W))
    return g, heads, meta


def _yolo_meta(yolos, heads, input_shape):
    if not yolos:
        return dict(input_shape=input_shape)
    classes = int(yolos[0]['classes'])
    anchors, strides, scales = [], [], []
    for L, hv in zip(yolos, heads):
        a = [int(v) for v in L['anchors']]
        mask = [int(m) for m in L['mask']]
        anchors.append([v for m in mask for v in a[2 * m:2 * m + 2]])
        strides.append(input_shape[1] // hv.h)
        scales.append(float(L.get('scale_x_y', 1.0)))
    return dict(input_shape=input_shape, classes=classes, anchors=anchors, strides=strides, scales=scales,
                new_coords=bool(int(yolos[0].get('new_coords', 0))))


def load_darknet(cfg_path, weights_path, in_hw=None):
    """Convenience: files -> (Graph, heads, meta); raises if the weights file does not match the cfg."""
    w = DarknetWeights(weights_path)
    g, heads, meta = darknet_graph(Path(cfg_path).read_text(), w, in_hw)
    if w.remaining() != 0:
        raise ValueError(f'{w.remaining()} bytes of the weights file were not consumed by the cfg')
    return g, heads, meta
