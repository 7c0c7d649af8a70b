"""ONNX model files -> parameter sources (and, for YOLO, the topology) of the HIP conv engine.

ONNX is the only format the reference ships its models in (README.md:73, scripts/download_models.sh:10-14:
`yolov4_crowdhuman.onnx`, `osnet_x0_25_msmt17.onnx`; MODEL_PATH of fastmot/models/yolo.py:155-156, reid.py:97,106):
TensorRT's OnnxParser reads them there (models/yolo.py:106-151, reid.py:48-92).  Here the protobuf wire format is
read directly -- no `onnx` package (there is none in this image): ModelProto -> GraphProto -> `node` list,
`initializer` tensors, graph inputs / outputs.  Field numbers are those of onnx.proto (IR version >= 3).

Two producers matter:

  * scripts/yolo2onnx.py (the reference's Darknet converter, :403-870).  Every cfg section i (1-based, `000_net` is
    the input) that creates nodes is named `NNN_<type>`: `NNN_convolutional` (+ `_bn`, `_lrelu` / `_softplus`,
    `_tanh`, `_mish` / `_sigmoid`, `_swish` / `_lgx`), `NNN_shortcut` (Add), `NNN_route` (Concat, or Split for
    groups / group_id), `NNN_upsample`, `NNN_maxpool`; initialisers `NNN_convolutional_conv_weights`,
    `..._conv_bias`, `..._bn_{scale,bias,mean,var}` (float_data).  [yolo] sections and single-source [route]
    sections leave no node (`_dummy`, :755-768,865-870).  `OnnxDarknetWeights` serves the initialisers in cfg order
    through the DarknetWeights interface, `darknet_cfg_from_onnx` rebuilds the cfg text from the node list (the
    [yolo] parameters come from the model descriptor, as in the reference, which hard-codes them per class) -- so
    an `.onnx` alone loads, through the same lowering as a `.cfg` + `.weights` pair (models/darknet.py).

  * torch.onnx.export of a torchreid OSNet.  Depending on the exporter version the initialisers keep the module's
    parameter names (`conv1.conv.weight`, `conv1.bn.running_mean`, ...) or BatchNorm is folded into the convs and
    the initialisers are anonymous, and the node order of the four streams and their gates differs between
    versions.  `torchreid_state_dict_from_onnx` relies on neither: it follows the data flow from the input through
    conv1, the OSBlocks (conv1 -> four LightConv3x3 chains of depth 1..4 -> the shared ChannelGate -> conv3 +
    downsample), the transitions, conv5 and the fc head, and names every weight by its place in that structure,
    checking operator types and weight shapes on the way.  The result feeds models/torchreid_weights.TorchreidWeights.
"""
import re
import struct
from pathlib import Path

import numpy as np

# TensorProto.DataType -> numpy
_DTYPES = {1: np.float32, 2: np.uint8, 3: np.int8, 4: np.uint16, 5: np.int16, 6: np.int32, 7: np.int64,
           9: np.bool_, 10: np.float16, 11: np.float64, 12: np.uint32, 13: np.uint64}


# ---------------------------------------------------------------------------------------------- wire format
def _varint(buf, pos):
    result = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7f) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 70:
            raise ValueError('malformed varint')


def _fields(buf):
    """Yields (field number, wire type, value) of one message: int for varint / fixed, memoryview for
    length-delimited payloads (no copy: the initialisers of a detector are hundreds of MB)."""
    pos, end = 0, len(buf)
    while pos < end:
        key, pos = _varint(buf, pos)
        field, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 1:
            val = bytes(buf[pos:pos + 8]); pos += 8
        elif wt == 2:
            n, pos = _varint(buf, pos)
            val = buf[pos:pos + n]; pos += n
            if len(val) != n:
                raise ValueError('truncated ONNX file')
        elif wt == 5:
            val = bytes(buf[pos:pos + 4]); pos += 4
        else:
            raise ValueError(f'unsupported protobuf wire type {wt}')
        yield field, wt, val


def _signed(v):
    return v - (1 << 64) if v >= 1 << 63 else v


def _packed_varints(val):
    out, pos = [], 0
    while pos < len(val):
        v, pos = _varint(val, pos)
        out.append(_signed(v))
    return out


def _ints(wt, val):
    """A repeated int64 field arrives packed (one length-delimited blob) or one varint per element."""
    return _packed_varints(val) if wt == 2 else [_signed(val)]


def _floats(wt, val):
    return np.frombuffer(val, '<f4') if wt == 2 else np.frombuffer(val, '<f4', count=1)


# ---------------------------------------------------------------------------------------------- messages
class Tensor:
    """TensorProto: dims = 1, data_type = 2, float_data = 4, int32_data = 5, int64_data = 7, name = 8,
    raw_data = 9, double_data = 10, uint64_data = 11, external_data = 13, data_location = 14."""

    def __init__(self, buf):
        self.name, self.dims, self.data_type = '', [], 0
        self._raw = None
        self._typed = []                 # (numpy dtype of the wire elements, chunk)
        self._external = False
        for f, wt, v in _fields(buf):
            if f == 1:
                self.dims += _ints(wt, v)
            elif f == 2:
                self.data_type = v
            elif f == 8:
                self.name = bytes(v).decode()
            elif f == 9:
                self._raw = v
            elif f == 4:
                self._typed.append(('f4', _floats(wt, v)))
            elif f == 10:
                self._typed.append(('f8', np.frombuffer(v, '<f8') if wt == 2 else np.frombuffer(v, '<f8', count=1)))
            elif f in (5, 7, 11):
                self._typed.append(('i', np.array(_ints(wt, v), np.int64)))
            elif f == 13 or (f == 14 and v == 1):
                self._external = True

    def array(self):
        if self._external:
            raise NotImplementedError(f'tensor {self.name!r} is stored in an external file')
        if self.data_type not in _DTYPES:
            raise NotImplementedError(f'tensor {self.name!r}: ONNX data type {self.data_type}')
        dt = np.dtype(_DTYPES[self.data_type])
        if self._raw is not None:
            a = np.frombuffer(self._raw, dt.newbyteorder('<'))
        elif self._typed:
            a = np.concatenate([c for _, c in self._typed]) if len(self._typed) > 1 else self._typed[0][1]
            if self.data_type == 10:     # float16 travels as uint16 bit patterns in int32_data
                a = a.astype(np.uint16).view(np.float16)
        else:
            a = np.zeros(0, dt)
        n = int(np.prod(self.dims)) if self.dims else a.size
        if a.size != n:
            raise ValueError(f'tensor {self.name!r}: {a.size} elements for dims {self.dims}')
        return np.asarray(a, dt).reshape(self.dims)


class Node:
    """NodeProto: input = 1, output = 2, name = 3, op_type = 4, attribute = 5 (AttributeProto: name = 1, f = 2,
    i = 3, s = 4, t = 5, floats = 7, ints = 8)."""

    def __init__(self, buf):
        self.inputs, self.outputs, self.name, self.op, self.attrs = [], [], '', '', {}
        for f, wt, v in _fields(buf):
            if f == 1:
                self.inputs.append(bytes(v).decode())
            elif f == 2:
                self.outputs.append(bytes(v).decode())
            elif f == 3:
                self.name = bytes(v).decode()
            elif f == 4:
                self.op = bytes(v).decode()
            elif f == 5:
                k, val = self._attribute(v)
                self.attrs[k] = val

    @staticmethod
    def _attribute(buf):
        name, val, ints, floats = '', None, None, None
        for f, wt, v in _fields(buf):
            if f == 1:
                name = bytes(v).decode()
            elif f == 2:
                val = struct.unpack('<f', v)[0]
            elif f == 3:
                val = _signed(v)
            elif f == 4:
                val = bytes(v)
            elif f == 5:
                val = Tensor(v)
            elif f == 7:
                floats = (floats or []) + _floats(wt, v).tolist()
            elif f == 8:
                ints = (ints or []) + _ints(wt, v)
        if ints is not None:
            val = ints
        elif floats is not None:
            val = floats
        return name, val

    def __repr__(self):
        return f'{self.op}({", ".join(self.inputs)}) -> {", ".join(self.outputs)}'


def _value_info(buf):
    """ValueInfoProto: name = 1, type = 2 (TypeProto.tensor_type = 1: elem_type = 1, shape = 2; dim = 1:
    dim_value = 1, dim_param = 2) -> (name, [dims], None for symbolic ones)."""
    name, shape = '', None
    for f, _, v in _fields(buf):
        if f == 1:
            name = bytes(v).decode()
        elif f == 2:
            for f2, _, v2 in _fields(v):
                if f2 != 1:
                    continue
                for f3, _, v3 in _fields(v2):
                    if f3 != 2:
                        continue
                    shape = []
                    for f4, _, v4 in _fields(v3):
                        if f4 != 1:
                            continue
                        d = None
                        for f5, _, v5 in _fields(v4):
                            if f5 == 1:
                                d = _signed(v5)
                        shape.append(d)
    return name, shape


class OnnxModel:
    """ModelProto (graph = 7, producer_name = 2, opset_import = 8) / GraphProto (node = 1, name = 2,
    initializer = 5, input = 11, output = 12).  `init[name]` decodes an initialiser on first use."""

    def __init__(self, source):
        data = Path(source).read_bytes() if isinstance(source, (str, Path)) else bytes(source)
        self._data = memoryview(data)
        self.producer, self.name, self.opset = '', '', None
        self.nodes, self.inputs, self.outputs = [], [], []
        self._tensors = {}
        graph = None
        for f, wt, v in _fields(self._data):
            if f == 7:
                graph = v
            elif f == 2:
                self.producer = bytes(v).decode()
            elif f == 8:
                for f2, _, v2 in _fields(v):
                    if f2 == 2:
                        self.opset = v2
        if graph is None:
            raise ValueError('not an ONNX model: no graph')
        for f, wt, v in _fields(graph):
            if f == 1:
                self.nodes.append(Node(v))
            elif f == 2:
                self.name = bytes(v).decode()
            elif f == 5:
                t = Tensor(v)
                self._tensors[t.name] = t
            elif f == 11:
                self.inputs.append(_value_info(v))
            elif f == 12:
                self.outputs.append(_value_info(v))
        # graph inputs that are not initialisers = the data inputs (IR < 4 lists the initialisers as inputs too)
        self.data_inputs = [(n, s) for n, s in self.inputs if n not in self._tensors]
        # Constant nodes carry tensors as well (torch exports small constants this way)
        for nd in self.nodes:
            if nd.op == 'Constant' and isinstance(nd.attrs.get('value'), Tensor):
                t = nd.attrs['value']
                t.name = nd.outputs[0]
                self._tensors.setdefault(t.name, t)
        self._cache = {}

    def has(self, name):
        return name in self._tensors

    def tensor(self, name):
        if name not in self._cache:
            if name not in self._tensors:
                raise KeyError(f'ONNX model has no initialiser {name!r}')
            self._cache[name] = self._tensors[name].array()
        return self._cache[name]

    def initializer_names(self):
        return list(self._tensors)


# ---------------------------------------------------------------------------------------------- yolo2onnx models
_CONV_INIT = re.compile(r'^(\d+)_convolutional_conv_weights$')


class OnnxDarknetWeights:
    """The initialisers of a scripts/yolo2onnx.py model through the DarknetWeights interface (models/darknet.py):
    `conv()` is called once per [convolutional] section in cfg order and answers with the next
    `NNN_convolutional_*` group (yolo2onnx.py:316-346: bn_bias / bn_scale / bn_mean / bn_var + conv_weights, or
    conv_bias + conv_weights), shapes checked."""

    def __init__(self, source):
        self.model = source if isinstance(source, OnnxModel) else OnnxModel(source)
        self.prefixes = sorted((int(m.group(1)), f'{m.group(1)}_convolutional')
                               for m in map(_CONV_INIT.match, self.model.initializer_names()) if m)
        if not self.prefixes:
            raise ValueError('no NNN_convolutional_conv_weights initialisers: not a scripts/yolo2onnx.py model')
        self.pos = 0

    def conv(self, name, cout, cin, k, bn=True, gain=1.0, groups=1):
        if self.pos >= len(self.prefixes):
            raise ValueError('the ONNX model has fewer convolutional layers than the topology')
        _, pre = self.prefixes[self.pos]
        self.pos += 1
        m = self.model
        w = m.tensor(pre + '_conv_weights').astype(np.float32)
        if tuple(w.shape) != (cout, cin // groups, k, k):
            raise ValueError(f'{pre}: ONNX weights {tuple(w.shape)} != layer {(cout, cin // groups, k, k)}')
        has_bn = m.has(pre + '_bn_scale')
        if has_bn != bool(bn):
            raise ValueError(f'{pre}: batch_normalize differs between the ONNX model and the topology')
        if bn:
            p = dict(gamma=m.tensor(pre + '_bn_scale'), beta=m.tensor(pre + '_bn_bias'),
                     mean=m.tensor(pre + '_bn_mean'), var=m.tensor(pre + '_bn_var'))
            p = {k_: v.astype(np.float32).reshape(cout) for k_, v in p.items()}
        else:
            p = dict(bias=m.tensor(pre + '_conv_bias').astype(np.float32).reshape(cout))
        p['w'] = w
        return p

    def remaining(self):
        """Convolutional layers of the file the topology did not consume (0 when they match)."""
        return len(self.prefixes) - self.pos


_ACT_SUFFIX = (('_lrelu', 'leaky'), ('_mish', 'mish'), ('_swish', 'swish'), ('_lgx', 'logistic'))


def darknet_cfg_from_onnx(model, descriptor):
    """Darknet cfg text equivalent to a scripts/yolo2onnx.py model (the inverse of GraphBuilderONNX, yolo2onnx.py:
    486-870).  `descriptor`: the YOLO model class -- its NUM_CLASSES / ANCHORS / SCALES / NEW_COORDS fill the [yolo]
    sections, which leave no trace in the file (the reference hard-codes them per class as well, models/yolo.py)."""
    model = model if isinstance(model, OnnxModel) else OnnxModel(model)
    if not model.data_inputs or model.data_inputs[0][1] is None or len(model.data_inputs[0][1]) != 4:
        raise ValueError('ONNX model has no NCHW data input')
    in_name, (_, cin0, in_h, in_w) = model.data_inputs[0]
    if not all(isinstance(v, int) and v > 0 for v in (cin0, in_h, in_w)):
        raise ValueError(f'ONNX input {in_name!r} has symbolic or unknown dimensions {model.data_inputs[0][1]}: '
                         'scripts/yolo2onnx.py writes concrete ones, and the layer table needs them')

    def idx_of(tensor):                       # 1-based section index of a tensor ('017_convolutional_lrelu' -> 17)
        if tensor == in_name:
            return 0
        if not re.match(r'^\d{3}_', tensor):
            raise ValueError(f'tensor {tensor!r} does not follow the yolo2onnx naming (three-digit section index first)')
        return int(tensor[:3])
    by_idx = {}
    for nd in model.nodes:
        m = re.match(r'^(\d{3})_(convolutional|shortcut|route|upsample|maxpool)(.*)$', nd.name)
        if not m:
            raise ValueError(f'node {nd.name!r} ({nd.op}) does not follow the yolo2onnx naming')
        by_idx.setdefault(int(m.group(1)), []).append(nd)
    outputs = {idx_of(n) for n, _ in model.outputs}
    last = max(max(by_idx), max(outputs) + 1)
    sections = [f'[net]\nbatch=1\nchannels={cin0}\nheight={in_h}\nwidth={in_w}\n']
    yolo_k = 0

    def yolo_section():
        nonlocal yolo_k
        k = yolo_k
        yolo_k += 1
        if k >= len(descriptor.ANCHORS):
            raise ValueError(f'the ONNX model has more outputs than {descriptor.__name__} has heads')
        a = descriptor.ANCHORS[k]
        n = len(a) // 2
        scales = descriptor.SCALES
        s = scales[k] if k < len(scales) else 1.0           # (YOLOv3 lists two scales for three heads, models/yolo.py:272)
        return (f'[yolo]\nmask={",".join(str(i) for i in range(n))}\nanchors={",".join(str(v) for v in a)}\n'
                f'classes={descriptor.NUM_CLASSES}\nnum={n}\nscale_x_y={s}\n' +
                ('new_coords=1\n' if descriptor.NEW_COORDS else ''))

    i = 1
    while i <= last:
        if i in by_idx:
            nodes = by_idx[i]
            head = nodes[0]
            if head.op == 'Conv':
                w = model.tensor(head.inputs[1])
                act = 'linear'
                for nd in nodes:
                    for suf, a in _ACT_SUFFIX:
                        if nd.name.endswith(suf):
                            act = a
                bn = any(nd.op == 'BatchNormalization' for nd in nodes)
                stride = head.attrs.get('strides', [1, 1])[0]
                sections.append(f'[convolutional]\n{"batch_normalize=1" + chr(10) if bn else ""}filters={w.shape[0]}\n'
                                f'size={w.shape[2]}\nstride={stride}\npad=1\nactivation={act}\n')
            elif head.op == 'Add':
                if idx_of(head.inputs[0]) != i - 1:
                    raise NotImplementedError(f'{head.name}: [shortcut] behind a [route]')
                sections.append(f'[shortcut]\nfrom={idx_of(head.inputs[1]) - 1}\nactivation=linear\n')
            elif head.op == 'Concat':
                sections.append('[route]\nlayers=' + ','.join(str(idx_of(t) - 1) for t in head.inputs) + '\n')
            elif head.op == 'Split':
                groups = len(head.outputs)
                gid = [k for k, o in enumerate(head.outputs) if 'dummy' not in o][0]
                sections.append(f'[route]\nlayers={idx_of(head.inputs[0]) - 1}\ngroups={groups}\ngroup_id={gid}\n')
            elif head.op in ('Upsample', 'Resize'):
                scales = model.tensor(head.inputs[-1])
                sections.append(f'[upsample]\nstride={int(round(float(scales[-1])))}\n')
            elif head.op == 'MaxPool':
                sections.append(f'[maxpool]\nsize={head.attrs["kernel_shape"][0]}\nstride={head.attrs["strides"][0]}\n')
            else:
                raise NotImplementedError(f'{head.name}: {head.op}')
            i += 1
            continue
        # a run of sections that left no node: [yolo] behind an output conv, then (unless the net ends) one
        # single-source [route] whose source is what the next section consumes
        run = [i]
        while run[-1] + 1 <= last and run[-1] + 1 not in by_idx:
            run.append(run[-1] + 1)
        nxt = run[-1] + 1
        consumer = by_idx.get(nxt, [None])[0]
        for j in run:
            if j - 1 in outputs and (j == run[0]):
                sections.append(yolo_section())
            elif consumer is not None and j == run[-1]:
                if consumer.op == 'Concat':
                    raise NotImplementedError('single-source [route] directly before a multi-source [route]')
                sections.append(f'[route]\nlayers={idx_of(consumer.inputs[0]) - 1}\n')
            else:
                raise ValueError(f'section {j} of the ONNX model cannot be reconstructed')
        i = nxt
    if yolo_k != len(descriptor.ANCHORS):
        raise ValueError(f'{yolo_k} [yolo] outputs found, {descriptor.__name__} declares {len(descriptor.ANCHORS)}')
    return '\n'.join(sections)


# ---------------------------------------------------------------------------------------------- torchreid OSNet
class _OSNetWalk:
    """Follows the DATA FLOW of an exported torchreid OSNet (torchreid/models/osnet.py) from the graph input and names
    every weight by its place in the structure -- not by its position in the node list (exporters and torchreid
    versions order the four streams and their gates differently) and not by its initialiser name (BatchNorm folding
    renames them).  Every step checks operator type and weight shape; anything unexpected raises ValueError."""

    def __init__(self, model, channels, feature_dim):
        self.m, self.channels, self.dim = model, channels, feature_dim
        self.cons = {}
        for nd in model.nodes:
            for t in nd.inputs:
                self.cons.setdefault(t, []).append(nd)
        self.sd = {}

    def fail(self, where, what):
        raise ValueError(f'{where}: {what} (the graph does not walk like torchreid OSNet, expected channels '
                         f'{self.channels})')

    def users(self, t, op=None):
        return [nd for nd in self.cons.get(t, []) if op is None or nd.op == op]

    def wshape(self, nd):
        return tuple(self.m.tensor(nd.inputs[1]).shape) if len(nd.inputs) > 1 and self.m.has(nd.inputs[1]) else None

    def conv(self, t, shape, name, bn=None, node=None):
        """The Conv reading tensor t with weights of `shape` -> its output tensor (behind its BatchNormalization, when
        the exporter left one); records `<name>.weight` (+ `.bias`) and the `<bn>.*` vectors."""
        nd = node
        if nd is None:
            cands = [n for n in self.users(t, 'Conv') if self.wshape(n) == tuple(shape)]
            if not cands:
                self.fail(name, f'no Conv with weights {tuple(shape)} reads {t!r} '
                                f'(found {[self.wshape(n) for n in self.users(t, "Conv")]})')
            nd = cands[0]
        self.sd[name + '.weight'] = self.m.tensor(nd.inputs[1]).astype(np.float32)
        if len(nd.inputs) > 2 and self.m.has(nd.inputs[2]):
            self.sd[name + '.bias'] = self.m.tensor(nd.inputs[2]).astype(np.float32)
        return self.batchnorm(nd.outputs[0], name, bn)

    def batchnorm(self, out, name, bn):
        bns = self.users(out, 'BatchNormalization')
        if bns:
            if bn is None:
                self.fail(name, 'unexpected BatchNormalization behind it')
            g, b, mean, var = (self.m.tensor(t).astype(np.float32) for t in bns[0].inputs[1:5])
            eps = bns[0].attrs.get('epsilon', 1e-5)
            if abs(eps - 1e-5) > 1e-9:           # fold_bn uses eps = 1e-5: var' + 1e-5 = var + eps
                var = var + np.float32(eps - 1e-5)
            self.sd.update({bn + '.weight': g, bn + '.bias': b, bn + '.running_mean': mean, bn + '.running_var': var})
            return bns[0].outputs[0]
        if bn is not None and name + '.bias' not in self.sd:
            self.fail(name, 'neither a BatchNormalization node behind it nor a folded bias')
        return out

    def through(self, t, op, where):
        nds = self.users(t, op)
        if not nds:
            self.fail(where, f'no {op} reads {t!r} (readers: {[n.op for n in self.users(t)]})')
        return nds[0].outputs[0]

    def light_chain(self, first_pw, mid, where):
        """One stream from its first pointwise conv: [(pw node, dw node)] and the tensor the gate reads."""
        chain, nd = [], first_pw
        while True:
            dws = [n for n in self.users(nd.outputs[0], 'Conv') if self.wshape(n) == (mid, 1, 3, 3)]
            if not dws:
                self.fail(where, 'LightConv3x3: no depthwise 3x3 behind the pointwise conv')
            chain.append((nd, dws[0]))
            out = dws[0].outputs[0]
            bns = self.users(out, 'BatchNormalization')
            if bns:
                out = bns[0].outputs[0]
            out = self.through(out, 'Relu', where)
            nxt = [n for n in self.users(out, 'Conv') if self.wshape(n) == (mid, mid, 1, 1)]
            if not nxt or len(chain) == 4:
                return chain, out
            nd = nxt[0]

    def block(self, x, cin, cout, prefix):
        mid = cout // 4
        hid = max(mid // 16, 1)
        x1 = self.through(self.conv(x, (mid, cin, 1, 1), prefix + '.conv1.conv', prefix + '.conv1.bn'), 'Relu', prefix)
        firsts = [n for n in self.users(x1, 'Conv') if self.wshape(n) == (mid, mid, 1, 1)]
        chains = sorted((self.light_chain(n, mid, prefix) for n in firsts), key=lambda c: len(c[0]))
        if [len(c[0]) for c in chains] != [1, 2, 3, 4]:
            self.fail(prefix, f'streams of depth {[len(c[0]) for c in chains]} instead of 1, 2, 3, 4')
        gated = []
        for (chain, out), sname in zip(chains, ('conv2a', 'conv2b', 'conv2c', 'conv2d')):
            for i, (pw, dw) in enumerate(chain):
                lc = f'{prefix}.{sname}' if sname == 'conv2a' else f'{prefix}.{sname}.{i}'
                self.conv(None, None, lc + '.conv1', node=pw)
                self.conv(None, None, lc + '.conv2', lc + '.bn', node=dw)
            # ChannelGate: global average pool -> fc1 -> ReLU -> fc2 -> sigmoid -> multiply (one gate, four uses)
            g = self.through(out, 'GlobalAveragePool', prefix + '.gate')
            g = self.through(self.conv(g, (hid, mid, 1, 1), prefix + '.gate.fc1'), 'Relu', prefix + '.gate')
            g = self.through(self.conv(g, (mid, hid, 1, 1), prefix + '.gate.fc2'), 'Sigmoid', prefix + '.gate')
            muls = [n for n in self.users(g, 'Mul') if out in n.inputs]
            if not muls:
                self.fail(prefix + '.gate', 'the gate does not multiply its stream')
            gated.append(muls[0].outputs[0])
        t = gated[0]                               # the four gated streams meet in a chain of Adds before conv3
        for _ in range(4):
            if [n for n in self.users(t, 'Conv') if self.wshape(n) == (cout, mid, 1, 1)]:
                break
            t = self.through(t, 'Add', prefix + ' (sum of the gated streams)')
        x3 = self.conv(t, (cout, mid, 1, 1), prefix + '.conv3.conv', prefix + '.conv3.bn')
        if cin != cout:
            self.conv(x, (cout, cin, 1, 1), prefix + '.downsample.conv', prefix + '.downsample.bn')
        return self.through(self.through(x3, 'Add', prefix), 'Relu', prefix)

    def run(self):
        c0, c1, c2, c3 = self.channels
        if not self.m.data_inputs:
            self.fail('input', 'no data input')
        x = self.m.data_inputs[0][0]
        x = self.through(self.conv(x, (c0, 3, 7, 7), 'conv1.conv', 'conv1.bn'), 'Relu', 'conv1')
        x = self.through(x, 'MaxPool', 'maxpool')
        for stage, cin, cout, transition in (('conv2', c0, c1, True), ('conv3', c1, c2, True), ('conv4', c2, c3, False)):
            x = self.block(x, cin, cout, stage + '.0')
            x = self.block(x, cout, cout, stage + '.1')
            if transition:
                x = self.through(self.conv(x, (cout, cout, 1, 1), f'{stage}.2.0.conv', f'{stage}.2.0.bn'), 'Relu', stage)
                x = self.through(x, 'AveragePool', stage)
        x = self.through(self.conv(x, (c3, c3, 1, 1), 'conv5.conv', 'conv5.bn'), 'Relu', 'conv5')
        x = self.through(x, 'GlobalAveragePool', 'global pool')
        for _ in range(6):                          # Flatten / Reshape / Squeeze (+ the shape arithmetic feeding them)
            fc = self.users(x, 'Gemm') + self.users(x, 'MatMul')
            if fc:
                break
            nxt = [n for n in self.users(x) if n.op in ('Flatten', 'Reshape', 'Squeeze')]
            if not nxt:
                self.fail('fc', f'no Gemm behind the global pool (readers of {x!r}: {[n.op for n in self.users(x)]})')
            x = nxt[0].outputs[0]
        else:
            self.fail('fc', 'no Gemm behind the global pool')
        nd = fc[0]
        w = self.m.tensor(nd.inputs[1]).astype(np.float32)
        if not (nd.op == 'Gemm' and nd.attrs.get('transB', 0)):
            w = w.T                                                # stored [in, out]
        if w.shape != (self.dim, c3):
            self.fail('fc', f'weights {w.shape}, expected {(self.dim, c3)}')
        self.sd['fc.0.weight'] = np.ascontiguousarray(w)
        out = nd.outputs[0]
        if len(nd.inputs) > 2 and self.m.has(nd.inputs[2]):
            self.sd['fc.0.bias'] = self.m.tensor(nd.inputs[2]).astype(np.float32).reshape(-1)
        else:                                                       # MatMul + Add(bias)
            for add in self.users(out, 'Add'):
                other = [t for t in add.inputs if t != out]
                if other and self.m.has(other[0]):
                    self.sd['fc.0.bias'] = self.m.tensor(other[0]).astype(np.float32).reshape(-1)
                    out = add.outputs[0]
        self.batchnorm(out, 'fc.0', 'fc.1')
        return self.sd


def torchreid_state_dict_from_onnx(model, channels, feature_dim=512):
    """-> dict of torchreid parameter names (`conv1.conv.weight`, `conv1.bn.running_mean`, `fc.0.bias`, ...) from an
    ONNX export of torchreid's OSNet.  A conv whose BatchNorm the exporter folded away carries `<conv>.bias` and no
    `<bn>.*` entries (TorchreidWeights treats that as an already folded layer)."""
    model = model if isinstance(model, OnnxModel) else OnnxModel(model)
    return _OSNetWalk(model, tuple(channels), feature_dim).run()
