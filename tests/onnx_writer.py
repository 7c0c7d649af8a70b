"""TEST INFRASTRUCTURE: a minimal ONNX protobuf ENCODER with the surface of `onnx.helper` that scripts/yolo2onnx.py
uses (`helper.make_node / make_tensor / make_tensor_value_info / make_graph / make_model / printable_graph`,
`TensorProto.FLOAT`, `checker.check_model`, `save`).  The `onnx` package is not in this image; installed as the module
`onnx` (install_stub()), this lets the reference's converter run UNMODIFIED (oracle/make_golden_onnx.py) and lets the
tests write model files for fastmot_amd/models/onnx_reader.py to read back.  Field numbers: onnx.proto.  (The reader
is held against a second, independent writer as well: torch's own C++ ONNX serializer, tests/test_onnx_reader.py.)"""
import struct
import sys
import types

import numpy as np


def _varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7f
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _key(field, wt):
    return _varint(field << 3 | wt)


def _ld(field, payload):
    return _key(field, 2) + _varint(len(payload)) + bytes(payload)


def _str(field, s):
    return _ld(field, s.encode())


def _int(field, v):
    return _key(field, 0) + _varint(int(v))


class TensorProto:
    FLOAT, INT64 = 1, 7


class _Msg:
    def __init__(self, data, **info):
        self.data = data
        self.__dict__.update(info)

    def SerializeToString(self):
        return self.data


def make_tensor(name, data_type, dims, vals, raw=False):
    body = b''.join(_int(1, d) for d in dims) + _int(2, data_type)
    a = np.asarray(vals, np.float32 if data_type == TensorProto.FLOAT else np.int64).reshape(-1)
    if raw:
        body += _ld(9, a.tobytes())
    elif data_type == TensorProto.FLOAT:
        body += _ld(4, a.astype('<f4').tobytes())              # float_data, packed
    else:
        body += _ld(7, b''.join(_varint(int(v)) for v in a))   # int64_data, packed
    body += _str(8, name)
    return _Msg(body, name=name)


def make_tensor_value_info(name, elem_type, shape):
    dims = b''.join(_ld(1, _int(1, d) if isinstance(d, int) else _str(2, str(d))) for d in shape)
    ttype = _int(1, elem_type) + _ld(2, dims)
    return _Msg(_str(1, name) + _ld(2, _ld(1, ttype)), name=name)


def _attribute(name, v):
    body = _str(1, name)
    if isinstance(v, float):
        body += _key(2, 5) + struct.pack('<f', v) + _int(20, 1)
    elif isinstance(v, (int, np.integer)):
        body += _int(3, v) + _int(20, 2)
    elif isinstance(v, (str, bytes)):
        body += _ld(4, v.encode() if isinstance(v, str) else v) + _int(20, 3)
    elif isinstance(v, (list, tuple)) and all(isinstance(x, (int, np.integer)) for x in v):
        body += b''.join(_int(8, x) for x in v) + _int(20, 7)           # ints, one varint per element
    elif isinstance(v, (list, tuple)):
        body += _ld(7, np.asarray(v, '<f4').tobytes()) + _int(20, 6)
    else:
        raise TypeError(f'attribute {name}: {type(v)}')
    return body


def make_node(op_type, inputs, outputs, name=None, **attrs):
    body = b''.join(_str(1, i) for i in inputs) + b''.join(_str(2, o) for o in outputs)
    if name:
        body += _str(3, name)
    body += _str(4, op_type) + b''.join(_ld(5, _attribute(k, v)) for k, v in attrs.items())
    return _Msg(body, name=name, op_type=op_type)


def make_graph(nodes, name, inputs, outputs, initializer=()):
    body = b''.join(_ld(1, n.data) for n in nodes) + _str(2, name) + b''.join(_ld(5, t.data) for t in initializer)
    body += b''.join(_ld(11, v.data) for v in inputs) + b''.join(_ld(12, v.data) for v in outputs)
    return _Msg(body, name=name)


def make_model(graph, producer_name='tests/onnx_writer.py', opset=11):
    body = _int(1, 6) + _str(2, producer_name) + _ld(7, graph.data) + _ld(8, _str(1, '') + _int(2, opset))
    return _Msg(body)


def save(model, path):
    with open(path, 'wb') as f:
        f.write(model.data)


def install_stub():
    """Registers this encoder as the importable module `onnx` (`import onnx`, `from onnx import helper, TensorProto`)."""
    me = sys.modules[__name__]
    onnx = types.ModuleType('onnx')
    helper = types.ModuleType('onnx.helper')
    for fn in ('make_tensor', 'make_tensor_value_info', 'make_node', 'make_graph', 'make_model'):
        setattr(helper, fn, getattr(me, fn))
    helper.printable_graph = lambda g: f'<graph {g.name}>'
    checker = types.ModuleType('onnx.checker')
    checker.check_model = lambda m: None
    onnx.helper, onnx.checker, onnx.TensorProto, onnx.save = helper, checker, TensorProto, save
    sys.modules.update({'onnx': onnx, 'onnx.helper': helper, 'onnx.checker': checker})
    return onnx
