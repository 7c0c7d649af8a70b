"""ONNX model reader (fastmot_amd/models/onnx_reader.py): ONNX is the only format the
reference ships its models in (README.md:73, scripts/download_models.sh:10-14).  CPU tests.

  * the files the reference's own converter writes (scripts/yolo2onnx.py run unmodified, oracle/make_golden_onnx.py ->
    tests/golden/yolo2onnx_*.onnx; re-generated live where /root/reference exists) read back as the network their
    cfg + weights describe: same tensors in the same order, and the topology rebuilt from the node list lowers to
    the same layer table as the cfg;
  * a full-size yolov4.cfg (the topology of the reference's yolov4_crowdhuman.onnx: 512x512, 2 classes) written with
    the reference converter's naming -> the YOLOv4 descriptor loads it through MODEL_PATH alone;
  * OSNet exported by torch's own ONNX serializer (an independent writer: libtorch's protobuf code), with BatchNorm
    folded (eval-mode export) and unfolded, named and anonymous initialisers -> the layer table equals the one built
    from the module's state_dict."""
import io
import sys
from pathlib import Path

import numpy as np
import pytest

import darknet_cases as dc
import onnx_writer as ow
from fastmot_amd.models import YOLO, ReID, darknet, onnx_reader
from fastmot_amd.models.torchreid_weights import TorchreidWeights

GOLDEN = Path(__file__).parent / 'golden'
REF_SCRIPT = Path('/root/reference/scripts/yolo2onnx.py')
CASES = {'mini_v4': dc.MINI_V4, 'mini_tiny': dc.MINI_TINY}


class MiniV4(YOLO):
    NUM_CLASSES = 2
    INPUT_SHAPE = (3, 64, 96)
    LAYER_FACTORS = [2, 4]
    SCALES = [1.2, 1.1]
    ANCHORS = [[12, 16, 19, 36, 40, 28], [36, 75, 76, 55, 72, 146]]


def descriptor_for(meta):
    """A YOLO descriptor carrying what the [yolo] sections of the cfg say (the ONNX file does not)."""
    _, h, w = meta['input_shape']
    return type('Desc', (), dict(NUM_CLASSES=meta['classes'], ANCHORS=meta['anchors'], SCALES=meta['scales'],
                                 NEW_COORDS=meta['new_coords'], INPUT_SHAPE=(3, h, w), __name__='Desc'))


def layer_signature(g):
    return [(d['op'], d.get('name'), d['cin'], d['cout'], d['k'], d['stride'], d['act'], d['up'], d['out'].tid, d['out'].coff,
             tuple((v.tid, v.coff) for v in d['ins'])) for d in g.layers]


@pytest.mark.parametrize('name', sorted(CASES))
def test_reference_converter_output_reads_back(name, tmp_path):
    cfg = darknet.parse_cfg(CASES[name])
    blob = dc.random_weights_file(cfg, seed=11)
    g0, heads0, meta = darknet.darknet_graph(cfg, darknet.DarknetWeights(blob))
    files = [GOLDEN / f'yolo2onnx_{name}.onnx']
    if REF_SCRIPT.is_file():                       # build container: run the reference converter again, live
        sys.path.insert(0, str(Path(__file__).parents[1] / 'oracle'))
        import make_golden_onnx
        (tmp_path / f'yolo2onnx_{name}.cfg').write_text(CASES[name])
        (tmp_path / f'yolo2onnx_{name}.weights').write_bytes(blob)
        live = make_golden_onnx.convert(tmp_path / f'yolo2onnx_{name}.cfg', tmp_path / f'yolo2onnx_{name}.weights', tmp_path)
        assert live.read_bytes() == files[0].read_bytes(), 'the committed golden is stale'
        files.append(live)
    for f in files:
        model = onnx_reader.OnnxModel(f)
        assert model.producer == 'NVIDIA TensorRT sample' and model.data_inputs[0][0] == '000_net'
        # (1) the initialisers in cfg order = the Darknet file's tensors
        a, b = onnx_reader.OnnxDarknetWeights(model), darknet.DarknetWeights(blob)
        n = 0
        for cout, cin, k, bn in dc.conv_sections(cfg):
            pa, pb = a.conv(f'c{n}', cout, cin, k, bn=bn), b.conv(f'c{n}', cout, cin, k, bn=bn)
            assert pa.keys() == pb.keys()
            for k in pa:
                np.testing.assert_array_equal(pa[k], pb[k])
            n += 1
        assert a.remaining() == 0 and b.remaining() == 0
        # (2) the topology rebuilt from the node list lowers to the same layer table
        text = onnx_reader.darknet_cfg_from_onnx(model, descriptor_for(meta))
        g1, heads1, meta1 = darknet.darknet_graph(text, onnx_reader.OnnxDarknetWeights(model))
        assert meta1 == meta
        assert [d['type'] for d in darknet.parse_cfg(text)] == [d['type'] for d in cfg]
        assert layer_signature(g1) == layer_signature(g0)
        assert bytes(g1.blob) == bytes(g0.blob)                  # every packed weight / bias of the engine
        assert [(h.tid, h.coff, h.c) for h in heads1] == [(h.tid, h.coff, h.c) for h in heads0]


def write_yolo2onnx_style(cfg_layers, blob, path, batch=1):
    """An ONNX file with the naming and layout of scripts/yolo2onnx.py (:228-262,316-400,558-870) from a parsed cfg +
    Darknet weights: used for the full-size case, where running the reference converter itself is possible only in
    the build container (test above) and a committed golden would be 250 MB."""
    w = darknet.DarknetWeights(blob)
    net = cfg_layers[0]
    nodes, inits, out_names = [], [], []
    inputs = [ow.make_tensor_value_info('000_net', ow.TensorProto.FLOAT, [batch, 3, int(net['height']), int(net['width'])])]
    names, chans = ['000_net'], [3]          # output tensor name / channels per section (index 0 = net)
    route = 0
    for i, L in enumerate(cfg_layers[1:], 1):
        t, base = L['type'], f'{i:03d}_{L["type"]}'
        prev = names[route] if route else names[-1]
        pc = chans[route] if route else chans[-1]
        if t == 'convolutional':
            route = 0
            k, f, bn = int(L.get('size', 1)), int(L['filters']), bool(L.get('batch_normalize', 0))
            p = w.conv(base, f, pc, k, bn=bn)
            ins = [prev, base + '_conv_weights']
            if bn:
                for suf, key in (('scale', 'gamma'), ('bias', 'beta'), ('mean', 'mean'), ('var', 'var')):
                    inits.append(ow.make_tensor(f'{base}_bn_{suf}', ow.TensorProto.FLOAT, [f], p[key]))
            else:
                inits.append(ow.make_tensor(base + '_conv_bias', ow.TensorProto.FLOAT, [f], p['bias']))
                ins.append(base + '_conv_bias')
            inits.append(ow.make_tensor(base + '_conv_weights', ow.TensorProto.FLOAT, list(p['w'].shape), p['w']))
            nodes.append(ow.make_node('Conv', ins, [base], name=base, kernel_shape=[k, k], strides=[int(L.get('stride', 1))] * 2,
                                      auto_pad='SAME_LOWER', dilations=[1, 1]))
            out = base
            if bn:
                nodes.append(ow.make_node('BatchNormalization', [out] + [f'{base}_bn_{s}' for s in ('scale', 'bias', 'mean', 'var')],
                                          [base + '_bn'], name=base + '_bn', epsilon=1e-5, momentum=0.99))
                out = base + '_bn'
            act = L.get('activation', 'linear')
            if act == 'leaky':
                nodes.append(ow.make_node('LeakyRelu', [out], [base + '_lrelu'], name=base + '_lrelu', alpha=0.1))
                out = base + '_lrelu'
            elif act == 'mish':
                nodes.append(ow.make_node('Softplus', [out], [base + '_softplus'], name=base + '_softplus'))
                nodes.append(ow.make_node('Tanh', [base + '_softplus'], [base + '_tanh'], name=base + '_tanh'))
                nodes.append(ow.make_node('Mul', [out, base + '_tanh'], [base + '_mish'], name=base + '_mish'))
                out = base + '_mish'
            elif act == 'logistic':
                nodes.append(ow.make_node('Sigmoid', [out], [base + '_lgx'], name=base + '_lgx'))
                out = base + '_lgx'
            names.append(out); chans.append(f)
        elif t == 'shortcut':
            route = 0
            src = int(L['from'])
            nodes.append(ow.make_node('Add', [prev, names[src if src < 0 else src + 1]], [base], name=base))
            names.append(base); chans.append(pc)
        elif t == 'route':
            idx = [int(v) for v in L['layers']]
            if len(idx) == 1:
                route = idx[0] - 1 if idx[0] < 0 else idx[0] + 1
                if route < 0:
                    route += len(names) + 1                     # (relative to the dummy entry appended below)
                names.append(base + '_dummy'); chans.append(1)
            else:
                srcs = [(j if j < 0 else j + 1) for j in idx]
                nodes.append(ow.make_node('Concat', [names[j] for j in srcs], [base], name=base, axis=1))
                names.append(base); chans.append(sum(chans[j] for j in srcs))
        elif t == 'upsample':
            route = 0
            inits.append(ow.make_tensor(base + '_scale', ow.TensorProto.FLOAT, [4], [1., 1., float(L['stride']), float(L['stride'])]))
            nodes.append(ow.make_node('Upsample', [prev, base + '_scale'], [base], name=base, mode='nearest'))
            names.append(base); chans.append(pc)
        elif t == 'maxpool':
            route = 0
            nodes.append(ow.make_node('MaxPool', [prev], [base], name=base, kernel_shape=[int(L['size'])] * 2,
                                      strides=[int(L['stride'])] * 2, auto_pad='SAME_UPPER'))
            names.append(base); chans.append(pc)
        elif t == 'yolo':
            out_names.append(names[-1])
            names.append(base + '_dummy'); chans.append(1)
    assert w.remaining() == 0
    outputs = [ow.make_tensor_value_info(n, ow.TensorProto.FLOAT, [batch, 1, 1, 1]) for n in out_names]
    graph = ow.make_graph(nodes, Path(path).stem, inputs + [ow.make_tensor_value_info(t.name, ow.TensorProto.FLOAT, [1]) for t in inits],
                          outputs, inits)
    ow.save(ow.make_model(graph, producer_name='NVIDIA TensorRT sample'), path)


def test_writer_above_equals_the_reference_converter():
    """(keeps write_yolo2onnx_style honest: on the mini cfg it must produce the file the reference converter wrote,
    up to the placeholder shapes of the graph inputs / outputs)"""
    cfg = darknet.parse_cfg(dc.MINI_V4)
    blob = dc.random_weights_file(cfg, seed=11)
    buf = GOLDEN.parent / '_tmp_mini.onnx'
    try:
        write_yolo2onnx_style(cfg, blob, buf)
        a, b = onnx_reader.OnnxModel(buf), onnx_reader.OnnxModel(GOLDEN / 'yolo2onnx_mini_v4.onnx')
    finally:
        buf.unlink(missing_ok=True)
    assert [(n.op, n.name, n.inputs, n.outputs, n.attrs) for n in a.nodes] == \
           [(n.op, n.name, n.inputs, n.outputs, n.attrs) for n in b.nodes]
    assert a.initializer_names() == b.initializer_names()
    for k in a.initializer_names():
        np.testing.assert_array_equal(a.tensor(k), b.tensor(k))
    assert [n for n, _ in a.outputs] == [n for n, _ in b.outputs]


def test_yolov4_descriptor_loads_its_onnx_model_path(tmp_path, monkeypatch):
    """The reference's own model file name and format: `yolov4_crowdhuman.onnx` (512x512, 2 classes, yolov4.cfg
    topology) at YOLOv4.MODEL_PATH, nothing else next to it."""
    model = YOLO.get_model('YOLOv4')
    assert model.MODEL_PATH.name == 'yolov4_crowdhuman.onnx'            # fastmot/models/yolo.py:156
    cfg = darknet.parse_cfg(dc.yolov4_cfg(512, 512, 2))
    blob = dc.random_weights_file(cfg, seed=3)
    path = tmp_path / 'yolov4_crowdhuman.onnx'
    write_yolo2onnx_style(cfg, blob, path)
    monkeypatch.setattr(model, 'MODEL_PATH', path)
    g, heads = model.build_graph()                                       # no weights=, no cfg file, no opt-in
    g0, heads0, _ = darknet.darknet_graph(cfg, darknet.DarknetWeights(blob))
    assert layer_signature(g) == layer_signature(g0) and bytes(g.blob) == bytes(g0.blob)
    assert len(heads) == 3 and [h.c for h in heads] == [21, 21, 21]
    # a file of another input size is refused like the reference's `assert INPUT_SHAPE == net_input.shape[1:]`
    write_yolo2onnx_style(darknet.parse_cfg(dc.yolov4_cfg(416, 416, 2)), blob, path)
    with pytest.raises(ValueError, match='INPUT_SHAPE'):
        model.build_graph()
    # the Darknet pair of the same stem is the fallback
    path.unlink()
    path.with_suffix('.weights').write_bytes(blob)
    g2, _ = model.build_graph()
    assert bytes(g2.conv_params[5][1].tobytes()) == bytes(g.conv_params[5][1].tobytes())


# ---------------------------------------------------------------------------------------------- OSNet
def export_with_torch(module, x, **kw):
    """torch.onnx.export (TorchScript exporter: the protobuf is serialized by libtorch) without the `onnx` package: only
    its last step -- looking for onnxscript functions in the finished proto -- imports it, and is skipped here."""
    import torch
    from torch.onnx._internal.torchscript_exporter import onnx_proto_utils
    orig = onnx_proto_utils._add_onnxscript_fn
    onnx_proto_utils._add_onnxscript_fn = lambda proto, custom_opsets: proto
    buf = io.BytesIO()
    try:
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            torch.onnx.export(module, x, buf, dynamo=False, opset_version=11, input_names=['input'],
                              output_names=['output'], **kw)
    finally:
        onnx_proto_utils._add_onnxscript_fn = orig
    return buf.getvalue()


@pytest.mark.parametrize('model_name,channels', [('OSNet025', (16, 64, 96, 128))])
@pytest.mark.parametrize('mode', ['eval_folded', 'unfolded'])
def test_osnet_onnx_from_torch_exporter(model_name, channels, mode, tmp_path, monkeypatch):
    import torch
    import torchreid_osnet as T
    net = T.random_osnet(channels, seed=5)
    net.eval()
    x = torch.randn(2, 3, 256, 128)
    if mode == 'eval_folded':
        data = export_with_torch(net, x)
    else:
        data = export_with_torch(net, x, training=torch.onnx.TrainingMode.PRESERVE, do_constant_folding=False,
                                 keep_initializers_as_inputs=True)
    m = onnx_reader.OnnxModel(data)
    ops = {n.op for n in m.nodes}
    assert {'Conv', 'Relu', 'Sigmoid', 'GlobalAveragePool'} <= ops and m.data_inputs[0] == ('input', [2, 3, 256, 128])
    n_bn = sum(n.op == 'BatchNormalization' for n in m.nodes)
    sd = onnx_reader.torchreid_state_dict_from_onnx(m, channels, 512)
    desc = ReID.get_model(model_name)
    g1, _ = desc.build_graph(TorchreidWeights(sd))
    g0, _ = desc.build_graph(TorchreidWeights({k: v.numpy() for k, v in net.state_dict().items()}))
    assert layer_signature(g1) == layer_signature(g0)
    assert len(g1.blob) == len(g0.blob)
    a, b = np.frombuffer(bytes(g1.blob), np.uint8), np.frombuffer(bytes(g0.blob), np.uint8)
    if mode == 'unfolded':
        assert n_bn > 40 and np.array_equal(a, b)                  # the very same parameters: bit-identical engine blob
    else:
        # BatchNorm folded by the exporter in float32 vs folded here: the packed fp16 weights may differ in the last bit
        assert n_bn <= 1
        for (l1, w1, b1), (l0, w0, b0) in zip(g1.conv_params, g0.conv_params):
            assert l1 == l0
            np.testing.assert_allclose(w1, w0, rtol=2e-3, atol=1e-6)
            np.testing.assert_allclose(b1, b0, rtol=1e-4, atol=1e-5)
    # and through the descriptor's MODEL_PATH (the reference's file name, fastmot/models/reid.py:97)
    assert desc.MODEL_PATH.name == 'osnet_x0_25_msmt17.onnx'
    path = tmp_path / desc.MODEL_PATH.name
    path.write_bytes(data)
    monkeypatch.setattr(desc, 'MODEL_PATH', path)
    g2, _ = desc.build_graph()
    assert bytes(g2.blob) == bytes(g1.blob)


def test_osnet_graph_that_does_not_walk_like_osnet_is_refused():
    import torch
    import torchreid_osnet as T
    net = T.random_osnet((16, 64, 96, 128), seed=5).eval()
    data = export_with_torch(net, torch.randn(1, 3, 256, 128))
    with pytest.raises(ValueError, match='expected'):
        onnx_reader.torchreid_state_dict_from_onnx(data, (64, 256, 384, 512), 512)


def test_wire_format_corner_cases():
    """unpacked repeated fields, raw_data tensors, negative ints, truncated input"""
    t = ow.make_tensor('w', ow.TensorProto.FLOAT, [2, 3], np.arange(6), raw=True)
    node = ow.make_node('Conv', ['x', 'w'], ['y'], name='c', pads=[-1, 2, 3, 4], alpha=0.5, auto_pad='SAME_LOWER')
    g = ow.make_graph([node], 'g', [ow.make_tensor_value_info('x', 1, [1, 'N', 4, 4])], [ow.make_tensor_value_info('y', 1, [1, 2, 4, 4])], [t])
    m = onnx_reader.OnnxModel(ow.make_model(g).data)
    np.testing.assert_array_equal(m.tensor('w'), np.arange(6, dtype=np.float32).reshape(2, 3))
    assert m.nodes[0].attrs == {'pads': [-1, 2, 3, 4], 'alpha': 0.5, 'auto_pad': b'SAME_LOWER'}
    assert m.data_inputs == [('x', [1, None, 4, 4])] and m.opset == 11
    with pytest.raises((ValueError, IndexError)):
        onnx_reader.OnnxModel(ow.make_model(g).data[:-7])
    with pytest.raises(KeyError):
        m.tensor('nope')


@pytest.mark.parametrize('model_name', ['OSNet025', 'OSNet10'])
def test_osnet_onnx_embeddings_cpu(model_name):
    """Whole way on the CPU interpreter of the layer table: torch module -> torch's ONNX export (BatchNorm folded) ->
    reader -> layer table -> embeddings == the module's own output."""
    import torch
    import torch_ref
    import torchreid_osnet as T
    desc = ReID.get_model(model_name)
    small = type('Small', (desc,), dict(INPUT_SHAPE=(3, 64, 32)))
    net = T.random_osnet(desc.CHANNELS, seed=2)
    x = torch.from_numpy(np.random.default_rng(1).normal(0, 1, (3, 3, 64, 32)).astype(np.float32))
    with torch.no_grad():
        expect = net(x)
        expect = (expect / expect.norm(dim=1, keepdim=True)).numpy()
    sd = onnx_reader.torchreid_state_dict_from_onnx(export_with_torch(net, x), desc.CHANNELS, 512)
    w = TorchreidWeights(sd)
    g, _ = small.build_graph(w)
    assert w.unused() == []
    _, emb = torch_ref.run_graph(g, x, emulate_fp16_storage=False)
    emb = emb.numpy()
    assert np.abs(emb - expect).max() < 5e-3 and (np.sum(emb * expect, axis=1) > 0.9999).all()


@pytest.mark.gpu
def test_models_from_onnx_files_on_the_engine(ctx, tmp_path, monkeypatch):
    """The reference's deployment: only `.onnx` files in the model directory.  OSNet (torch export) and a Darknet
    detector (reference converter's golden file) load through their descriptors' MODEL_PATH and run on the HIP engine;
    outputs against PyTorch."""
    import torch
    import torch_ref
    import torchreid_osnet as T
    from fastmot_amd.engine import HipNet, NET_DETECTOR, NET_EXTRACTOR
    desc = ReID.get_model('OSNet025')
    small = type('SmallOnnx', (desc,), dict(INPUT_SHAPE=(3, 128, 64)))
    net = T.random_osnet(desc.CHANNELS, seed=4)
    x = torch.from_numpy(np.random.default_rng(3).normal(0, 1, (4, 3, 128, 64)).astype(np.float32))
    with torch.no_grad():
        expect = net(x)
        expect = (expect / expect.norm(dim=1, keepdim=True)).numpy()
    path = tmp_path / 'osnet_x0_25_msmt17.onnx'
    path.write_bytes(export_with_torch(net, x))
    monkeypatch.setattr(small, 'MODEL_PATH', path)
    g, _ = small.build_graph()
    ctx.feat_configure(512)
    eng = HipNet(ctx, NET_EXTRACTOR, g, 4, reuse_buffers=True)
    eng.write(g.input, x.numpy().transpose(0, 2, 3, 1).astype(np.float16))
    eng.run(4)
    emb = eng.read_embeddings(4)
    eng.close()
    assert np.abs(emb - expect).max() < 2e-2 and (np.sum(emb * expect, axis=1) > 0.999).all()

    monkeypatch.setattr(MiniV4, 'MODEL_PATH', GOLDEN / 'yolo2onnx_mini_v4.onnx')
    g, heads = MiniV4.build_graph()
    cfg = darknet.parse_cfg(dc.MINI_V4)
    blob = dc.random_weights_file(cfg, seed=11)
    xin = torch.from_numpy(np.random.default_rng(2).uniform(0, 1, (1, 3, 64, 96)).astype(np.float32))
    ref = dc.torch_darknet(cfg, blob, xin)
    eng = HipNet(ctx, NET_DETECTOR, g, 1, reuse_buffers=False)
    eng.write(g.input, xin.numpy().transpose(0, 2, 3, 1).astype(np.float16))
    eng.run(1)
    for hv, r in zip(heads, ref):
        got = eng.read(hv, 1).transpose(0, 3, 1, 2)
        err = np.abs(got - r.numpy()).max()
        assert err <= 2e-2 * np.abs(r.numpy()).max() + 2e-3, err
    eng.close()
