"""torchreid OSNet checkpoint loader (next row n3 of SURVEY section 8f): the layer table filled from a
state_dict with torchreid's parameter names reproduces the embeddings of an independent PyTorch OSNet of the
same structure (tests/torchreid_osnet.py) -- on the CPU interpreter and on the HIP engine."""
import numpy as np
import pytest
import torch

import torch_ref
import torchreid_osnet as tr
from fastmot_amd.models import ReID
from fastmot_amd.models.torchreid_weights import TorchreidWeights


def setup(model_name, size, seed=1, n=3):
    cls = ReID.get_model(model_name)

    class Small(cls):
        INPUT_SHAPE = (3, *size)
    ref = tr.random_osnet(cls.CHANNELS, seed)
    x = torch.from_numpy(np.random.default_rng(seed).normal(0, 1, (n, 3, *size)).astype(np.float32))
    with torch.no_grad():
        f = ref(x)
        f = f / f.norm(dim=1, keepdim=True)
    return Small, ref, x, f.numpy()


@pytest.mark.parametrize('model_name', ['OSNet025', 'OSNet10'])
@pytest.mark.parametrize('fuse', [False, True])
def test_checkpoint_loader_cpu(model_name, fuse, tmp_path):
    Small, ref, x, expect = setup(model_name, (64, 32))
    path = tmp_path / 'osnet.pth'
    torch.save({'state_dict': {'module.' + k: v for k, v in ref.state_dict().items()}}, path)   # DataParallel-style
    w = TorchreidWeights(path)
    g, _ = Small.build_graph(w, fuse_lightconv=fuse)
    assert w.unused() == []
    _, emb = torch_ref.run_graph(g, x, emulate_fp16_storage=False)
    emb = emb.numpy()
    assert np.abs(emb - expect).max() < 5e-3                       # fp16-rounded weights only
    assert (np.sum(emb * expect, axis=1) > 0.9999).all()


def test_fused_tail_table_is_the_eleven_layers_it_replaces(monkeypatch):
    """At the real 256 x 128 input OSNet x0.25's table ends in ONE FM_OP_OSTAIL layer (Graph.fuse_ostail): its parameter
    blobs have the sizes csrc/ostail.hip indexes, the layers it stands for stay attached for the PyTorch interpreter, and
    that interpreter gives the torchreid module's embeddings for both tables.  x1.0's widths keep their layers."""
    Full, ref, x, expect = setup('OSNet025', (256, 128), n=2)
    sd = {k: v.numpy() for k, v in ref.state_dict().items()}
    embs = {}
    for fused in ('1', '0'):
        monkeypatch.setenv('FASTMOT_OSTAIL', fused)
        w = TorchreidWeights(sd)
        g, _ = Full.build_graph(w)
        assert w.unused() == []
        tail = g.layers[-1]
        if fused == '1':
            assert len(g.layers) == 22 and tail['op'] == 20 and len(tail['sub']) == 11
            assert (tail['cin'], tail['hid'], tail['k'], tail['cout'], tail['stride'], tail['pad']) == (96, 32, 128, 512, 135808, 1928)
            assert tail['w_off'] % 16 == 0 and tail['b_off'] % 16 == 0 and tail['b_off'] + 4 * 1928 <= len(g.blob)
            assert all(i < len(g.layers) - 1 for i, _, _ in g.conv_params)
            g.tables(max_batch=2, reuse=True)
        else:
            assert len(g.layers) == 32 and tail['op'] == 8
        _, emb = torch_ref.run_graph(g, x, emulate_fp16_storage=False)
        embs[fused] = emb.numpy()
        assert np.abs(embs[fused] - expect).max() < 5e-3
    np.testing.assert_array_equal(embs['1'], embs['0'])
    monkeypatch.setenv('FASTMOT_OSTAIL', '1')
    g10, _ = ReID.get_model('OSNet10').build_graph(TorchreidWeights({k: v.numpy() for k, v in
                                                                      tr.random_osnet(ReID.get_model('OSNet10').CHANNELS).state_dict().items()}))
    assert g10.layers[-1]['op'] == 8


def test_checkpoint_mismatch_is_an_error():
    Small, ref, _, _ = setup('OSNet025', (64, 32))
    sd = {k: v.numpy() for k, v in ref.state_dict().items()}
    bad = dict(sd)
    del bad['conv3.0.conv2c.1.conv2.weight']
    with pytest.raises(KeyError):
        Small.build_graph(TorchreidWeights(bad))
    big = tr.random_osnet(ReID.get_model('OSNet10').CHANNELS)
    with pytest.raises(ValueError):
        Small.build_graph(TorchreidWeights({k: v.numpy() for k, v in big.state_dict().items()}))


@pytest.mark.gpu
@pytest.mark.parametrize('model_name', ['OSNet025', 'OSNet10'])
def test_checkpoint_loader_engine(ctx, model_name):
    from fastmot_amd.engine import HipNet, NET_EXTRACTOR
    Small, ref, x, expect = setup(model_name, (128, 64), n=4)
    g, _ = Small.build_graph(TorchreidWeights({k: v.numpy() for k, v in ref.state_dict().items()}))
    ctx.feat_configure(512)
    net = HipNet(ctx, NET_EXTRACTOR, g, 4, reuse_buffers=True)
    net.write(g.input, x.numpy().transpose(0, 2, 3, 1).astype(np.float16))
    net.run(4)
    emb = net.read_embeddings(4)
    net.close()
    assert np.abs(emb - expect).max() < 2e-2
    assert (np.sum(emb * expect, axis=1) > 0.999).all()
