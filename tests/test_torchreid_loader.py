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


def test_one_launch_block_tail_table_cpu(monkeypatch, model_name='OSNet025'):
    """FASTMOT_GATEDCONV=1: gate + gated sum + conv3 + shortcut as one table row (FM_OP_GATEDCONV; blocks whose stream
    chains run as one launch, i.e. the x0.25 widths -- wider blocks keep the two-launch tail).  The row's reference
    semantics reproduce the independent PyTorch OSNet, and its packed weight matrix is the two-segment layout the
    kernel walks: columns [0, mid) over the gated sum, [ceil16(mid), + cin) over the block input, zero elsewhere."""
    from fastmot_amd.models import graph as G
    Small, ref, x, expect = setup(model_name, (64, 32))
    monkeypatch.setenv('FASTMOT_GATEDCONV', '1')
    g, _ = Small.build_graph(TorchreidWeights({k: v.numpy() for k, v in ref.state_dict().items()}))
    rows = [d for d in g.layers if d['op'] == G.OP_GATEDCONV]
    assert len(rows) == 6 and not any(d['op'] == G.OP_GATED_SUM for d in g.layers)
    assert [d['res_mode'] for d in rows] == [G.RES_CONCAT, G.RES_BEFORE_ACT] * 3
    _, emb = torch_ref.run_graph(g, x, emulate_fp16_storage=False)
    emb = emb.numpy()
    assert np.abs(emb - expect).max() < 5e-3 and (np.sum(emb * expect, axis=1) > 0.9999).all()
    blob = np.frombuffer(bytes(g.blob), np.uint8)
    for d in rows:
        c, c2, cout = d['cin'], d['cin2'], d['cout']
        c16, kpad = -(-c // 16) * 16, -(-(-(-c // 16) * 16 + -(-c2 // 16) * 16) // 64) * 64
        w = blob[d['w_off']:d['w_off'] + (-(-cout // 32) * 32) * kpad * 2].view(np.float16).reshape(-1, kpad).astype(np.float32)
        wref = d['conv_ref'][0].reshape(cout, c + c2)
        np.testing.assert_array_equal(w[:cout, :c], wref[:, :c])
        np.testing.assert_array_equal(w[:cout, c16:c16 + c2], wref[:, c:])
        assert not w[:cout, c:c16].any() and not w[:cout, c16 + c2:].any() and not w[cout:].any()


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
