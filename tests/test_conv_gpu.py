"""GPU parity of the conv engine (conv.hip / ops.hip / net.hip) against a plain PyTorch fp32 CPU
reference of the same layer table (tests/torch_ref.py) with the same seeded weights.

Tolerance (fp16 storage, fp32 accumulate -- the reference enables TensorRT FP16 the same way,
models/yolo.py:130-131): per tensor  max|gpu - ref| <= 2e-2 * max|ref| + 2e-3  when the torch
reference also rounds stored activations to fp16; the statement for detections / embeddings is in
DESIGN.md."""
import numpy as np
import pytest
import torch

import torch_ref
from fastmot_amd.engine import HipNet, NET_DETECTOR, NET_EXTRACTOR
from fastmot_amd.models import YOLO, ReID
from fastmot_amd.models.graph import Graph, RandomWeights, RES_BEFORE_ACT

pytestmark = pytest.mark.gpu


def close(gpu, ref, rel=2e-2, abs_=2e-3, what=''):
    ref = np.asarray(ref, np.float32)
    err = np.abs(gpu - ref).max()
    lim = rel * np.abs(ref).max() + abs_
    assert err <= lim, f'{what}: max err {err} > {lim} (ref max {np.abs(ref).max()})'


def nchw(a):
    return torch.from_numpy(np.ascontiguousarray(a.transpose(0, 3, 1, 2)))


def nhwc(t):
    return t.numpy().transpose(0, 2, 3, 1)


@pytest.mark.parametrize('cin,cout,k,stride,act,h,w,n', [
    (8, 32, 3, 1, 'mish', 20, 24, 1),       # stem-like
    (64, 64, 1, 1, 'leaky', 19, 19, 1),
    (32, 64, 3, 2, 'mish', 38, 38, 1),      # downsample
    (128, 255, 1, 1, 'linear', 13, 13, 1),  # head-like, ragged cout
    (256, 512, 3, 1, 'leaky', 19, 19, 1),   # big K
    (16, 16, 1, 1, 'relu', 64, 32, 5),      # OSNet x0.25 pointwise, batch
    (24, 96, 1, 1, 'relu', 16, 8, 7),
    (8, 16, 7, 2, 'relu', 64, 32, 3),       # OSNet conv1 (7x7 s2 p3)
    (512, 128, 1, 1, 'swish', 9, 11, 2),
])
def test_single_conv(ctx, cin, cout, k, stride, act, h, w, n):
    rng = np.random.default_rng(cin * 1000 + cout)
    g = Graph(RandomWeights(seed=cin + cout + k), (h, w), cin)
    res = None
    y = g.conv('c', g.input, cout, k, stride, act, pad=3 if k == 7 else None)
    net = HipNet(ctx, NET_DETECTOR, g, n)
    x = rng.normal(0, 1, (n, h, w, cin)).astype(np.float16)
    net.write(g.input, x)
    net.run(n)
    out = net.read(y, n)
    bufs, _ = torch_ref.run_graph(g, nchw(x.astype(np.float32)))
    close(out, nhwc(bufs[y.tid][:, :cout]), what=f'conv {cin}->{cout} k{k}')
    net.close()


@pytest.mark.parametrize('cin,cout,k,stride,act,h,w,n,extra', [
    (512, 1024, 3, 1, 'leaky', 19, 19, 1, ''),      # 2 cout tiles per workgroup, 8 waves
    (1024, 255, 1, 1, 'linear', 19, 19, 2, 'f32'),  # YOLO head: ragged cout, fp32 output, batch
    (256, 512, 3, 2, 'leaky', 38, 38, 1, ''),       # stride-2 downsample into the 19 x 19 level
    (512, 512, 1, 1, 'mish', 19, 19, 1, 'res'),     # shortcut after the activation
    (512, 256, 1, 1, 'leaky', 19, 19, 1, 'up'),     # fused nearest x2 upsample into a concat slice
    (128, 64, 3, 1, 'relu', 7, 5, 3, 'res'),        # 1 cout tile per workgroup, ragged pixel tile, batch
    (64, 40, 3, 1, 'swish', 9, 9, 1, ''),           # 4 waves (K = 576: 9 chunks), cout padded to 64
    (2048, 512, 1, 1, 'leaky', 19, 19, 1, ''),
])
def test_streamed_conv(ctx, cin, cout, k, stride, act, h, w, n, extra):
    """Streamed conv (K split inside the workgroup, convs.hip) == LDS-tiled conv + split-K reduce == torch."""
    rng = np.random.default_rng(cin + cout)
    x = rng.normal(0, 1, (n, h, w, cin)).astype(np.float16)
    outs = []
    for maxp in (10 ** 6, 0):
        g = Graph(RandomWeights(seed=cin + 3 * cout + k), (h, w), cin)
        g.convs_max_pixels = maxp
        g.convd_level = 0
        ho = (h + 2 * (k // 2) - k) // stride + 1
        wo = (w + 2 * (k // 2) - k) // stride + 1
        up = 2 if extra == 'up' else 1
        wide = g.new(ho * up, wo * up, cout + 16, f32=extra == 'f32')
        res = g.conv('r', g.input, cout, 1, stride, 'linear') if extra == 'res' else None
        y = g.conv('c', g.input, cout, k, stride, act, dst=wide.slice(16, cout), res=res, up=up, f32_out=extra == 'f32')
        assert g.layers[-1]['op'] == (15 if maxp else 0)
        net = HipNet(ctx, NET_DETECTOR, g, n)
        net.write(g.input, x)
        net.run(n)
        outs.append(net.read(wide, n)[..., 16:16 + cout])
        bufs, _ = torch_ref.run_graph(g, nchw(x.astype(np.float32)))
        close(outs[-1], nhwc(bufs[wide.tid][:, 16:16 + cout]), what=f'conv maxp={maxp}')
        net.close()
    close(outs[0], outs[1], rel=1e-2, abs_=2e-3, what='streamed vs tiled')


@pytest.mark.parametrize('cin,cout,k,stride,act,h,w,n', [
    (3, 32, 3, 1, 'mish', 40, 37, 1),        # YOLOv4 layer 0
    (3, 16, 7, 2, 'relu', 64, 32, 3),        # OSNet conv1
    (3, 64, 7, 2, 'relu', 32, 16, 1),        # OSNet x1.0 conv1: cout > 32 -> generic kernel
    (4, 24, 3, 2, 'leaky', 33, 18, 2),
    (1, 8, 3, 1, 'linear', 16, 16, 1),
])
def test_stem_conv(ctx, cin, cout, k, stride, act, h, w, n):
    """LDS-patch stem kernel == generic implicit-GEMM kernel (same fp16 weights; different summation
    order) == torch."""
    rng = np.random.default_rng(cin + cout)
    x = rng.normal(0, 1, (n, h, w, cin)).astype(np.float16)
    outs = []
    for stem in (True, False):
        g = Graph(RandomWeights(seed=k + cout), (h, w), cin)
        g.use_stem = stem
        y = g.conv('c', g.input, cout, k, stride, act, pad=3 if k == 7 else None)
        assert (g.layers[0]['op'] == 12) == (stem and cout <= 32)
        net = HipNet(ctx, NET_DETECTOR, g, n)
        net.write(g.input, x)
        net.run(n)
        outs.append(net.read(y, n))
        bufs, _ = torch_ref.run_graph(g, nchw(x.astype(np.float32)))
        close(outs[-1], nhwc(bufs[y.tid][:, :cout]), what=f'stem conv {cin}->{cout} k{k} stem={stem}')
        net.close()
    assert np.abs(outs[0] - outs[1]).max() <= 4e-3 * max(np.abs(outs[1]).max(), 1.0)


def test_splitk_repeated_runs(ctx):
    """Split-K layers hand their fp32 partials to the last-arriving workgroup through a workspace that
    every layer and every run reuses: results must track the inputs run after run (no stale cached
    partials, counters self-reset) and be deterministic."""
    g = Graph(RandomWeights(seed=31), (19, 19), 256)
    a = g.conv('a', g.input, 512, 3, 1, 'leaky')
    b = g.conv('b', a, 256, 1, 1, 'leaky')
    c = g.conv('c', b, 512, 3, 1, 'mish', res=a)
    net = HipNet(ctx, NET_DETECTOR, g, 1)
    rng = np.random.default_rng(32)
    for it in range(4):
        x = rng.normal(0, 1 + it, (1, 19, 19, 256)).astype(np.float16)
        net.write(g.input, x)
        net.run(1)
        first = net.read(c, 1)
        bufs, _ = torch_ref.run_graph(g, nchw(x.astype(np.float32)))
        close(first, nhwc(bufs[c.tid][:, :512]), what=f'split-K run {it}')
        net.run(1)
        np.testing.assert_array_equal(first, net.read(c, 1))
    net.close()


def test_conv_residual_concat_fp32_out(ctx):
    rng = np.random.default_rng(5)
    g = Graph(RandomWeights(seed=9), (24, 24), 32)
    cat = g.new(24, 24, 96)
    a = g.conv('a', g.input, 32, 1, 1, 'mish', dst=cat.slice(64, 32))
    b = g.conv('b', g.input, 64, 3, 1, 'mish')
    b2 = g.conv('b2', b, 64, 3, 1, 'mish', res=b)                          # shortcut after activation
    g.conv('b3', b2, 64, 1, 1, 'relu', dst=cat.slice(0, 64), res=b, res_mode=RES_BEFORE_ACT)
    o = g.conv('o', cat, 21, 1, 1, 'linear', bn=False, f32_out=True)
    net = HipNet(ctx, NET_DETECTOR, g, 2)
    x = rng.normal(0, 1, (2, 24, 24, 32)).astype(np.float16)
    net.write(g.input, x)
    net.run(2)
    bufs, _ = torch_ref.run_graph(g, nchw(x.astype(np.float32)))
    close(net.read(cat, 2), nhwc(bufs[cat.tid][:, :96]), what='concat')
    close(net.read(o, 2), nhwc(bufs[o.tid][:, :21]), what='fp32 head')
    net.close()


def test_pool_upsample_dw_gate_ops(ctx):
    rng = np.random.default_rng(6)
    g = Graph(RandomWeights(seed=3), (16, 12), 32)
    spp = g.new(16, 12, 128)
    x0 = g.conv('c', g.input, 32, 1, 1, 'leaky', dst=spp.slice(96, 32))
    g.pool(x0, 13, 1, 6, dst=spp.slice(0, 32))
    g.pool(x0, 9, 1, 4, dst=spp.slice(32, 32))
    g.pool(x0, 5, 1, 2, dst=spp.slice(64, 32))
    up = g.upsample2(x0)
    mp = g.pool(up, 3, 2, 1)
    ap = g.pool(up, 2, 2, 0, avg=True)
    dw = g.dwconv3('dw', ap, 'relu')
    gid1, gp = g.gate('gate', dw, 2)
    gid2, _ = g.gate('gate', mp, 2, gp)
    gs = g.gate_sum([dw, mp], [gid1, gid2])
    net = HipNet(ctx, NET_DETECTOR, g, 3)
    x = rng.normal(0, 1, (3, 16, 12, 32)).astype(np.float16)
    net.write(g.input, x)
    net.run(3)
    bufs, _ = torch_ref.run_graph(g, nchw(x.astype(np.float32)))
    for name, v in (('spp', spp), ('up', up), ('maxpool', mp), ('avgpool', ap), ('dwconv', dw), ('gate_sum', gs)):
        close(net.read(v, 3), nhwc(bufs[v.tid][:, v.coff:v.coff + v.c]), what=name)
    net.close()


@pytest.mark.parametrize('c,h,w,n', [(512, 19, 19, 1), (16, 7, 5, 3), (64, 40, 23, 2)])
def test_spp_fused(ctx, c, h, w, n):
    """Fused SPP (k = 13, 9, 5 cascade in LDS) == three independent max-pool launches == torch."""
    rng = np.random.default_rng(c)
    x = rng.normal(0, 1, (n, h, w, c)).astype(np.float16)
    g = Graph(RandomWeights(seed=1), (h, w), c)
    fused = g.new(h, w, 4 * c)
    g.spp(g.input, fused.slice(c, 3 * c))
    assert g.layers[-1]['op'] == 10
    plain = g.new(h, w, 3 * c)
    for i, k in enumerate((13, 9, 5)):
        g.pool(g.input, k, 1, k // 2, dst=plain.slice(i * c, c))
    net = HipNet(ctx, NET_DETECTOR, g, n)
    net.write(g.input, x)
    net.run(n)
    a, b = net.read(fused, n)[..., c:], net.read(plain, n)
    np.testing.assert_array_equal(a, b)
    bufs, _ = torch_ref.run_graph(g, nchw(x.astype(np.float32)))
    np.testing.assert_array_equal(a, nhwc(bufs[plain.tid][:, :3 * c]))
    net.close()


@pytest.mark.parametrize('c,mid,h,w,n', [(64, 32, 19, 27, 2), (64, 64, 40, 40, 1), (128, 128, 76, 76, 1),
                                         (128, 64, 9, 5, 3), (256, 256, 38, 38, 1), (256, 128, 7, 13, 2)])
def test_resblock_fused(ctx, c, mid, h, w, n):
    """Fused residual unit == 1x1 conv -> 3x3 conv + shortcut as separate layers (same parameters), and ==
    the torch reference; input/output in channel slices of wider tensors."""
    rng = np.random.default_rng(c + mid)
    x = rng.normal(0, 1, (n, h, w, c)).astype(np.float16)
    g = Graph(RandomWeights(seed=5), (h, w), c + 8)
    xin = g.input.slice(8, c)
    wide = g.new(h, w, c + 16)
    fused = g.resblock('a', 'b', xin, mid, dst=wide.slice(16, c))
    assert g.layers[-1]['op'] == 14
    w1, b1, w2, b2 = g.layers[-1]['res_ref']
    t = g.conv('a2', xin, mid, 1, 1, 'mish', wb=(w1, b1))
    plain = g.conv('b2', t, c, 3, 1, 'mish', res=xin, wb=(w2, b2))
    net = HipNet(ctx, NET_DETECTOR, g, n)
    xi = np.zeros((n, h, w, c + 8), np.float16)
    xi[..., 8:] = x
    net.write(g.input, xi)
    net.run(n)
    a, b = net.read(wide, n)[..., 16:], net.read(plain, n)
    close(a, b, what='fused vs unfused')
    bufs, _ = torch_ref.run_graph(g, nchw(xi.astype(np.float32)))
    close(a, nhwc(bufs[wide.tid][:, 16:16 + c]), what='fused vs torch')
    net.close()


@pytest.mark.parametrize('cin,cout,h,w,n', [(512, 256, 19, 19, 1), (64, 24, 6, 9, 2)])
def test_conv_fused_upsample(ctx, cin, cout, h, w, n):
    """conv(up=2) == conv followed by the nearest x2 upsample layer (split-K path for the first case)."""
    rng = np.random.default_rng(cin)
    x = rng.normal(0, 1, (n, h, w, cin)).astype(np.float16)
    g = Graph(RandomWeights(seed=2), (h, w), cin)
    cat = g.new(2 * h, 2 * w, cout + 8)
    g.conv('c', g.input, cout, 1, 1, 'leaky', dst=cat.slice(8, cout), up=2)
    g2 = Graph(RandomWeights(seed=2), (h, w), cin)
    y = g2.conv('c', g2.input, cout, 1, 1, 'leaky')
    u = g2.upsample2(y)
    outs = []
    for gg, v in ((g, cat), (g2, u)):
        net = HipNet(ctx, NET_DETECTOR, gg, n)
        net.write(gg.input, x)
        net.run(n)
        outs.append(net.read(v, n))
        net.close()
    np.testing.assert_array_equal(outs[0][..., 8:8 + cout], outs[1][..., :cout])
    bufs, _ = torch_ref.run_graph(g, nchw(x.astype(np.float32)))
    close(outs[0][..., 8:8 + cout], nhwc(bufs[cat.tid][:, 8:8 + cout]), what='conv up2')


@pytest.mark.parametrize('c,h,w,n', [(16, 64, 32, 3), (24, 32, 16, 5), (32, 16, 8, 7), (16, 21, 19, 2),
                                     (8, 5, 3, 1), (64, 20, 17, 2), (96, 32, 16, 2), (128, 16, 8, 3),
                                     (72, 9, 24, 1)])
def test_lightconv_fused(ctx, c, h, w, n):
    """Fused LightConv3x3 (liteconv.hip) == pointwise conv + depthwise kernel pair, bit for bit, and
    both within tolerance of the torch reference; ragged tiles, halo at every border."""
    rng = np.random.default_rng(c + h)
    x = rng.normal(0, 1, (n, h, w, c)).astype(np.float16)
    outs = []
    for fuse in (True, False):
        g = Graph(RandomWeights(seed=c), (h, w), c)
        y = g.lightconv('lc', g.input, c, 'relu', fuse=fuse)
        assert len(g.layers) == (1 if fuse else 2)
        net = HipNet(ctx, NET_DETECTOR, g, n)
        net.write(g.input, x)
        net.run(n)
        outs.append(net.read(y, n))
        bufs, _ = torch_ref.run_graph(g, nchw(x.astype(np.float32)))
        close(outs[-1], nhwc(bufs[y.tid][:, :c]), what=f'lightconv c{c} fuse={fuse}')
        net.close()
    np.testing.assert_array_equal(outs[0], outs[1])


def test_osnet_fused_and_arena_reuse_identical(ctx):
    """The production configuration (stream chains in one launch, gate + gated sum in one launch, conv3 + shortcut: 32
    launches, activation arena shared between tensors with disjoint live ranges) against the layer-per-kernel graph (173
    launches, private buffers): arena reuse changes nothing bit for bit; the fused graph differs only by the summation
    grouping of the gate's average pool."""
    class Small(ReID.get_model('OSNet025')):
        INPUT_SHAPE = (3, 128, 64)
    rng = np.random.default_rng(3)
    x = rng.normal(0, 1, (6, 128, 64, 3)).astype(np.float16)
    ctx.feat_configure(512)
    embs = {}
    for fuse, reuse in ((False, False), (True, False), (True, True)):
        g, _ = Small.build_graph(RandomWeights(seed=5), fuse_lightconv=fuse)
        assert len(g.layers) == (32 if fuse else 173)   # 50 with one launch per stream depth (next test)
        net = HipNet(ctx, NET_EXTRACTOR, g, 6, reuse_buffers=reuse)
        for _ in range(2):                      # second run: stale arena contents must not matter
            net.write(g.input, x)
            net.run(6)
        embs[fuse, reuse] = net.read_embeddings(6)
        net.close()
    np.testing.assert_array_equal(embs[True, False], embs[True, True])
    assert np.abs(embs[True, True] - embs[False, False]).max() < 2e-3
    assert (np.sum(embs[True, True] * embs[False, False], axis=1) > 0.9999).all()


@pytest.mark.parametrize('c,hid,h,w,n,k', [(16, 1, 64, 32, 3, 4), (24, 1, 32, 16, 5, 4), (32, 2, 16, 8, 2, 4),
                                           (128, 8, 16, 8, 2, 3), (64, 4, 9, 7, 1, 2)])
def test_gated_sum_fused(ctx, c, hid, h, w, n, k):
    """FM_OP_GATED_SUM == k gate launches + gate_sum (up to the average-pool grouping) == torch."""
    rng = np.random.default_rng(c)
    x = rng.normal(0, 1, (n, h, w, k * c)).astype(np.float16)
    outs = []
    for fused in (True, False):
        g = Graph(RandomWeights(seed=c), (h, w), k * c)
        xs = [g.input.slice(i * c, c) for i in range(k)]
        if fused:
            y = g.gated_sum('gate', xs, hid)
        else:
            gp, gids = None, []
            for v in xs:
                gid, gp = g.gate('gate', v, hid, gp)
                gids.append(gid)
            y = g.gate_sum(xs, gids)
        net = HipNet(ctx, NET_DETECTOR, g, n)
        net.write(g.input, x)
        net.run(n)
        outs.append(net.read(y, n))
        bufs, _ = torch_ref.run_graph(g, nchw(x.astype(np.float32)))
        close(outs[-1], nhwc(bufs[y.tid][:, :c]), what=f'gated_sum fused={fused}')
        net.close()
    assert np.abs(outs[0] - outs[1]).max() <= 4e-3 * np.abs(outs[1]).max()


@pytest.mark.parametrize('c,h,w,n', [(16, 64, 32, 3), (24, 32, 16, 5), (32, 16, 8, 4), (16, 21, 19, 2), (8, 9, 40, 1),
                                     (16, 64, 32, 13), (16, 64, 32, 8), (24, 32, 16, 40)])
def test_lightconv_chain(ctx, c, h, w, n):
    """The four LightConv streams of an OSNet block in one launch (litechain.hip) == one grouped launch per
    depth, BIT FOR BIT (same fp16 rounding points, same MFMA order), incl. the gate's per-tile channel sums
    (checked through the gated sum that consumes them); and == the torch reference."""
    rng = np.random.default_rng(c + h)
    x = rng.normal(0, 1, (n, h, w, c + 8)).astype(np.float16)
    outs = []
    for chain in (True, False):
        g = Graph(RandomWeights(seed=9), (h, w), c + 8)
        x1 = g.input.slice(8, c)
        params = {(t, i): g.lightconv_params(f's{t}.{i}', c) for t in range(1, 5) for i in range(t)}
        if chain:
            assert Graph.lightchain_fits(c, h, w)
            y = g.lightchain('streams', x1, [params[(t, i)] for t in range(1, 5) for i in range(t)], 'relu')
            assert g.layers[-1]['op'] == 16
            streams, parts = [y.slice(t * c, c) for t in range(4)], g.last_gap_slots
        else:
            streams, parts, prev = [], [], None
            for i in range(4):
                ts = list(range(i + 1, 5))
                xs = [x1] * len(ts) if i == 0 else [prev.slice((t - i) * c, c) for t in ts]
                prev = g.lightconv_group(f'depth{i}', xs, [params[(t, i)] for t in ts], 'relu', gap_slot=True)
                streams.append(prev.slice(0, c))
                parts.append(g.last_gap_slot)
        z = g.gated_sum('gate', streams, max(c // 16, 1), parts=parts)
        net = HipNet(ctx, NET_DETECTOR, g, n)
        net.write(g.input, x)
        net.run(n)
        outs.append([net.read(v, n) for v in streams] + [net.read(z, n)])
        if chain:
            bufs, _ = torch_ref.run_graph(g, nchw(x.astype(np.float32)))
            for t, v in enumerate(streams):
                close(outs[-1][t], nhwc(bufs[v.tid][:, v.coff:v.coff + c]), what=f'stream {t}')
        net.close()
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a, b)


def test_lightconv_chain_launch_shape_independent(ctx):
    """litechain.hip runs launches with few workgroups as 512- or 1024-thread workgroups (256 otherwise): a
    sample's streams and gated sum are bit-identical whatever the batch it is part of."""
    c, h, w, nmax = 16, 64, 32, 13                      # 8 tiles x 4 streams per sample
    rng = np.random.default_rng(5)
    x = rng.normal(0, 1, (nmax, h, w, c + 8)).astype(np.float16)
    g = Graph(RandomWeights(seed=9), (h, w), c + 8)
    x1 = g.input.slice(8, c)
    params = [g.lightconv_params(f's{t}.{i}', c) for t in range(1, 5) for i in range(t)]
    y = g.lightchain('streams', x1, params, 'relu')
    z = g.gated_sum('gate', [y.slice(t * c, c) for t in range(4)], 1, parts=g.last_gap_slots)
    net = HipNet(ctx, NET_DETECTOR, g, nmax)
    outs = {}
    for n in (13, 8, 2):                                 # 416 / 256 / 64 workgroups -> 256 / 512 / 1024 threads
        net.write(g.input, x[:n])
        net.run(n)
        outs[n] = (net.read(y, n)[:2].copy(), net.read(z, n)[:2].copy())
    net.close()
    for n in (8, 2):
        np.testing.assert_array_equal(outs[13][0], outs[n][0])
        np.testing.assert_array_equal(outs[13][1], outs[n][1])


def test_lightconv_grouped(ctx):
    """Four LightConvs in one launch (blockIdx.y = group) == four single launches, bit for bit."""
    c, h, w, n = 16, 32, 16, 4
    rng = np.random.default_rng(1)
    x = rng.normal(0, 1, (n, h, w, 2 * c)).astype(np.float16)
    g = Graph(RandomWeights(seed=4), (h, w), 2 * c)
    ins = [g.input.slice(0, c), g.input.slice(c, c), g.input.slice(0, c)]
    params = [g.lightconv_params(f'p{i}', c) for i in range(3)]
    grouped = g.lightconv_group('grp', ins, params)
    singles = [g.lightconv_group(f's{i}', [ins[i]], [params[i]]) for i in range(3)]
    net = HipNet(ctx, NET_DETECTOR, g, n)
    net.write(g.input, x)
    net.run(n)
    full = net.read(grouped, n)
    bufs, _ = torch_ref.run_graph(g, nchw(x.astype(np.float32)))
    for i, s in enumerate(singles):
        np.testing.assert_array_equal(full[..., i * c:(i + 1) * c], net.read(s, n))
    close(full, nhwc(bufs[grouped.tid][:, :3 * c]), what='grouped lightconv')
    net.close()


def test_yolov4_arena_reuse_identical(ctx):
    class Small(YOLO.get_model('YOLOv4')):
        INPUT_SHAPE = (3, 128, 160)
    x = np.random.default_rng(8).uniform(0, 1, (1, 128, 160, 3)).astype(np.float16)
    outs = []
    for reuse in (False, True):
        g, heads = Small.build_graph(RandomWeights(seed=2))
        net = HipNet(ctx, NET_DETECTOR, g, 1, reuse_buffers=reuse)
        if reuse:
            assert g.arena_bytes < 0.5 * sum(h * w * c * (4 if f else 2) for h, w, c, f in g.tensors)
        for _ in range(2):
            net.write(g.input, x)
            net.run(1)
        outs.append([net.read(h, 1) for h in heads])
        net.close()
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize('model,size,batch', [('OSNet025', (64, 32), 5), ('OSNet10', (64, 32), 3)])
def test_osnet_embeddings(ctx, model, size, batch):
    cls = ReID.get_model(model)

    class Small(cls):
        INPUT_SHAPE = (3, *size)
    g, _ = Small.build_graph(RandomWeights(seed=11))
    ctx.feat_configure(512)
    net = HipNet(ctx, NET_EXTRACTOR, g, batch)
    rng = np.random.default_rng(12)
    x = rng.normal(0, 1, (batch, *size, 3)).astype(np.float16)
    net.write(g.input, x)
    net.run(batch)
    emb = net.read_embeddings(batch)
    _, ref = torch_ref.run_graph(g, nchw(x.astype(np.float32)))
    np.testing.assert_allclose(np.linalg.norm(emb, axis=1), 1.0, atol=1e-5)
    # embeddings: unit vectors; tolerance 2e-2 absolute per component, cosine similarity > 0.999
    assert np.abs(emb - ref.numpy()).max() < 2e-2
    assert (np.sum(emb * ref.numpy(), axis=1) > 0.999).all()
    net.close()


@pytest.mark.parametrize('batch', [1, 3])
def test_osnet_tail_fused(ctx, monkeypatch, batch):
    """FM_OP_OSTAIL (ostail.hip: OSNet x0.25's 16 x 8 stage, conv5 and head as one launch, one workgroup per sample) against
    the eleven launches it replaces and against PyTorch.  Every stored tensor of the unfused path is rounded to fp16 at the
    same point and the MFMA K order is the same; the two average pools (gate, head) add their fp32 terms in another order."""
    cls = ReID.get_model('OSNet025')
    rng = np.random.default_rng(21 + batch)
    x = rng.normal(0, 1, (batch, 256, 128, 3)).astype(np.float16)
    ctx.feat_configure(512)
    embs = []
    for fused in ('1', '0'):
        monkeypatch.setenv('FASTMOT_OSTAIL', fused)
        g, _ = cls.build_graph(RandomWeights(seed=17))
        assert (g.layers[-1]['op'] == 20) == (fused == '1') and len(g.layers) == (22 if fused == '1' else 32)
        net = HipNet(ctx, NET_EXTRACTOR, g, batch)
        for _ in range(2):                       # the second run starts from a used LDS / arena
            net.write(g.input, x)
            net.run(batch)
        embs.append(net.read_embeddings(batch))
        net.close()
        _, ref = torch_ref.run_graph(g, nchw(x.astype(np.float32)))
        assert np.abs(embs[-1] - ref.numpy()).max() < 4e-3, (fused, np.abs(embs[-1] - ref.numpy()).max())
    np.testing.assert_allclose(np.linalg.norm(embs[0], axis=1), 1.0, atol=1e-5)
    assert np.abs(embs[0] - embs[1]).max() < 2e-4, np.abs(embs[0] - embs[1]).max()
    assert (np.sum(embs[0] * embs[1], axis=1) > 0.999999).all()


@pytest.mark.parametrize('resblock', ['1', '0'])
def test_yolov4_small_input(ctx, monkeypatch, resblock):
    """Whole YOLOv4 topology (110 convs, SPP, PAN) at 96x96 so the CPU reference finishes in seconds;
    with the residual units fused (19 of the 23: 64..256 channels) and as separate conv layers."""
    monkeypatch.setenv('FASTMOT_RESBLOCK', resblock)

    class Small(YOLO.get_model('YOLOv4')):
        INPUT_SHAPE = (3, 96, 96)
    g, heads = Small.build_graph(RandomWeights(seed=21))
    n_res = sum(d['op'] == 14 for d in g.layers)
    assert n_res == (19 if resblock == '1' else 0)
    # 110 conv layers of yolov4.cfg; the two sibling 1x1 convs of each of the 5 CSP stages run as one
    # ... and the stem with the stride-2 conv and the first stage's merged 1x1 conv behind it (FM_OP_STEM2 = 18: three convs
    # in one entry), two pointwise convs around a concat (FM_OP_PAIR11 = 19: two; only on maps of >= 8192 pixels)
    convs = {0: 1, 12: 1, 15: 1, 17: 1, 14: 2, 19: 2}
    n_convs = sum(convs.get(d['op'], 0) for d in g.layers) + sum(len(d['gates']) for d in g.layers if d['op'] == 18)
    assert n_convs == 110 - 5
    assert g.layers[0]['op'] == 18 and len(g.layers[0]['gates']) == 3
    net = HipNet(ctx, NET_DETECTOR, g, 1)
    rng = np.random.default_rng(22)
    x = rng.uniform(0, 1, (1, 96, 96, 3)).astype(np.float16)
    net.write(g.input, x)
    net.run(1)
    bufs, _ = torch_ref.run_graph(g, nchw(x.astype(np.float32)))
    for i, hd in enumerate(heads):
        close(net.read(hd, 1), nhwc(bufs[hd.tid][:, :hd.c]), rel=3e-2, abs_=5e-3, what=f'head {i}')
    # fp16-storage engine vs PURE fp32 reference (the fp tolerance quoted for detections)
    bufs32, _ = torch_ref.run_graph(g, nchw(x.astype(np.float32)), emulate_fp16_storage=False)
    for i, hd in enumerate(heads):
        close(net.read(hd, 1), nhwc(bufs32[hd.tid][:, :hd.c]), rel=5e-2, abs_=1e-2, what=f'head {i} vs fp32')
    net.close()


def _convd_code(bm, bn, kg, ns=0, role=0, spb=1):
    return bm | bn << 8 | kg << 16 | ns << 20 | role << 24 | (spb - 1) << 25


_CONVD_SHAPES = [
    (64, 128, 1, 1, 'mish', 76, 76, 1, ''),            # CSP stage entry
    (64, 64, 1, 1, 'mish', 48, 40, 1, 'slice'),        # reads a channel slice, writes a concat slice
    (128, 64, 1, 1, 'mish', 35, 35, 1, ''),            # ragged last pixel tile
    (128, 128, 1, 1, 'leaky', 32, 32, 2, ''),          # batch 2
    (256, 255, 1, 1, 'linear', 38, 38, 1, 'f32'),      # head: ragged cout, fp32 output
    (128, 255, 1, 1, 'logistic', 20, 20, 1, 'f32'),    # NEW_COORDS head
    (512, 256, 1, 1, 'leaky', 19, 19, 1, 'up'),        # fused nearest x2 upsample into a concat slice
    (512, 512, 1, 1, 'mish', 19, 19, 1, 'res'),        # shortcut after the activation
    (2048, 512, 1, 1, 'leaky', 19, 19, 1, ''),         # 32 K steps
    (128, 256, 3, 1, 'leaky', 38, 38, 1, ''),          # 3x3, padding on every border
    (64, 128, 3, 2, 'mish', 76, 76, 1, ''),            # stride-2 downsample, 9 K steps (odd: two steps per barrier repeat one)
    (256, 512, 3, 2, 'leaky', 38, 38, 1, ''),
    (512, 1024, 3, 1, 'leaky', 19, 19, 1, ''),         # 72 K steps
    (128, 64, 3, 1, 'relu', 7, 5, 3, 'res'),           # tiny maps, batch, shortcut
    (64, 40, 3, 1, 'swish', 9, 9, 1, ''),              # cout padded to 64
    (192, 96, 3, 1, 'relu', 16, 8, 4, 'resb'),         # cin = 3 * 64: a tap is three K steps; shortcut BEFORE the activation
    (16, 64, 1, 1, 'relu', 64, 32, 5, ''),             # OSNet x0.25 pointwise convs: cin % 64 != 0 (K zero-padded to a step,
    (96, 24, 1, 1, 'relu', 16, 8, 7, 'resb'),          #   chunks beyond the pixel's channels not fetched), batch
    (24, 96, 1, 1, 'linear', 32, 16, 3, 'slice'),
]
# forced configurations (bm, bn, K groups, ring slots, loader waves, steps per barrier); every shape runs the launcher's own
# choice and three of them (a different three per shape: every pairing of a shape class with a code path occurs)
_CONVD_CFGS = [(128, 128, 1, 0, 1), (128, 64, 2, 0, 1), (128, 64, 1, 2, 0), (64, 64, 4), (64, 64, 1, 3, 1), (64, 64, 2, 0, 0),
               (128, 128, 2, 2, 1), (64, 64, 2, 2, 1, 2), (128, 64, 1, 3, 1, 2), (64, 64, 1, 0, 0, 2)]
_CONVD_CASES = [(*sh, cfg) for i, sh in enumerate(_CONVD_SHAPES)
                for cfg in ['auto'] + [_CONVD_CFGS[(3 * i + j) % len(_CONVD_CFGS)] for j in range(3)]]


@pytest.mark.parametrize('cin,cout,k,stride,act,h,w,n,extra,cfg', _CONVD_CASES)
def test_convd_conv(ctx, cin, cout, k, stride, act, h, w, n, extra, cfg):
    """convd.hip (operands by DMA into an LDS ring, up to 2 x 2 accumulators per wave, K groups, loader waves) against the
    LDS-tiled kernel on the same layer and against PyTorch, under forced (tile, K groups, ring depth, loader waves, steps per
    barrier) configurations and the launcher's own choice."""
    rng = np.random.default_rng(cin + cout + h)
    sl = extra == 'slice'
    x = rng.normal(0, 1, (n, h, w, cin + (64 if sl else 0))).astype(np.float16)
    outs = []
    for level in (2, 0):
        ctx.set_option('convd_cfg', 0 if cfg == 'auto' or not level else _convd_code(*cfg))
        g = Graph(RandomWeights(seed=cin + 3 * cout + k), (h, w), x.shape[-1])
        g.convd_level = level
        g.convs_max_pixels = 0
        src = g.input.slice(64, cin) if sl else g.input
        ho = (h + 2 * (k // 2) - k) // stride + 1
        wo = (w + 2 * (k // 2) - k) // stride + 1
        up = 2 if extra == 'up' else 1
        wide = g.new(ho * up, wo * up, cout + 64, f32=extra == 'f32')
        res = g.conv('r', src, cout, 1, stride, 'linear') if extra in ('res', 'resb') else None
        y = g.conv('c', src, cout, k, stride, act, dst=wide.slice(64, cout), res=res, up=up, f32_out=extra == 'f32',
                   res_mode=RES_BEFORE_ACT if extra == 'resb' else 1, bn=extra != 'f32')
        assert g.layers[-1]['op'] == (17 if level else 0)
        net = HipNet(ctx, NET_DETECTOR, g, n)
        for xi in (x[:, ::-1].copy(), x):           # eager validation + capture, then a graph replay -- on another input, so
            net.write(g.input, xi)                  # that a replay which did nothing could not pass on the first launch's output
            net.run(n)
        outs.append(net.read(wide, n)[..., 64:64 + cout])
        if level:
            bufs, _ = torch_ref.run_graph(g, nchw(x.astype(np.float32)))
            close(outs[-1], nhwc(bufs[wide.tid][:, 64:64 + cout]), what=f'convd {cin}->{cout} k{k}s{stride} {cfg}')
        net.close()
    ctx.set_option('convd_cfg', 0)
    assert np.isfinite(outs[0]).all() and np.abs(outs[0]).max() > 0
    close(outs[0], outs[1], rel=1e-2, abs_=2e-3, what='convd vs tiled')


@pytest.mark.parametrize('cout,act1,act2,h,w,n,c3', [(64, 'mish', 'mish', 64, 96, 1, 0), (64, 'leaky', 'mish', 37, 51, 2, 0),
                                                    (128, 'mish', 'leaky', 24, 40, 1, 0), (64, 'mish', 'mish', 608, 608, 1, 0),
                                                    (64, 'mish', 'mish', 37, 51, 2, 128), (64, 'mish', 'leaky', 64, 96, 1, 64),
                                                    (64, 'mish', 'mish', 608, 608, 1, 128)])
def test_stem_pair_fused(ctx, cout, act1, act2, h, w, n, c3):
    """FM_OP_STEM2 (stem2.hip): the stem conv (3 -> 32, 3x3 s1) and the 3x3 stride-2 conv behind it in one launch -- and,
    c3 > 0, the pointwise conv behind that as a third stage -- == the layers of the unfused table (same parameters: the
    stem's fp16 values are the same by construction, the second conv accumulates in another order) == the torch reference;
    odd map sizes, batch 2, YOLOv4's own 608 x 608."""
    rng = np.random.default_rng(cout + h)
    x = np.zeros((n, h, w, 8), np.float16)
    x[..., :3] = rng.uniform(0, 1, (n, h, w, 3))
    outs = []
    for fuse in (True, False):
        g = Graph(RandomWeights(seed=9), (h, w), 3)
        g.use_stem2 = fuse
        y = g.conv('s', g.input, 32, 3, 1, act1)
        y = g.conv('d', y, cout, 3, 2, act2)
        if c3:
            y = g.conv('p', y, c3, 1, 1, 'mish')
        g.outputs.append(y)
        assert ([d['op'] for d in g.layers] == [18]) == fuse and len(g.layers) == (1 if fuse else 2 + (c3 > 0))
        assert not fuse or len(g.layers[0]['gates']) == (3 if c3 else 1)
        net = HipNet(ctx, NET_DETECTOR, g, n)
        net.write(g.input, x)
        net.run(n)
        outs.append(net.read(y, n))
        net.close()
        if fuse:
            bufs, _ = torch_ref.run_graph(g, nchw(x[..., :3].astype(np.float32)))
            close(outs[0], nhwc(bufs[y.tid][:, :y.c]), what='fused stem pair vs torch')
    close(outs[0], outs[1], rel=4e-3, abs_=1e-3, what='fused stem pair vs the separate layers')


@pytest.mark.parametrize('cout,act1,act2,h,w,n', [(64, 'mish', 'mish', 96, 96, 1), (128, 'mish', 'leaky', 91, 101, 2),
                                                 (64, 'leaky', 'mish', 304, 304, 1), (128, 'mish', 'mish', 200, 200, 1)])
def test_pointwise_pair_fused(ctx, cout, act1, act2, h, w, n):
    """FM_OP_PAIR11 (pair11.hip): a 64 -> 64 pointwise conv into the first half of a concat + the pointwise conv over the
    concat in one launch == the two layers (bit for bit: same K order, same MFMA sequence per output element) == torch;
    pixel counts that are not a multiple of the 128-pixel tile, batch 2, the 304 x 304 map of YOLOv4's first stage."""
    rng = np.random.default_rng(cout + h)
    x = rng.normal(0, 1, (n, h, w, 64)).astype(np.float16)
    outs = []
    for fuse in (True, False):
        g = Graph(RandomWeights(seed=13), (h, w), 64)
        g.use_pair11 = fuse
        cat = g.new(h, w, 128)
        g.conv('A', g.input, 64, 1, 1, 'mish', dst=cat.slice(64, 64))
        b = g.conv('B', g.input, 64, 1, 1, 'mish')
        g.conv('t', b, 64, 1, 1, act1, dst=cat.slice(0, 64))
        y = g.conv('y', cat, cout, 1, 1, act2)
        g.outputs.append(y)
        assert (g.layers[-1]['op'] == 19) == fuse and len(g.layers) == (3 if fuse else 4)
        net = HipNet(ctx, NET_DETECTOR, g, n)
        net.write(g.input, x)
        net.run(n)
        outs.append(net.read(y, n))
        net.close()
        if fuse:
            bufs, _ = torch_ref.run_graph(g, nchw(x.astype(np.float32)))
            close(outs[0], nhwc(bufs[y.tid][:, :cout]), what='fused pointwise pair vs torch')
    np.testing.assert_array_equal(outs[0], outs[1])
