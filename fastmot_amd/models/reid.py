"""ReID model descriptors (registry + attributes of fastmot/models/reid.py:10-45,95-109) and the
OSNet layer table for the HIP conv engine (torchreid osnet_x1_0 / osnet_x0_25 topology,
SURVEY.md appendix A: 978.9 / 82.3 MMAC per 256x128 crop)."""
from pathlib import Path

import numpy as np

from .graph import Graph, RES_BEFORE_ACT, fold_bn, missing_weights


class ReID:
    """Base class for ReID models.

    ENGINE_PATH / MODEL_PATH : cache / checkpoint locations; INPUT_SHAPE (c, h, w);
    OUTPUT_LAYOUT : feature dimension; METRIC : {'euclidean', 'cosine'}.
    """
    __registry = {}

    PLUGIN_PATH = None
    ENGINE_PATH = None
    MODEL_PATH = None
    INPUT_SHAPE = None
    OUTPUT_LAYOUT = None
    METRIC = None
    CHANNELS = None

    def __init_subclass__(cls, **kwargs):
        super().__init_subclass__(**kwargs)
        cls.__registry[cls.__name__] = cls

    @classmethod
    def get_model(cls, name):
        return cls.__registry[name]

    @classmethod
    def weight_file(cls):
        """MODEL_PATH if it exists, else the same stem as a torchreid checkpoint (.pth, .pth.tar); None."""
        if cls.MODEL_PATH is None:
            return None
        base = Path(cls.MODEL_PATH)
        for cand in (base, base.with_suffix('.pth'), base.with_suffix('.pth.tar')):
            if cand.is_file():
                return cand
        return None

    @classmethod
    def build_graph(cls, weights=None, fuse_lightconv=True):
        """Weights: the explicit source, else the model file at MODEL_PATH -- the reference's ONNX export
        (models/onnx_reader.py; fastmot/models/reid.py:97,106) or, next to it with the suffix .pth / .pth.tar, the
        torchreid checkpoint it was exported from (models/torchreid_weights.py); a missing file raises
        FileNotFoundError unless seeded random parameters were opted into (models.allow_random_weights())."""
        ckpt = None
        if weights is None:
            path = cls.weight_file()
            if path is not None:
                from .torchreid_weights import TorchreidWeights
                if path.suffix == '.onnx':
                    from .onnx_reader import torchreid_state_dict_from_onnx
                    weights = ckpt = TorchreidWeights(torchreid_state_dict_from_onnx(path, cls.CHANNELS,
                                                                                     cls.OUTPUT_LAYOUT))
                else:
                    weights = ckpt = TorchreidWeights(path)
            else:
                weights = missing_weights(cls, seed=1)
        out = osnet_graph(cls, weights, fuse_lightconv)
        if ckpt is not None and ckpt.unused():
            raise ValueError(f'{cls.MODEL_PATH}: parameters not used by {cls.__name__}: {ckpt.unused()[:5]} ...')
        return out


def osnet_graph(model, weights, fuse_lightconv=True):
    _, H, W = model.INPUT_SHAPE
    c0, c1, c2, c3 = model.CHANNELS
    g = Graph(weights, (H, W), 3)

    def osblock(name, x, cout, cat=None):
        """cat: [mid + x.c]-channel tensor whose upper slice already is x (the producer wrote it there):
        the gated sum goes to the lower slice and the block tail relu(conv3(x2) + downsample(x)) becomes
        ONE 1x1 conv over the concatenated channels with weights [W3 | Wd], bias b3 + bd."""
        mid = cout // 4
        x1 = g.conv(name + '.conv1', x, mid, 1, 1, 'relu')
        hid = max(mid // 16, 1)
        if fuse_lightconv and mid % 8 == 0 and mid <= 128:
            # stream t (1..4) is a chain of t LightConv3x3; the chains are independent: all of them run as
            # ONE launch when two halo tiles fit in LDS (litechain.hip), otherwise depth i of all streams that
            # reach it as one grouped launch (4, 3, 2, 1 groups); the four gates + the gated sum are one more.
            params = {(t, i): g.lightconv_params(f'{name}.s{t}.{i}', mid) for t in range(1, 5) for i in range(t)}
            streams, parts, prev = [], [], None
            chain = g.use_lightchain and g.lightchain_fits(mid, x1.h, x1.w)
            if chain:                    # all four chains in one launch: 2 launches per block tail
                y = g.lightchain(f'{name}.streams', x1, [params[(t, i)] for t in range(1, 5) for i in range(t)], 'relu')
                streams = [y.slice(t * mid, mid) for t in range(4)]
                parts = g.last_gap_slots
            for i in range(0 if chain else 4):
                ts = list(range(i + 1, 5))                       # streams alive at depth i
                xs = [x1] * len(ts) if i == 0 else [prev.slice((t - i) * mid, mid) for t in ts]
                prev = g.lightconv_group(f'{name}.depth{i}', xs, [params[(t, i)] for t in ts], 'relu', gap_slot=True)
                streams.append(prev.slice(0, mid))               # stream i+1 ends at depth i (group 0)
                parts.append(g.last_gap_slot)
            x2 = g.gated_sum(name + '.gate', streams, hid, parts=parts,
                             dst=cat.slice(0, mid) if cat is not None else None)
        else:
            streams, gids = [], []
            gp = None
            for t in range(1, 5):
                s = x1
                for i in range(t):
                    s = g.lightconv(f'{name}.s{t}.{i}', s, mid, 'relu', fuse=False)
                streams.append(s)
            for s in streams:
                gid, gp = g.gate(name + '.gate', s, hid, gp)
                gids.append(gid)
            x2 = g.gate_sum(streams, gids)
        if cat is not None:
            wd, bd = fold_bn(weights.conv(name + '.down', cout, x.c, 1, bn=True))
            w3, b3 = fold_bn(weights.conv(name + '.conv3', cout, mid, 1, bn=True))
            return g.conv(name + '.conv3+down', cat, cout, 1, 1, 'relu', wb=(np.concatenate([w3, wd], axis=1), b3 + bd))
        if x.c != cout:
            ident = g.conv(name + '.down', x, cout, 1, 1, 'linear')
        else:
            ident = x
        return g.conv(name + '.conv3', x2, cout, 1, 1, 'relu', res=ident, res_mode=RES_BEFORE_ACT)

    def down_block(name, make_x, cin, cout, h, w):
        """First block of a stage (cin != cout).  make_x(dst) emits the layer producing the block input."""
        mid = cout // 4
        if fuse_lightconv and mid % 8 == 0 and mid <= 128:
            cat = g.new(h, w, mid + cin)
            return osblock(name, make_x(cat.slice(mid, cin)), cout, cat=cat)
        return osblock(name, make_x(None), cout)

    s = g.conv('conv1', g.input, c0, 7, 2, 'relu', pad=3)
    x = down_block('conv2.0', lambda dst: g.pool(s, 3, 2, 1, dst=dst), c0, c1, H // 4, W // 4)
    x = osblock('conv2.1', x, c1)
    t2 = g.conv('conv2.t', x, c1, 1, 1, 'relu')
    x = down_block('conv3.0', lambda dst: g.pool(t2, 2, 2, 0, avg=True, dst=dst), c1, c2, H // 8, W // 8)
    x = osblock('conv3.1', x, c2)
    t3 = g.conv('conv3.t', x, c2, 1, 1, 'relu')
    tail = len(g.layers)
    x = down_block('conv4.0', lambda dst: g.pool(t3, 2, 2, 0, avg=True, dst=dst), c2, c3, H // 16, W // 16)
    x = osblock('conv4.1', x, c3)
    x = g.conv('conv5', x, c3, 1, 1, 'relu')
    g.head('fc', x, model.OUTPUT_LAYOUT)
    g.outputs = [x]
    g.fuse_ostail(tail)       # x0.25 widths: everything after the last transition conv as one launch (ostail.hip)
    return g, x


class OSNet025(ReID):
    ENGINE_PATH = Path(__file__).parent / 'osnet_x0_25_msmt17.hipnet'
    MODEL_PATH = Path(__file__).parent / 'osnet_x0_25_msmt17.onnx'
    INPUT_SHAPE = (3, 256, 128)
    OUTPUT_LAYOUT = 512
    METRIC = 'euclidean'
    CHANNELS = (16, 64, 96, 128)


class OSNet10(ReID):
    """Multi-source model trained on MSMT17, DukeMTMC, and CUHK03, not provided."""
    ENGINE_PATH = Path(__file__).parent / 'osnet_x1_0_msdc.hipnet'
    MODEL_PATH = Path(__file__).parent / 'osnet_x1_0_msdc.onnx'
    INPUT_SHAPE = (3, 256, 128)
    OUTPUT_LAYOUT = 512
    METRIC = 'cosine'
    CHANNELS = (64, 256, 384, 512)
