"""The fused residual unit (csrc/resblock.hip) per map size of YOLOv4 @ 608 under another tile / wave split
(FASTMOT_RB_VARIANT, read once per process): NREP units with distinct weights in a row, HIP events around every eager launch.

    for v in 0 1 2 3; do FASTMOT_RB_VARIANT=$v python scripts/resblock_sweep.py; done"""
import os
os.environ.setdefault('FASTMOT_RANDOM_WEIGHTS', '1')
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fastmot_amd.runtime import get_context
from fastmot_amd.engine import HipNet, NET_DETECTOR
from fastmot_amd.models.graph import Graph, RandomWeights

NREP = 8
SHAPES = [(64, 32, 304), (64, 64, 152), (128, 128, 76), (256, 256, 38)]       # (channels, mid, map)


def main():
    ctx = get_context()
    v = os.environ.get('FASTMOT_RB_VARIANT', '0')
    out = []
    for c, m, h in SHAPES:
        g = Graph(RandomWeights(seed=1), (h, h), c)
        x = g.input
        for i in range(NREP):
            x = g.resblock(f'r{i}.1', f'r{i}.2', x, m, 'mish')
        assert all(d['op'] == 14 for d in g.layers), [d['op'] for d in g.layers]
        net = HipNet(ctx, NET_DETECTOR, g, 1)
        net.write(g.input, np.random.default_rng(0).normal(0, 1, (1, h, h, c)).astype(np.float16))
        for _ in range(3):
            net.run(1)
        ctx.synchronize()
        t = np.asarray(net.profile_layers(1, 20)) * 1e3
        net.close()
        out.append(f'{h}^2 x {c} (mid {m}): {np.mean(t[1:]):6.2f} us')
    print(f'variant {v}: ' + ' | '.join(out), flush=True)


if __name__ == '__main__':
    main()
