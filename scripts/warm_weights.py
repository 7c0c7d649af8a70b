"""How much of a small layer's time is the first trip to its weights?  Per layer shape of YOLOv4 @ 608: NREP layers of
that shape reading one input, with DISTINCT weights (every launch fetches its weights from HBM / the Infinity Cache, as
in the network) and with the SAME weights (L2-resident from the previous launch) -- the difference is what a perfect
prefetch of the next layer's weights into L2 could buy.  HIP events around every eager launch, the builder's own
kernel choice per layer.

    python scripts/warm_weights.py > gpurun_out/warm_weights.txt"""
import os
os.environ.setdefault('FASTMOT_RANDOM_WEIGHTS', '1')
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fastmot_amd.runtime import get_context
from fastmot_amd.engine import HipNet, NET_DETECTOR
from fastmot_amd.models.graph import Graph, RandomWeights

NREP = 8
SHAPES = [  # (cin, cout, k, stride, h)
    (512, 512, 1, 1, 19), (1024, 512, 1, 1, 19), (2048, 512, 1, 1, 19), (1024, 1024, 1, 1, 19), (512, 1024, 3, 1, 19),
    (512, 512, 3, 1, 19), (512, 256, 1, 1, 38), (512, 512, 1, 1, 38), (256, 512, 3, 1, 38), (256, 128, 1, 1, 76),
    (256, 256, 1, 1, 76), (128, 256, 3, 1, 76), (128, 128, 1, 1, 152), (64, 64, 1, 1, 304)]


def measure(ctx, shape, same):
    cin, cout, k, stride, h = shape
    g = Graph(RandomWeights(seed=1), (h, h), cin)
    for i in range(NREP):
        g.conv(f'c{i}', g.input, cout, k, stride, 'leaky')
    if same:
        for d in g.layers[1:]:
            d['w_off'], d['b_off'] = g.layers[0]['w_off'], g.layers[0]['b_off']
    net = HipNet(ctx, NET_DETECTOR, g, 1)
    net.write(g.input, np.random.default_rng(0).normal(0, 1, (1, h, h, cin)).astype(np.float16))
    for _ in range(3):
        net.run(1)
    ctx.synchronize()
    t = np.asarray(net.profile_layers(1, 10)) * 1e3          # us per layer (mean over iterations)
    net.close()
    return float(np.mean(t[1:])), g.layers[0]['op']


def main():
    ctx = get_context()
    names = {0: 'tiled', 15: 'streamed', 17: 'convd'}
    tot_c = tot_w = 0.
    for shape in SHAPES:
        cold, op = measure(ctx, shape, False)
        warm, _ = measure(ctx, shape, True)
        cin, cout, k, stride, h = shape
        mb = cout * k * k * cin * 2 / 1e6
        print(f'k{k}s{stride} {h}x{h}x{cin} -> {cout} ({names.get(op, op)}, {mb:5.2f} MB of weights): distinct weights {cold:6.2f} us, '
              f'same weights {warm:6.2f} us  ({cold - warm:+.2f})', flush=True)


if __name__ == '__main__':
    main()
