"""A/B sweep of the conv kernels on single layers (round 5): for each layer shape, the LDS-tiled kernel (conv.hip), the
streamed kernel (convs.hip) where it applies, and convd.hip under a list of forced (tile, K groups, stages, loader
waves = 'L', two steps per barrier = 'x2') configurations.  Each measurement is a table of NREP layers of the same shape with DISTINCT weights reading one input
(so a layer's weights are not L2-resident from its previous run, as in the real network), timed two ways:
  ev  -- HIP events around every eager launch (fm_net_profile_layers), mean over layers and iterations;
  rep -- wall time of graph replays of the whole table / NREP (includes the ~1.5 us dependent-launch boundary).
Outputs are compared with the LDS-tiled kernel's (max abs difference, fp16 tensors).

    python scripts/convd_sweep.py [shape-set] > gpurun_out/convd_sweep.txt
"""
import os
os.environ.setdefault('FASTMOT_RANDOM_WEIGHTS', '1')
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fastmot_amd.runtime import get_context
from fastmot_amd.engine import HipNet, NET_DETECTOR
from fastmot_amd.models.graph import Graph, RandomWeights

NREP = 8
SHAPES = {
    # name: (cin, cout, k, stride, h, w)
    'k3_76_128_256': (128, 256, 3, 1, 76, 76),
    'k3_38_256_512': (256, 512, 3, 1, 38, 38),
    'k3_19_512_1024': (512, 1024, 3, 1, 19, 19),
    'k3_19_512_512': (512, 512, 3, 1, 19, 19),
    'k3s2_152_128_256': (128, 256, 3, 2, 152, 152),
    'k3s2_76_256_512': (256, 512, 3, 2, 76, 76),
    'k3s2_304_64_128': (64, 128, 3, 2, 304, 304),
    'k1_304_64_128': (64, 128, 1, 1, 304, 304),
    'k1_152_128_128': (128, 128, 1, 1, 152, 152),
    'k1_76_256_256': (256, 256, 1, 1, 76, 76),
    'k1_76_256_128': (256, 128, 1, 1, 76, 76),
    'k1_38_512_256': (512, 256, 1, 1, 38, 38),
    'k1_19_1024_512': (1024, 512, 1, 1, 19, 19),
    'k1_19_2048_512': (2048, 512, 1, 1, 19, 19),
    # P6 @ 1280 (config[4])
    'k3_160_128_128': (128, 128, 3, 1, 160, 160),
    'k3_80_256_256': (256, 256, 3, 1, 80, 80),
    'k3_40_512_512': (512, 512, 3, 1, 40, 40),
    'k3s2_320_64_128': (64, 128, 3, 2, 320, 320),
    'k1_160_256_256': (256, 256, 1, 1, 160, 160),
}
SETS = {
    'all': list(SHAPES),
    'quick': ['k3_76_128_256', 'k3_38_256_512', 'k3_19_512_1024', 'k1_76_256_256', 'k1_19_1024_512'],
}


def code(bm, bn, kg, ns=0, role=0, spb=1):
    return bm | bn << 8 | kg << 16 | ns << 20 | role << 24 | (spb - 1) << 25


CFGS = [('auto', 0)] + [(f'{bm}x{bn} kg{kg}' + (' L' if role else '') + (' x2' if spb == 2 else '') + (f' ns{ns}' if ns else ''),
                         code(bm, bn, kg, ns, role, spb))
                        for bm, bn in ((128, 128), (128, 64), (64, 64))
                        for kg in (1, 2, 4) if not (kg == 4 and bm * bn > 64 * 64)
                        for role in (0, 1) if not (role and kg == 4)
                        for spb in (1, 2) if not (spb == 2 and (bm * bn == 128 * 128 or kg == 4 or (bm == 128 and kg == 2)))
                        for ns in ((0, 2) if (bm, kg, spb) in ((128, 1, 1), (64, 2, 1)) else (0,))]


def measure(ctx, shape, level, maxp, cfg):
    cin, cout, k, stride, h, w = shape
    g = Graph(RandomWeights(seed=1), (h, w), cin)
    g.convd_level = level
    g.convs_max_pixels = maxp
    ys = [g.conv(f'c{i}', g.input, cout, k, stride, 'leaky') for i in range(NREP)]
    ops = {d['op'] for d in g.layers}
    ctx.set_option('convd_cfg', cfg)
    net = HipNet(ctx, NET_DETECTOR, g, 1)
    x = np.random.default_rng(0).normal(0, 1, (1, h, w, cin)).astype(np.float16)
    net.write(g.input, x)
    for _ in range(3):
        net.run(1)
    ctx.synchronize()
    out = net.read(ys[0], 1)
    ev = float(np.mean(net.profile_layers(1, 10))) * 1e3
    for _ in range(3):
        net.run(1)
    ctx.synchronize()
    t0 = time.perf_counter()
    R = 30
    for _ in range(R):
        net.run(1)
    ctx.synchronize()
    rep = (time.perf_counter() - t0) / (R * NREP) * 1e6
    net.close()
    ctx.set_option('convd_cfg', 0)
    return ev, rep, out, ops


def kscan(ctx):
    """Same map and couts, growing reduction: time = fixed + steps * slope per configuration (least squares)."""
    for (k, stride, h, cout) in ((3, 1, 76, 256), (3, 1, 38, 512), (1, 1, 76, 256), (3, 1, 160, 128)):
        print(f'## K scan: k{k}s{stride} {h}x{h}, cout {cout}', flush=True)
        for label, cfg in [('tiled', None)] + [(f'{bm}x{bn} kg{kg} L', code(bm, bn, kg, 0, 1)) for bm, bn, kg in
                                               ((128, 128, 1), (128, 64, 1), (64, 64, 1), (128, 64, 2), (64, 64, 2))]:
            xs, ys = [], []
            for cin in (64, 128, 256, 512):
                ev, rep, out, ops = measure(ctx, (cin, cout, k, stride, h, h), 0 if cfg is None else 2, 0, cfg or 0)
                xs.append(k * k * cin // 64)
                ys.append(ev)
            slope, fixed = np.polyfit(xs, ys, 1)
            print(f'   {label:<14} ' + '  '.join(f'{x} steps: {y:6.2f} us' for x, y in zip(xs, ys)) +
                  f'   -> fixed {fixed:5.2f} us + {slope * 1e3:6.1f} ns / step', flush=True)


def main():
    ctx = get_context()
    if len(sys.argv) > 1 and sys.argv[1] == 'kscan':
        return kscan(ctx)
    names = SETS.get(sys.argv[1] if len(sys.argv) > 1 else 'all') or sys.argv[1].split(',')
    for name in names:
        shape = SHAPES[name]
        cin, cout, k, stride, h, w = shape
        ho = (h + 2 * (k // 2) - k) // stride + 1
        gf = 2.0 * k * k * cin * cout * ho * ho / 1e9
        print(f'## {name}: k{k}s{stride} {h}x{w}x{cin} -> {ho}x{ho}x{cout}  {gf:.2f} GFLOP  (MFMA floor {gf / 2.5:.2f} us)', flush=True)
        ev, rep, ref, ops = measure(ctx, shape, 0, 0, 0)
        print(f'   {"tiled (conv.hip)":<22} ev {ev:7.2f} us  rep {rep:7.2f} us  {gf / ev * 1e3:7.1f} TFLOP/s', flush=True)
        if cin % 64 == 0 and k * k * cin >= 512 and ho * ho <= 1444:
            ev, rep, out, ops = measure(ctx, shape, 0, 10 ** 6, 0)
            d = float(np.abs(out.astype(np.float32) - ref.astype(np.float32)).max())
            print(f'   {"streamed (convs.hip)":<22} ev {ev:7.2f} us  rep {rep:7.2f} us  {gf / ev * 1e3:7.1f} TFLOP/s  maxdiff {d:.3g}', flush=True)
        seen = set()
        for label, cfg in CFGS:
            try:
                ev, rep, out, ops = measure(ctx, shape, 2, 0, cfg)
            except Exception as e:                                   # a configuration the launcher refuses
                print(f'   convd {label:<16} failed: {str(e)[:100]}', flush=True)
                continue
            d = float(np.abs(out.astype(np.float32) - ref.astype(np.float32)).max())
            print(f'   convd {label:<16} ev {ev:7.2f} us  rep {rep:7.2f} us  {gf / ev * 1e3:7.1f} TFLOP/s  maxdiff {d:.3g}'
                  f'  (max |ref| {float(np.abs(ref).max()):.3g})', flush=True)


if __name__ == '__main__':
    main()
