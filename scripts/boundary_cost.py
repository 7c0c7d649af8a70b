"""What does one more DEPENDENT launch cost on this part?  Chains of N trivial 1x1 conv layers (1 workgroup each,
or 256 workgroups each) replayed as one hipGraph: the slope of wall time over N is kernel + boundary, the HIP-event
per-layer times give the kernel itself.  MI355X_MICROARCH.md quotes 1.2-1.9 us per same-stream boundary."""
import os as _os
_os.environ.setdefault('FASTMOT_RANDOM_WEIGHTS', '1')
import sys, time
import numpy as np
sys.path.insert(0, '.')
from fastmot_amd.runtime import get_context
from fastmot_amd.engine import HipNet, NET_DETECTOR
from fastmot_amd.models.graph import Graph, RandomWeights

ctx = get_context()
for label, hw, c in (('1 workgroup (8x8x32)', (8, 8), 32), ('256 workgroups (128x128x32)', (128, 128), 32),
                     ('1024 workgroups (256x256x64)', (256, 256), 64)):
    res = {}
    for n in (10, 40, 160):
        g = Graph(RandomWeights(seed=1), hw, c)
        x = g.input
        for i in range(n):
            x = g.conv(f'c{i}', x, c, 1, 1, 'leaky')
        net = HipNet(ctx, NET_DETECTOR, g, 1, reuse_buffers=True)
        for _ in range(5):
            net.run(1)
        ctx.synchronize()
        reps = 50
        t0 = time.perf_counter()
        for _ in range(reps):
            net.run(1)
        ctx.synchronize()
        wall = (time.perf_counter() - t0) / reps * 1e6
        ev = net.profile_layers(1, 5)
        res[n] = (wall, float(np.mean(ev)) * 1e3)
        net.close()
    slope = (res[160][0] - res[40][0]) / 120
    print(f'{label}: graph replay wall us for N=10/40/160 layers: ' + ' / '.join(f'{res[n][0]:.1f}' for n in (10, 40, 160)) +
          f'  -> {slope:.2f} us per dependent launch; kernel alone (HIP events) {res[160][1]:.2f} us '
          f'-> boundary ~{slope - res[160][1]:.2f} us')
