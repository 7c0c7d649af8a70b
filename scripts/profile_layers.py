"""Per-layer HIP-event table of a network: shape, tile-independent roofline time and achieved rate."""
import os as _os
_os.environ.setdefault('FASTMOT_RANDOM_WEIGHTS', '1')   # no weight files offline
import sys
import numpy as np
sys.path.insert(0, '.')
from fastmot_amd.runtime import get_context
from fastmot_amd.engine import HipNet
from fastmot_amd.models import YOLO, ReID

which = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ctx = get_context()
if which == 0:
    g, _ = YOLO.get_model('YOLOv4_608').build_graph(); batch = 1
else:
    ctx.feat_configure(512)
    g, _ = ReID.get_model('OSNet025').build_graph(); batch = 50
net = HipNet(ctx, which, g, batch, reuse_buffers=True)
for _ in range(3):
    net.run(batch)
ms = net.profile_layers(batch, 10)
rows = []
for d, t in zip(g.layers, ms):
    o, i = d['out'], d['ins'][0]
    P = batch * o.h * o.w
    if d['op'] in (0, 15):
        K = d['k'] * d['k'] * d['cin']
        fl = 2.0 * K * d['cout'] * P
        by = (batch * i.h * i.w * d['cin'] + P * d['cout'] + K * d['cout']) * 2
        ideal = max(fl / 2.5e15, by / 6.3e12) * 1e6
        rows.append((t * 1e3, f"conv{'S' if d['op'] == 15 else ' '} k{d['k']}s{d['stride']} P={P:<7} Cin={d['cin']:<5} Cout={d['cout']:<5} K={K:<6} "
                               f"{fl / 1e9:6.2f} GF {t * 1e3:7.1f} us {fl / t / 1e9:7.1f} TF/s ideal {ideal:5.1f} us"))
    else:
        rows.append((t * 1e3, f"op{d['op']} P={P} C={d['cin']} {t * 1e3:7.1f} us"))
print('total ms', ms.sum(), ' sum conv', sum(t for d, t in zip(g.layers, ms) if d['op'] in (0, 15)))
for i, (t, r) in enumerate(rows):
    print(f'{i:3d} {r}')
