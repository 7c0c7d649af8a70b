"""Profiling build only (bash scripts/build_timing_lib.sh ostail.hip -DFM_OST_TIMING; FASTMOT_LIB_PATH=fastmot_amd/libfastmot_hip_timing.so):
where workgroup 0 of the fused OSNet tail (csrc/ostail.hip) spends its time -- wall-clock stamps (100 MHz) between its phases."""
import os as _os
_os.environ.setdefault('FASTMOT_RANDOM_WEIGHTS', '1')
_os.environ['FASTMOT_GRAPHS'] = '0'
import sys, ctypes as C
sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
from fastmot_amd.runtime import get_context
from fastmot_amd.engine import HipNet
from fastmot_amd.models import ReID

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 50
ctx = get_context()
ctx.feat_configure(512)
g, _ = ReID.get_model('OSNet025').build_graph()
assert g.layers[-1]['op'] == 20
net = HipNet(ctx, 1, g, batch, reuse_buffers=True)
for _ in range(3):
    net.run(batch)
    ctx.synchronize()
st = (C.c_longlong * 48)()
ctx.lib.fm_debug_ost_stamps(st)
s = list(st)
us = lambda a, b: (s[b] - s[a]) / 100.0
print(f'fused OSNet tail, batch {batch}, workgroup 0 (thread 0): total {us(0, 11):.2f} us')
names = ['pool + parameter staging', 'conv1 A', 'chain A', 'gate A', 'conv3 + downsample', 'conv1 B', 'chain B', 'gate B', 'conv3 B',
         'conv5', 'head']
for i, nm in enumerate(names):
    print(f'  {nm:<26} {us(i, i + 1):6.2f} us')
for blk, base, start in (('A', 16, 2), ('B', 24, 6)):
    prev = s[start]
    for lvl in range(4):
        a, b = s[base + 2 * lvl], s[base + 2 * lvl + 1]
        print(f'  chain {blk} level {lvl}: pointwise {(a - prev) / 100.0:5.2f} us  depthwise {(b - a) / 100.0:5.2f} us')
        prev = b
