"""Profiling build only (FASTMOT_EXTRA_HIPCC_FLAGS=-DFM_LCH_TIMING python -m fastmot_amd.build --force): where
the deepest stream's workgroup of each litechain launch of OSNet-x0.25 spends its time (s_memtime cycles)."""
import os as _os
_os.environ.setdefault('FASTMOT_RANDOM_WEIGHTS', '1')
_os.environ['FASTMOT_GRAPHS'] = '0'      # layer by layer: the stamps of a launch are read right after it
import sys, ctypes as C
sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
from fastmot_amd.runtime import get_context
from fastmot_amd.engine import HipNet
from fastmot_amd.models import ReID

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 25
ctx = get_context()
ctx.feat_configure(512)
g, _ = ReID.get_model('OSNet025').build_graph()
net = HipNet(ctx, 1, g, batch, reuse_buffers=True)
for _ in range(3):
    net.run(batch)
    ctx.synchronize()
st = (C.c_longlong * 32)()
ctx.lib.fm_debug_lch_stamps(st)
s = list(st)
print('litechain launch', _os.environ.get('FASTMOT_LCH_TIMING_LAUNCH', '5'), 'of the six of the network (0, 1: 64 x 32 maps, C = 16; 2, 3: 32 x 16, C = 24; 4, 5: 16 x 8, C = 32), batch', batch, '-- deepest stream, workgroup 0; cycles')
print('wall (100 MHz ticks):', s[31] - s[30], '-> us', (s[31] - s[30]) / 100.0, ' cycles total', s[8] - s[0])
prev = s[0]
for lvl in range(4):
    a_done, a_bar, b_done, b_bar = s[16 + lvl], s[1 + 2 * lvl], s[20 + lvl], s[2 + 2 * lvl]
    print(f'level {lvl}: phase A (wave 0) {a_done - prev:7d}  barrier wait {a_bar - a_done:6d}  '
          f'phase B (thread 0) {b_done - a_bar:7d}  barrier wait {b_bar - b_done:6d}')
    prev = b_bar
