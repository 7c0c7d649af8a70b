"""Profiling build only (FASTMOT_EXTRA_HIPCC_FLAGS=-DFM_CONVS_TIMING): cycle stamps of workgroup 0 / wave 0 of every
streamed-conv launch of YOLOv4@608 (layer by layer, no graph)."""
import os as _os
_os.environ.setdefault('FASTMOT_RANDOM_WEIGHTS', '1')
_os.environ['FASTMOT_GRAPHS'] = '0'
import sys, ctypes as C
sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
from fastmot_amd.runtime import get_context
from fastmot_amd.engine import HipNet
from fastmot_amd.models import YOLO
from fastmot_amd.models import graph as G
from fastmot_amd import _lib

ctx = get_context()
g, _ = YOLO.get_model('YOLOv4_608').build_graph()
net = HipNet(ctx, 0, g, 1, reuse_buffers=True)
for _ in range(3):
    net.run(1)
    ctx.synchronize()
# run layer by layer is not exposed: the stamps of the LAST streamed launch of the network are read instead, plus a few
# chosen layers re-run through one-layer graphs would need the builder; print what the last launch left
st = (C.c_longlong * 64)()
ctx.lib.fm_debug_convs_stamps(st)
s_all = list(st)
for off, tag in ((0, 'workgroup 0 (first on its CU)'), (32, 'workgroup 320 (a later round on a CU that has run the kernel)')):
    s = s_all[off:off + 32]
    if s[5] <= s[0]:
        print(tag, ': no stamps')
        continue
    print(tag, ': CT/NW/PT code', s[7], 'chunks per wave', s[6])
    print('  prologue %d  first loads issued %d  loop %d  wait for the other waves %d  reduce+epilogue %d  total %d cycles, started %d cycles after workgroup 0' % (
        s[1] - s[0], s[2] - s[1], s[3] - s[2], s[4] - s[3], s[5] - s[4], s[5] - s[0], s[0] - s_all[0]))
