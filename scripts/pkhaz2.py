"""Victim x neighbour matrix of the packed-fp32 mis-execution (csrc/diag.hip fm_diag_pkhaz2)."""
import os as _os
_os.environ.setdefault('FASTMOT_RANDOM_WEIGHTS', '1')
import sys, threading
sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import numpy as np
from fastmot_amd.runtime import get_context
sys.path.insert(1, _os.path.dirname(_os.path.abspath(__file__)))
from diag_bindings import flow_lk_diag, diag_pkhaz, diag_pkhaz2   # needs a -DFM_DIAG build
from fastmot_amd.engine import HipNet, NET_EXTRACTOR
from fastmot_amd.models import ReID

L = int(sys.argv[1]) if len(sys.argv) > 1 else 24
ctx = get_context()
ctx.feat_configure(512)
VIC = ['v_pk_mul_f32', 'v_pk_add_f32', 'v_pk_fma_f32', 'v_pk_mul_f32 op_sel swap', 'v_pk_mul_f32 neg', 'pk_mul -> pk_add pair']
AGG = {-1: 'none', 0: 'v_fma_mix_f32 op_sel:[1,0,0]', 1: 'v_fma_mix_f32 op_sel:[0,0,0]', 2: 'v_cvt_f32_f16_sdwa', 3: 'v_pk_fma_f16',
       4: 'v_pk_fma_f32 / v_pk_mul_f32', 5: 'v_fma_f32', 6: 'v_mul_f32 / v_add_f32'}
print(f'{L} launches x 600 wavefronts x 2000 evaluations per cell; cell = wrong lanes [q0, q1, q2, q3] low half / high half')
for a, an in AGG.items():
    for v, vn in enumerate(VIC):
        o = diag_pkhaz2(ctx, v, a, L)
        print(f'neighbour {an:<30} victim {vn:<26} low {o[:4].tolist()} high {o[4:].tolist()}', flush=True)
g, _ = ReID.get_model('OSNet025').build_graph()
g.layers[:] = [d for d in g.layers if d['op'] == 16]
net = HipNet(ctx, NET_EXTRACTOR, g, 50, reuse_buffers=False)
net.run(50)
ctx.synchronize()
stop = []


def hammer():
    ctx.bind_thread()
    while not stop:
        net.run(50)
        ctx.synchronize()


th = threading.Thread(target=hammer)
th.start()
try:
    for v, vn in enumerate(VIC):
        o = diag_pkhaz2(ctx, v, -1, L)
        print(f'neighbour {"litechain_kernel x6 (OSNet)":<30} victim {vn:<26} low {o[:4].tolist()} high {o[4:].tolist()}', flush=True)
finally:
    stop.append(1)
    th.join()
