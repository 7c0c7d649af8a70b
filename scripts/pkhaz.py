"""Stand-alone reproducer of the packed-fp32 mis-execution (csrc/diag.hip): the LK position-update chain on wave-uniform
operands, idle and beside the fused-LightConv hammer (the chain kernels of OSNet-x0.25 on the ReID stream)."""
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

LAUNCHES = int(sys.argv[1]) if len(sys.argv) > 1 else 60
ctx = get_context()
ctx.feat_configure(512)
NAMES = {0: 'chain as compiled', 1: 'swapped multiply into a fresh pair', 2: '+1 wait state between the multiplies',
         3: 's_nop 3 everywhere', 4: 'unpacked v_mul / v_sub'}


def sweep(label):
    for v in range(5):
        tot = np.zeros(8, np.int64)
        for _ in range(LAUNCHES):
            tot += diag_pkhaz(ctx, v, 600, 2000)
        print(f'{label:<22} variant {v} ({NAMES[v]:<38}) wrong lanes by quarter: low half {tot[:4].tolist()}  high half {tot[4:].tolist()}'
              f'  of {LAUNCHES * 600 * 2000} evaluations', flush=True)


sweep('idle')
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
    sweep('beside litechain x6')
finally:
    stop.append(1)
    th.join()
