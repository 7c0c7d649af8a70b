"""Host vs device placement of the assignment solver (fm_lap: lap_host.hip vs lap64_kernel / lap_kernel) over the
matrix sizes of BASELINE configs [1] (50 x 50) and [4] (300 x 300): wall time per solve incl. the transfers each
placement needs.  -> profiles/r02_lap_crossover.txt"""
import os as _os
_os.environ.setdefault('FASTMOT_RANDOM_WEIGHTS', '1')
import sys, time
import numpy as np
sys.path.insert(0, '.')
sys.path.insert(0, 'oracle')
from fastmot_amd.runtime import get_context
import np_oracle as o

ctx = get_context()
rng = np.random.default_rng(0)
print('# fm_lap wall time per solve (us), 30 % of the entries gated to 1e5; results checked against scipy')
print(f'{"n x n":>10} {"host":>10} {"device":>10} {"scipy":>10}')
for n in (8, 16, 32, 50, 64, 100, 128, 200, 300, 400):
    cost = rng.uniform(0, 1, (n, n))
    cost[rng.random(cost.shape) < 0.3] = 1e5
    er, ec = o.lsa(cost)
    res = {}
    for label, elems in (('host', 1 << 30), ('device', 0)):
        ctx.set_option('host_lap_elems', elems)
        r, c = ctx.lap(cost)
        assert (r == er).all() and (c == ec).all(), (label, n)
        reps = 200 if n <= 100 else 50
        for _ in range(5):
            ctx.lap(cost)
        t = time.perf_counter()
        for _ in range(reps):
            ctx.lap(cost)
        res[label] = (time.perf_counter() - t) / reps * 1e6
    t = time.perf_counter()
    for _ in range(20):
        o.lsa(cost)
    res['scipy'] = (time.perf_counter() - t) / 20 * 1e6
    print(f'{n:>4} x {n:<4} {res["host"]:>10.1f} {res["device"]:>10.1f} {res["scipy"]:>10.1f}')
ctx.set_option('host_lap_elems', 262144)
