"""Per-layer roofline table of the detector: for every launch of one hipGraph replay (durations from a rocprofv3
--kernel-trace database of scripts/trace_net.py) the algorithmic FLOPs and bytes of the layer(s) it computes,
t_roof = max(FLOP / 2.5 PFLOP/s, bytes / 8 TB/s) and the measured duration.  Sum(t_roof) / sum(measured) is the
"fraction of the mixed conv roofline" of SURVEY.md section 8d.

    rocprofv3 --kernel-trace -d /tmp/tr -o t -- python scripts/trace_net.py 0
    python scripts/layer_roofline.py <db> [YOLOv4_608]   -> profiles/r02_yolo_layer_roofline.txt"""
import os as _os
_os.environ.setdefault('FASTMOT_RANDOM_WEIGHTS', '1')
import sqlite3
import sys
sys.path.insert(0, '.')
from fastmot_amd.models import YOLO

P_PEAK, BW = 2.5e15, 8.0e12
import glob
import os


def open_trace(path):
    """path: a rocpd database or a rocprofv3 output directory (the database with kernel dispatches is picked)."""
    cands = [path] if os.path.isfile(path) else sorted(glob.glob(os.path.join(path, '**', '*.db'), recursive=True),
                                                       key=os.path.getsize, reverse=True)
    for c in cands:
        d = sqlite3.connect(c)
        try:
            d.execute('select count(*) from kernels').fetchone()
            return d
        except sqlite3.OperationalError:
            continue
    raise SystemExit(f'no rocpd database with a kernels table under {path}: {cands}')


db = open_trace(sys.argv[1])
model = sys.argv[2] if len(sys.argv) > 2 else 'YOLOv4_608'
g, _ = YOLO.get_model(model).build_graph()
n = len(g.layers)
rows = []
for s_, e_, name_ in db.execute("select start, end, name from kernels order by start").fetchall():
    if 'splitk_reduce' in name_ and rows:      # the reduce launch of a split-K layer belongs to its conv's row (same layer)
        ps, pe, pn = rows[-1]
        rows[-1] = (ps, pe + (e_ - s_), pn)
    else:
        rows.append((s_, e_, name_))
rows = rows[-n:]
assert len(rows) == n
print(f'# {model}: {n} launches per frame (last graph replay of the trace); peaks: 2.5 PFLOP/s dense fp16 MFMA, 8 TB/s HBM3E')
print(f'{"#":>3} {"op":<10} {"shape":<44} {"GFLOP":>7} {"MB":>7} {"roof_us":>8} {"meas_us":>8} {"TFLOP/s":>8} {"GB/s":>7} {"bound":>5}')
tot_fl = tot_by = tot_roof = tot_meas = 0.
for i, (d, (s, e, name)) in enumerate(zip(g.layers, rows)):
    o, x = d['out'], d['ins'][0]
    P = o.h * o.w
    op = d['op']
    if op in (0, 12, 15, 17):                  # conv (LDS-tiled / stem / streamed / DMA-fed)
        K = d['k'] * d['k'] * d['cin']
        up = d.get('up') or 1
        Pc = P // (up * up)
        fl = 2.0 * K * d['cout'] * Pc
        by = (x.h * x.w * d['cin'] + P * d['cout'] * (2 if g.tensors[o.tid][3] else 1) + K * d['cout']) * 2
        shape = f"k{d['k']}s{d['stride']} {x.h}x{x.w}x{d['cin']} -> {o.h}x{o.w}x{d['cout']}"
        kind = {0: 'conv', 12: 'stemconv', 15: 'convS', 17: 'convD'}[op]
    elif op == 19:                             # fused pointwise pair: counted as the two layers it replaces
        m, c2 = d['hid'], d['cin']
        fl = 2.0 * P * (d['cin'] * m + (m + c2) * d['cout'])
        by = (P * d['cin'] + P * m + d['cin'] * m) * 2 + (P * (m + c2) + P * d['cout'] + (m + c2) * d['cout']) * 2
        shape = f"k1+k1 {o.h}x{o.w}x{d['cin']}(+{c2}) -> {d['cout']}"
        kind = 'pair11'
    elif op == 18:                             # fused stem pair: conv 3x3 s1 (cin -> 32) + conv 3x3 s2 (32 -> cout), counted as the
        m, Pm = d['hid'], x.h * x.w            # two layers it replaces (the table's roofline is the network's, not the fusion's)
        c2 = d['gates'][1] if len(d['gates']) > 1 else d['cout']
        fl = 2.0 * 9 * d['cin'] * m * Pm + 2.0 * 9 * m * c2 * P
        by = (Pm * d['cin'] + Pm * m + 9 * d['cin'] * m) * 2 + (Pm * m + P * c2 + 9 * m * c2) * 2
        shape = f"k3s1+k3s2 {x.h}x{x.w}x{d['cin']} -> {o.h}x{o.w}x{c2}"
        if len(d['gates']) > 1:                # ... + the pointwise conv behind them
            fl += 2.0 * c2 * d['cout'] * P
            by += (P * c2 + P * d['cout'] + c2 * d['cout']) * 2
            shape = f"k3s1+k3s2+k1 {x.h}x{x.w}x{d['cin']} -> {o.h}x{o.w}x{d['cout']}"
        kind = 'stem2'
    elif op == 14:                             # fused residual unit: 1x1 (c -> m) + 3x3 (m -> c) + shortcut
        w1, _, w2, _ = d['res_ref']
        m, c = w1.shape[0], w1.shape[1]
        fl = 2.0 * P * (c * m + 9 * m * c)
        by = (2 * P * c + c * m + 9 * m * c) * 2
        shape = f"{o.h}x{o.w}x{c} (mid {m})"
        kind = 'resblock'
    else:                                      # SPP etc.
        fl = 0.
        by = (x.h * x.w * x.c + P * o.c) * 2
        shape = f"{x.h}x{x.w}x{x.c} -> {o.h}x{o.w}x{o.c}"
        kind = f'op{op}'
    roof = max(fl / P_PEAK, by / BW) * 1e6
    meas = (e - s) / 1e3
    tot_fl += fl; tot_by += by; tot_roof += roof; tot_meas += meas
    print(f'{i:>3} {kind:<10} {shape:<44} {fl / 1e9:>7.2f} {by / 1e6:>7.2f} {roof:>8.2f} {meas:>8.2f} {fl / meas / 1e6:>8.1f} '
          f'{by / meas / 1e3:>7.0f} {"mfma" if fl / P_PEAK > by / BW else "hbm":>5}')
span = (rows[-1][1] - rows[0][0]) / 1e3
print(f'\ntotal: {tot_fl / 1e9:.1f} GFLOP, {tot_by / 1e6:.0f} MB algorithmic; sum of per-layer rooflines {tot_roof:.1f} us; '
      f'sum of kernel durations {tot_meas:.1f} us; replay span {span:.1f} us')
print(f'fraction of the mixed roofline: {tot_roof / span:.4f} (span) / {tot_roof / tot_meas:.4f} (kernel time); '
      f'{tot_fl / span / 1e6:.1f} TFLOP/s = {tot_fl / span / 1e6 / 2500:.4f} of the dense MFMA peak over the replay span')
