"""Per-layer counter view of the detector: MFMA utilisation and HBM-side read amplification of every
launch of one network replay, from separate rocprofv3 --pmc passes over scripts/trace_net.py (collect_pmc.sh):

    python scripts/layer_pmc.py <model> <sq_db> [<fetch_db> [<write_db>]]   -> table + one JSON line ('JSON {...}')

  * MFMA utilisation of a launch = SQ_VALU_MFMA_BUSY_CYCLES / (duration x 2.4 GHz x 1024 SIMDs): the counter ticks one
    cycle per SIMD cycle its matrix pipe is busy (MI355X_MICROARCH.md: 32 per v_mfma_f32_32x32x16); the expected value
    32 x FLOP / 32768 is printed next to it (a check of the unit: they agree when every MFMA is a useful one).
    The duration is the one of the PMC pass itself (profiled passes clock 2-5 % lower), the clock the chip's maximum:
    a lower bound of the utilisation.
  * read amplification = 2 x FETCH_SIZE (KB; gfx950 tallies 128-byte requests of 16 B/lane loads at 64 B) / the launch's
    algorithmic read bytes (input view + weights + shortcut); written bytes: WRITE_SIZE / algorithmic output bytes.
Launches are matched to the layer table in dispatch order (the last replay of the trace)."""
import os
os.environ.setdefault('FASTMOT_RANDOM_WEIGHTS', '1')
import glob
import json
import sqlite3
import sys
sys.path.insert(0, '.')
from fastmot_amd.models import YOLO

CLK, SIMDS = 2.4e9, 1024


def open_db(path):
    cands = [path] if os.path.isfile(path) else sorted(glob.glob(os.path.join(path, '**', '*.db'), recursive=True),
                                                       key=os.path.getsize, reverse=True)
    for c in cands:
        d = sqlite3.connect(c)
        try:
            d.execute('select count(*) from pmc_events').fetchone()
            return d
        except sqlite3.OperationalError:
            continue
    raise SystemExit(f'no rocpd database with pmc_events under {path}')


def last_replay(db, n):
    """-> list (dispatch order) of dicts: name, duration ns, counters {name: value}, for the last n dispatches."""
    rows = db.execute('select dispatch_id, start, end, name, counter_name, counter_value from pmc_events order by start, dispatch_id').fetchall()
    disp = {}
    order = []
    for did, s, e, name, cn, cv in rows:
        if did not in disp:
            disp[did] = dict(name=name, dur=e - s, c={})
            order.append(did)
        disp[did]['c'][cn] = disp[did]['c'].get(cn, 0.0) + cv
    return [disp[d] for d in order[-n:]]


def layer_cost(g, d):
    """-> (flop on the matrix cores, algorithmic read bytes, algorithmic written bytes, label)"""
    o, x = d['out'], d['ins'][0]
    P = o.h * o.w
    op = d['op']
    if op in (0, 12, 15, 17):
        K = d['k'] * d['k'] * d['cin']
        up = d.get('up') or 1
        fl = 2.0 * K * d['cout'] * (P // (up * up))
        rd = (x.h * x.w * d['cin'] + K * d['cout'] + (P * d['cout'] if d.get('res') is not None else 0)) * 2
        wr = P * d['cout'] * (4 if g.tensors[o.tid][3] else 2)
        return fl, rd, wr, f"{ {0: 'conv', 12: 'stem', 15: 'convS', 17: 'convD'}[op]} k{d['k']}s{d['stride']} {x.h}x{x.w}x{d['cin']}->{o.h}x{o.w}x{d['cout']}"
    if op == 19:          # fused pointwise pair (pair11.hip): reads = both inputs + both weight sets
        m, c2 = d['hid'], d['cin']
        return 2.0 * P * (d['cin'] * m + (m + c2) * d['cout']), (P * (d['cin'] + c2) + d['cin'] * m + (m + c2) * d['cout']) * 2, P * d['cout'] * 2, \
            f"pair11 k1+k1 {o.h}x{o.w}x{d['cin']}(+{c2})->{d['cout']}"
    if op == 18:          # fused stem pair (stem2.hip): the two convs it replaces; reads = the input + both weight sets, the
        m, Pm = d['hid'], x.h * x.w                      # 32-channel tensor between them no longer exists
        c2 = d['gates'][1] if len(d['gates']) > 1 else d['cout']
        fl = 2.0 * 9 * d['cin'] * m * Pm + 2.0 * 9 * m * c2 * P + (2.0 * c2 * d['cout'] * P if len(d['gates']) > 1 else 0.)
        return fl, (Pm * d['cin'] + 9 * d['cin'] * m + 9 * m * c2 + (c2 * d['cout'] if len(d['gates']) > 1 else 0)) * 2, P * d['cout'] * 2, \
            f"stem2 k3s1+k3s2{'+k1' if len(d['gates']) > 1 else ''} {x.h}x{x.w}x{d['cin']}->{o.h}x{o.w}x{d['cout']}"
    if op == 14:
        w1, _, w2, _ = d['res_ref']
        m, c = w1.shape[0], w1.shape[1]
        return 2.0 * P * (c * m + 9 * m * c), (P * c + c * m + 9 * m * c) * 2, P * c * 2, f'resblock {o.h}x{o.w}x{c} (mid {m})'
    return 0.0, x.h * x.w * x.c * 2, P * o.c * 2, f'op{op} {x.h}x{x.w}x{x.c}'


def main():
    model = sys.argv[1]
    g, _ = YOLO.get_model(model).build_graph()
    n = len(g.layers)
    sq = last_replay(open_db(sys.argv[2]), n)
    fe = last_replay(open_db(sys.argv[3]), n) if len(sys.argv) > 3 else None
    wr = last_replay(open_db(sys.argv[4]), n) if len(sys.argv) > 4 else None
    assert len(sq) == n
    print(f'# {model}: {n} launches (last replay); MFMA utilisation against {CLK / 1e9} GHz x {SIMDS} SIMDs; FETCH_SIZE x2')
    print(f'{"#":>3} {"layer":<50} {"us":>7} {"mfma busy":>11} {"expected":>10} {"util %":>7} {"fetch MB":>9} {"alg rd MB":>9} {"x":>5} {"write MB":>9} {"alg wr":>7}')
    tot = dict(busy=0.0, exp=0.0, dur=0.0, fetch=0.0, rd=0.0, write=0.0, wr=0.0, conv_dur=0.0, conv_busy=0.0)
    per_kind = {}
    for i, d in enumerate(g.layers):
        fl, rdb, wrb, label = layer_cost(g, d)
        dur = sq[i]['dur']
        busy = sq[i]['c'].get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0)
        exp = 32.0 * fl / 32768.0
        util = busy / (dur * 1e-9 * CLK * SIMDS) * 100
        fkb = fe[i]['c'].get('FETCH_SIZE', 0.0) if fe else 0.0
        wkb = wr[i]['c'].get('WRITE_SIZE', 0.0) if wr else 0.0
        fetch = 2 * fkb * 1024
        print(f'{i:>3} {label[:50]:<50} {dur / 1e3:>7.2f} {busy:>11.0f} {exp:>10.0f} {util:>7.2f} {fetch / 1e6:>9.2f} {rdb / 1e6:>9.2f} '
              f'{fetch / rdb if rdb and fe else 0:>5.2f} {wkb * 1024 / 1e6:>9.2f} {wrb / 1e6:>7.2f}')
        tot['busy'] += busy; tot['exp'] += exp; tot['dur'] += dur; tot['fetch'] += fetch; tot['rd'] += rdb
        tot['write'] += wkb * 1024; tot['wr'] += wrb
        if fl:
            tot['conv_dur'] += dur; tot['conv_busy'] += busy
        kind = label.split()[0]
        k = per_kind.setdefault(kind, dict(launches=0, dur=0.0, busy=0.0, fetch=0.0, rd=0.0))
        k['launches'] += 1; k['dur'] += dur; k['busy'] += busy; k['fetch'] += fetch; k['rd'] += rdb
    out = dict(model=model, launches=n, kernel_time_us=round(tot['dur'] / 1e3, 1),
               mfma_util=round(tot['conv_busy'] / (tot['conv_dur'] * 1e-9 * CLK * SIMDS), 4),
               mfma_busy_cycles=tot['busy'], mfma_busy_expected=tot['exp'],
               read_amplification=round(tot['fetch'] / tot['rd'], 3) if fe else None,
               fetch_bytes_per_frame=round(tot['fetch']) if fe else None, algorithmic_read_bytes=round(tot['rd']),
               write_bytes_per_frame=round(tot['write']) if wr else None, algorithmic_write_bytes=round(tot['wr']),
               per_kernel_class={k: dict(launches=v['launches'], us=round(v['dur'] / 1e3, 1),
                                         mfma_util=round(v['busy'] / (v['dur'] * 1e-9 * CLK * SIMDS), 4),
                                         read_amplification=round(v['fetch'] / v['rd'], 2) if fe and v['rd'] else None)
                                 for k, v in per_kind.items()},
               clock_ghz_assumed=CLK / 1e9, note='durations of the PMC pass; FETCH_SIZE x2 per MI355X_MICROARCH.md')
    print(f"\nconv launches: MFMA utilisation {out['mfma_util'] * 100:.2f} % of the matrix pipes over their {tot['conv_dur'] / 1e3:.1f} us "
          f"(busy {tot['busy']:.3g} cycles, expected from the FLOP count {tot['exp']:.3g})")
    if fe:
        print(f"HBM-side reads {tot['fetch'] / 1e6:.1f} MB for {tot['rd'] / 1e6:.1f} MB algorithmic = {out['read_amplification']}x"
              + (f"; writes {tot['write'] / 1e6:.1f} MB for {tot['wr'] / 1e6:.1f} MB" if wr else ''))
    print('JSON ' + json.dumps(out))


if __name__ == '__main__':
    main()
