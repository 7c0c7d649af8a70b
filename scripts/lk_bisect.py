"""Round-3 bisect of the LK results that differ under load: un-isolated LK launches beside a hammer that
replays the fused LightConv launches of OSNet-x0.25 on another stream.

  (1) rates: production kernel, the diagnostic body with DPP sums, with LDS sums, with both (compared in the kernel),
      with duplicate loads + lane-agreement checks;
  (2) capture: the DPP kernel writes every (level, iteration)'s samples, running sums, broadcast sums and every lane's
      position copy; a differing call is compared record by record against the idle capture of the same call parity:
      the FIRST differing quantity says whether a load, a scan step, the broadcast or the uniform arithmetic went
      wrong; HW_ID / XCC_ID of the disturbed waves are listed.

    python scripts/lk_bisect.py [calls per experiment] [hammer: litechain|osnet|yolo]
"""
import os as _os
_os.environ.setdefault('FASTMOT_RANDOM_WEIGHTS', '1')
import sys, threading
sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
sys.path.insert(1, _os.path.join(sys.path[0], 'tests'))
import numpy as np
from synthetic import SyntheticVideo
from fastmot_amd.flow import Flow
from fastmot_amd.detector import DeviceFrame, bind_frame
from fastmot_amd.runtime import get_context
sys.path.insert(1, _os.path.dirname(_os.path.abspath(__file__)))
from diag_bindings import flow_lk_diag, diag_pkhaz, diag_pkhaz2   # needs a -DFM_DIAG build
from fastmot_amd.engine import HipNet, NET_EXTRACTOR, NET_DETECTOR
from fastmot_amd.models import ReID, YOLO

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
HAMMER = sys.argv[2] if len(sys.argv) > 2 else 'litechain'
ONLY = sys.argv[3].split(',') if len(sys.argv) > 3 else None
NPTS = 600
size = (960, 540)
video = SyntheticVideo(size, n_ids=6, n_frames=2, seed=4)
ctx = get_context()
ctx.feat_configure(512)
ctx.frame_configure(size[0], size[1], 2)
for i in range(2):
    ctx.frame_ring_store(i, video.frames[i])
flow = Flow(size)
flow.init(DeviceFrame(0))
bind_frame(ctx, DeviceFrame(1), size)
ctx.flow_begin()
ctx.synchronize()
rng = np.random.default_rng(0)
pts = np.stack([rng.uniform(20, size[0] / 2 - 20, NPTS), rng.uniform(20, size[1] / 2 - 20, NPTS)], 1).astype(np.float32)
ctx.set_option('lk_variant', 0)
base = [ctx.flow_lk(pts), ctx.flow_lk(pts)]      # consecutive calls track in opposite directions (the call swaps the image sets)

VARIANTS = {'production': 0, 'diag-dpp': 1 | 64, 'diag-dpp+capture': 1 | 32, 'diag-dpp+checks': 1 | 16, 'diag-lds': 2,
            'diag-lds+checks': 2 | 16, 'diag-both': 3, 'diag-both+checks': 3 | 16}


def same(res, k):
    nxt, st, er = res
    b = base[k]
    if not np.array_equal(st, b[1]):
        return False, np.flatnonzero(st != b[1])
    ok = st > 0
    bad = np.flatnonzero(ok & ((nxt != b[0]).any(1) | (er != b[2])))
    return len(bad) == 0, bad


# ---- idle: every variant must reproduce the production result bit for bit
cap_idle = [None, None]
for name, v in VARIANTS.items():
    ctx.set_option('lk_variant', v)
    oks = []
    for k in range(2):
        r = ctx.flow_lk(pts)
        oks.append(same(r, k)[0])
        if v == (1 | 32):
            _, hdr, rec = flow_lk_diag(ctx, NPTS)
            cap_idle[k] = (hdr.copy(), rec.copy())
    c = flow_lk_diag(ctx)[0]
    print(f'idle {name:<18} identical to production: {oks}  counters {c[:12].tolist()}', flush=True)
# idle capture twice more: is the capture itself reproducible?
ctx.set_option('lk_variant', 1 | 32)
for k in range(2):
    ctx.flow_lk(pts)
    _, hdr, rec = flow_lk_diag(ctx, NPTS)
    nrec = hdr[:, 2]
    eq = all(np.array_equal(rec[i, :nrec[i]], cap_idle[k][1][i, :nrec[i]]) for i in range(NPTS))
    print(f'idle capture parity {k}: records reproducible {eq}, records per point {nrec.min()}..{nrec.max()}', flush=True)
ctx.set_option('lk_variant', 0)


def make_hammer():
    if HAMMER == 'yolo':
        g, _ = YOLO.get_model('YOLOv4_608').build_graph()
        net = HipNet(ctx, NET_DETECTOR, g, 1, reuse_buffers=True)
        batch = 1
    else:
        g, _ = ReID.get_model('OSNet025').build_graph()
        if HAMMER == 'litechain':
            g.layers[:] = [d for d in g.layers if d['op'] == 16]
        net = HipNet(ctx, NET_EXTRACTOR, g, 50, reuse_buffers=False)
        batch = 50
    net.run(batch)
    ctx.synchronize()
    return net, batch


net, batch = make_hammer()
stop = []


def hammer():
    ctx.bind_thread()
    while not stop:
        net.run(batch)
        ctx.synchronize()


def decode_hw(hw, xcc):
    return dict(xcc=xcc & 15, se=(hw >> 13) & 7, sh=(hw >> 12) & 1, cu=(hw >> 8) & 15, simd=(hw >> 4) & 3, wave=hw & 15)


ROWS_A = ['ival', 'ixval', 'iyval', 'pfxA11', 'pfxA12', 'pfxA22', 'A11', 'meta', 'r8', 'r9', 'r10', 'r11']
ROWS_I = ['diff', 'pfx_b1', 'pfx_b2', 'nx', 'ny', 'b1', 'b2', 'meta', 'nx_before', 'b1_scaled', 'dx', 'dy']
ORDER_A = [7, 0, 1, 2, 3, 4, 5, 6]
ORDER_I = [7, 0, 1, 2, 5, 6, 8, 9, 10, 11, 3, 4]


def seq_prefix(v):
    out = np.zeros(len(v), np.float32)
    acc = np.float32(0)
    for i, x in enumerate(v):
        acc = np.float32(acc + np.float32(x))
        out[i] = acc
    return out


def analyse(i, hdr, rec, ref_hdr, ref_rec, verbose):
    n_bad, n_ref = int(hdr[i, 2]), int(ref_hdr[i, 2])
    where = decode_hw(int(hdr[i, 0]), int(hdr[i, 1]))
    last_a = None
    for t in range(min(n_bad, n_ref, 80)):
        r, q = rec[i, t], ref_rec[i, t]
        typ = int(r[7, 0]) & 255
        if typ == 1:
            last_a = r
        if np.array_equal(r, q):
            continue
        names, order = (ROWS_A, ORDER_A) if typ == 1 else (ROWS_I, ORDER_I)
        first = next(row for row in order if not np.array_equal(r[row], q[row]))
        lanes = np.flatnonzero(r[first] != q[first])
        meta = int(r[7, 0])
        desc = f'point {i} record {t}/{n_ref} type {typ} level {(meta >> 8) & 255} iter {(meta >> 16) & 255}: first differing row = {names[first]}, lanes {lanes.tolist()[:40]}'
        if verbose:
            print('   ', desc, where, flush=True)
            fl = names[first].startswith(('pfx', 'A11', 'b', 'n'))
            view = (lambda a: a.view(np.float32)) if fl else (lambda a: a)
            print('      ref :', view(q[first])[lanes][:4].tolist())
            print('      got :', view(r[first])[lanes][:4].tolist())
            if typ == 2:
                for row in (8, 9, 10, 11, 3, 4):
                    a, b = r[row].view(np.float32), q[row].view(np.float32)
                    print(f'      {names[row]:<10} lane0 {a[0]!r:>22} lane63 {a[63]!r:>22} | idle lane0 {b[0]!r:>22} lane63 {b[63]!r:>22} | lanes differing from lane 0: {np.flatnonzero(a != a[0]).tolist()}')
            if names[first].startswith('pfx') and typ == 2 and last_a is not None:
                col = 1 if names[first] == 'pfx_b1' else 2
                v = (r[0][:25].astype(np.int64) * last_a[col][:25].astype(np.int64)).astype(np.float32)
                exp = seq_prefix(v)
                got = r[first][:25].view(np.float32)
                print('      terms   :', v.tolist())
                print('      expected:', exp.tolist())
                print('      got     :', got.tolist())
                # what would produce the wrong lane?  candidates: a term skipped, a term doubled, left neighbour stale by one step
                for L in lanes[:3]:
                    if L >= 25:
                        continue
                    cands = {}
                    for skip in range(L + 1):
                        cands[f'skip term {skip}'] = seq_prefix(np.delete(v[:L + 1], skip))[-1] if L > 0 else np.float32(0)
                    for dup in range(L + 1):
                        cands[f'term {dup} twice'] = seq_prefix(np.insert(v[:L + 1], dup, v[dup]))[-1]
                    hit = [k for k, c in cands.items() if np.float32(c) == got[L]]
                    print(f'      lane {L}: got {got[L]!r} expected {exp[L]!r} delta {float(got[L]) - float(exp[L]):+.1f} explanations: {hit[:6]}')
        return names[first], typ, where
    return ('records equal, outputs differ' if n_bad == n_ref else 'record count'), 0, where


def run(name, calls, capture=False, max_verbose=8):
    v = VARIANTS[name]
    ctx.set_option('lk_variant', v)
    flow_lk_diag(ctx)
    bad_calls = bad_pts = 0
    kinds = {}
    places = []
    shown = 0
    for r in range(calls):
        res = ctx.flow_lk(pts)
        k = r % 2
        ok, bad = same(res, k)
        if ok:
            continue
        bad_calls += 1
        bad_pts += len(bad)
        if capture and bad_calls <= 40:
            _, hdr, rec = flow_lk_diag(ctx, NPTS)
            for i in bad[:4]:
                what, typ, where = analyse(int(i), hdr, rec, cap_idle[k][0], cap_idle[k][1], shown < max_verbose)
                shown += 1
                kinds[(what, typ)] = kinds.get((what, typ), 0) + 1
                places.append(where)
    c = flow_lk_diag(ctx)[0]
    print(f'hammer={HAMMER} variant={name:<18} calls differing {bad_calls}/{calls}, points {bad_pts}; counters '
          f'[sum mismatch, dpp changed, lds changed, unresolved, dup-load mismatch, lanes disagree, iterations] = {c[:7].tolist()} disagreeing lanes by quarter {c[8:12].tolist()}', flush=True)
    if capture:
        print('   first differing quantity histogram:', kinds)
        print('   disturbed waves (xcc, se, sh, cu, simd, wave):', [tuple(p.values()) for p in places][:60])
    ctx.set_option('lk_variant', 0)
    return bad_calls


th = threading.Thread(target=hammer)
th.start()
try:
    for name in ONLY or ['production', 'diag-dpp', 'diag-dpp+capture', 'diag-dpp+checks', 'diag-lds', 'diag-lds+checks',
                         'diag-both', 'diag-both+checks', 'production']:
        run(name, N, capture=name.endswith('capture'))
finally:
    stop.append(1)
    th.join()
    pass
# where do waves of an idle call land?  (reference distribution for the disturbed-wave list)
hdr = cap_idle[0][0]
xcc = hdr[:, 1] & 15
print('idle capture: waves per XCC', np.bincount(xcc, minlength=8).tolist())
