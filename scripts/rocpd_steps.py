"""Per-frame, per-stream occupancy from a rocprofv3 rocpd database of bench.py: frames are delimited by the
detector's preprocess kernel; for the last `n` frames prints, for every stream, first start / last end
(us after the frame's preprocess launch) and the summed kernel time -- shows which stream bounds a step."""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
scol = 'stream_id' if 'stream_id' in cols else 'queue_id'
rows = db.execute(f"select start, end, name, {scol} from kernels order by start").fetchall()
marks = [s for s, e, name, st in rows if 'preprocess_kernel' in name]
for f in range(len(marks) - n - 1, len(marks) - 1):
    t0, t1 = marks[f], marks[f + 1]
    per = defaultdict(lambda: [1e18, 0, 0.0, 0, ''])
    for s, e, name, st in rows:
        if t0 <= s < t1:
            p = per[st]
            if s < p[0]:
                p[0] = s
                p[4] = name.replace('(anonymous namespace)::', '').replace('void ', '')[:28]
            p[1] = max(p[1], e)
            p[2] += e - s
            p[3] += 1
    print(f'frame {f}: period {(t1 - t0) / 1e3:.1f} us')
    for st, (a, b, busy, cnt, first) in sorted(per.items(), key=lambda kv: kv[1][0]):
        print(f'   stream {st}: {(a - t0) / 1e3:8.1f} .. {(b - t0) / 1e3:8.1f} us  busy {busy / 1e3:8.1f} us  {cnt:4d} kernels  first={first}')
