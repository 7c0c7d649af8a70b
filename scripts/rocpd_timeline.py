"""Timeline of the last `n` kernel dispatches of a rocprofv3 rocpd database with queue/stream ids:
one line per dispatch (start offset, duration, queue, short name) -- shows what overlaps and where the
GPU idles inside a tracker step."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
print('# columns:', cols)
qcol = 'queue_id' if 'queue_id' in cols else ('queue' if 'queue' in cols else None)
scol = 'stream_id' if 'stream_id' in cols else ('stream' if 'stream' in cols else None)
sel = "start, end, name, grid_x, workgroup_x" + (f", {qcol}" if qcol else ", 0") + (f", {scol}" if scol else ", 0")
rows = db.execute(f"select {sel} from kernels order by start").fetchall()[-n:]
t0 = rows[0][0]
for i, (s, e, name, gx, wx, q, st) in enumerate(rows):
    short = name.replace('(anonymous namespace)::', '').replace('void ', '')
    short = short[:short.find('(')] if '(' in short else short
    print(f'{i:4d} t={(s - t0) / 1e3:9.1f} dur={(e - s) / 1e3:7.2f} q={q} s={st} wg={gx // max(wx, 1):6d} {short[:50]}')
