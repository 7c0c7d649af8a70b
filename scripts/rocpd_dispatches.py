"""Lists the last N kernel dispatches of a rocprofv3 rocpd database in start order: start offset,
duration, gap to the previous dispatch's end, grid, name (per-layer view of one graph replay)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2])
rows = db.execute("select start, end, name, grid_x, grid_y, grid_z, workgroup_x from kernels order by start").fetchall()[-n:]
t0, prev_end = rows[0][0], rows[0][0]
busy = 0
for i, (s, e, name, gx, gy, gz, wx) in enumerate(rows):
    short = name.replace('(anonymous namespace)::', '').replace('void ', '')
    short = short[:short.find('(')] if '(' in short else short
    print(f'{i:3d} t={(s - t0) / 1e3:8.1f} dur={(e - s) / 1e3:7.2f} gap={(s - prev_end) / 1e3:6.2f} grid=({gx // wx},{gy},{gz}) {short[:60]}')
    prev_end = max(prev_end, e)
    busy += e - s
print(f'span {(prev_end - t0) / 1e3:.1f} us, kernel time {busy / 1e3:.1f} us')
