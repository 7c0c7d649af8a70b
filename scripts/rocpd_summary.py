"""Turns a rocprofv3 rocpd sqlite database (--kernel-trace --stats) into the text summary that is
committed under profiles/: per kernel name calls / total / average / share, plus the conv kernel
split by grid (tile config) so the slow layers are visible."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                   "from kernels group by name order by sum(duration) desc").fetchall()
total = sum(r[2] for r in rows)
print(f'# rocprofv3 kernel-trace summary of {sys.argv[1]} (durations in us)')
print(f'{"kernel":<78} {"calls":>7} {"total_us":>12} {"avg_us":>10} {"min_us":>9} {"max_us":>9} {"share":>7}')
for name, calls, tot, avg, mn, mx in rows:
    short = name if len(name) <= 77 else name[:74] + '...'
    print(f'{short:<78} {calls:>7} {tot / 1e3:>12.1f} {avg / 1e3:>10.2f} {mn / 1e3:>9.2f} {mx / 1e3:>9.2f} {100 * tot / total:>6.2f}%')
print(f'\ntotal GPU kernel time: {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches')
print('\n# conv_igemm_kernel by (template, grid) -- top 40 by total time')
rows = cur.execute("select name, grid_x, grid_y, workgroup_x, count(*), sum(duration), avg(duration) from kernels "
                   "where name like '%conv_igemm%' group by name, grid_x, grid_y order by sum(duration) desc limit 40").fetchall()
for name, gx, gy, wx, calls, tot, avg in rows:
    tmpl = name[name.find('<'):name.find('>') + 1] if '<' in name else name[-40:]
    print(f'{tmpl:<24} grid=({gx // wx},{gy}) calls={calls:>5} total_us={tot / 1e3:>10.1f} avg_us={avg / 1e3:>8.2f}')
