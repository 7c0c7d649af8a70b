"""Per (kernel, grid) averages of every counter in a rocprofv3 PMC rocpd database (pmc_events view):
python scripts/rocpd_pmc_multi.py <db>  -> one row per kernel instance group, one column per counter."""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(pmc_events)")]
gcol = 'grid_x' if 'grid_x' in cols else ('grid_size_x' if 'grid_size_x' in cols else None)
wcol = 'workgroup_x' if 'workgroup_x' in cols else ('workgroup_size_x' if 'workgroup_size_x' in cols else None)
sel = f"name, {gcol or 0}, {wcol or 1}, counter_name, count(*), sum(counter_value)"
rows = db.execute(f"select {sel} from pmc_events group by name, {gcol or 'name'}, counter_name").fetchall()
tab = defaultdict(dict)
names = []
for name, gx, wx, cn, calls, total in rows:
    short = name.replace('(anonymous namespace)::', '').replace('void ', '').replace('_ZN12_GLOBAL__N_1', '')
    short = short[:short.find('(')] if '(' in short else short
    tab[(short[:44], gx // max(wx, 1))][cn] = total / calls
    tab[(short[:44], gx // max(wx, 1))]['calls'] = calls
    if cn not in names:
        names.append(cn)
print(f'{"kernel":<46}{"wgs":>6}{"calls":>6} ' + ' '.join(f'{n[-14:]:>14}' for n in names))
for (k, g), d in sorted(tab.items(), key=lambda kv: -kv[1].get(names[0], 0) * kv[1]['calls']):
    print(f'{k:<46}{g:>6}{d["calls"]:>6} ' + ' '.join(f'{d.get(n, 0):>14.0f}' for n in names))
