"""Sums one rocprofv3 PMC counter per kernel name from a rocpd database (pmc_events view):
python scripts/rocpd_pmc.py <db> <COUNTER>  ->  table + one JSON line (kernel -> {calls, total, per_call})."""
import json
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
counter = sys.argv[2]
rows = db.execute("select name, count(*), sum(counter_value) from pmc_events where counter_name = ? "
                  "group by name order by sum(counter_value) desc", (counter,)).fetchall()
out = {}
print(f'# {counter} per kernel ({sys.argv[1]})')
for name, calls, total in rows:
    short = name.replace('(anonymous namespace)::', '').replace('void ', '')
    short = short[:short.find('(')] if '(' in short else short
    out[short] = dict(calls=calls, total=total, per_call=total / calls)
    print(f'{short[:70]:<72} calls={calls:>6} total={total:>14.1f} per_call={total / calls:>12.2f}')
print('JSON ' + json.dumps(out))
