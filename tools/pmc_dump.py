"""python tools/pmc_dump.py DB [kernel-substring] -- average PMC counter values per kernel from a rocprofv3 rocpd DB."""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
pat = sys.argv[2] if len(sys.argv) > 2 else ''
cols = [r[1] for r in cur.execute("pragma table_info(pmc_events)").fetchall()]
if '--cols' in sys.argv:
    print(cols)
kcol = 'name' if 'name' in cols else 'kernel_name'
rows = cur.execute(f"select {kcol}, counter_name, count(*), avg(counter_value) from pmc_events group by {kcol}, counter_name").fetchall()
for name, c, cnt, avg in rows:
    if pat in str(name):
        print(f'{str(name)[:60]:60s} {c:32s} n={cnt:4d} avg={avg:14.1f}')
if not rows:
    print('no rows; tables:', [r[0] for r in cur.execute("select name from sqlite_master").fetchall()][:40])
