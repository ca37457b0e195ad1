"""python tools/pmc_total.py DB -- sum of every PMC counter over all dispatches + total kernel time (rocprofv3 rocpd DB)."""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select counter_name, sum(counter_value), count(*) from pmc_events group by counter_name").fetchall()
for n, v, c in rows:
    print(f'{n:36s} sum={v:18.0f} events={c}')
t = cur.execute("select min(start), max(end), sum(end-start), count(*) from kernels").fetchone()
print(f'wall {(t[1]-t[0])/1e6:.1f} ms, sum kernel time {t[2]/1e6:.1f} ms, dispatches {t[3]}')
