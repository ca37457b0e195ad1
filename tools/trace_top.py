"""python tools/trace_top.py DB STEP_COUNT -- one step's kernel launches in order with durations (rocprofv3 rocpd DB)."""
import re, sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
steps = int(sys.argv[2])
rows = cur.execute('select name, start, end, grid_x, grid_y, grid_z, workgroup_x from kernels order by start').fetchall()
per = len(rows) // steps
last = rows[-per:]
t0 = last[0][1]
for name, s, e, gx, gy, gz, wx in last:
    short = name.replace('(anonymous namespace)::', '').replace('void ', '')
    short = re.sub(r'\((?:[^()]|\([^()]*\))*\)$', '', short)[:48]
    print(f'{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  blocks={gx * gy * gz // max(wx, 1):7d}  {short}')
