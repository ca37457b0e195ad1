"""python tools/timeline.py DB RUNS -- the kernel launches of the LAST run in a rocprofv3 rocpd DB (RUNS equal runs recorded), in
start order: start offset, duration, queue, workgroups, name."""
import re, sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
runs = int(sys.argv[2])
cols = [r[1] for r in cur.execute("pragma table_info('kernels')").fetchall()]
qcol = 'queue_id' if 'queue_id' in cols else ('stream_id' if 'stream_id' in cols else None)
rows = cur.execute(f'select name, start, end, grid_x, grid_y, grid_z, workgroup_x, {qcol or 0} from kernels order by start').fetchall()
rows = [r for r in rows if 'elementwise' not in r[0] and 'copyBuffer' not in r[0]]
per = len(rows) // runs
last = rows[-per:]
t0 = last[0][1]
queues = {}
end_prev = {}
for name, s, e, gx, gy, gz, wx, q in last:
    short = name.replace('(anonymous namespace)::', '').replace('void ', '')
    short = re.sub(r'\((?:[^()]|\([^()]*\))*\)$', '', short)[:44]
    qi = queues.setdefault(q, len(queues))
    gap = (s - end_prev[q]) / 1e3 if q in end_prev else 0.0
    end_prev[q] = e
    print(f'{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  gap {gap:6.1f}  q{qi}  blocks={gx * gy * gz // max(wx, 1):7d}  {"    " * qi}{short}')
print('columns:', cols)
