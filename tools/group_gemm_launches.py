"""The dense-product launches of ONE lock-step group (one stream, GPU to itself) from a rocprofv3 kernel trace of
`bench.py --streams 1 --lockstep 4`: position in the group's launch sequence, kernel, workgroups, duration -- and, given a second
trace of another build / knob setting, the two durations side by side (the launch sequences match one to one).
    python tools/group_gemm_launches.py <trace.db> [<other.db>]"""
import glob, os, re, sqlite3, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles'))
from summarize_rocprof import short_name


def group_launches(db, which=20):
    if os.path.isdir(db):
        db = glob.glob(os.path.join(db, '**', '*.db'), recursive=True)[0]
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute('select name,start,end,grid_x,workgroup_x from kernels order by start').fetchall()
    names = [short_name(r[0]) for r in rows]
    idx = [i for i, n in enumerate(names) if 'concat_points' in n]
    starts = [idx[k] for k in range(len(idx) - 3) if idx[k + 3] - idx[k] == 3]  # four collates in a row = a group of four
    s, e = starts[which], starts[which + 1]
    out = []
    for i in range(s, e):
        if re.search(r'gemm_(kernel|wide|small)|splitk', names[i]):
            out.append((i - s, names[i], rows[i][3] // max(rows[i][4], 1), (rows[i][2] - rows[i][1]) / 1e3))
    return out, sum((rows[i][2] - rows[i][1]) / 1e3 for i in range(s, e)), (rows[e - 1][2] - rows[s][1]) / 1e3


a, a_all, a_wall = group_launches(sys.argv[1])
b = group_launches(sys.argv[2]) if len(sys.argv) > 2 else None
print(f'group: all kernels {a_all:.0f} us, wall {a_wall:.0f} us' + (f'; other: {b[1]:.0f} us, wall {b[2]:.0f} us' if b else ''))
tot = [0.0, 0.0]
for k, (pos, name, wgs, us) in enumerate(a):
    line = f'{pos:4d} {name[:64]:64s} {wgs:6d} wgs {us:8.1f} us'
    tot[0] += us
    if b and k < len(b[0]):
        line += f'   | {b[0][k][1][:40]:40s} {b[0][k][2]:6d} wgs {b[0][k][3]:8.1f} us  {b[0][k][3] - us:+7.1f}'
        tot[1] += b[0][k][3]
    print(line)
print(f'dense products: {tot[0]:.0f} us' + (f' | {tot[1]:.0f} us' if b else ''))
