#!/usr/bin/env python
"""Per-kernel FETCH_SIZE / WRITE_SIZE (KB per dispatch) from two rocprofv3 --pmc passes."""
import sqlite3
import sys


def per_kernel(db, counter):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, count(*), avg(counter_value), sum(counter_value) from pmc_events "
                       "where counter_name=? group by name", (counter,)).fetchall()
    return {r[0]: (r[1], r[2], r[3]) for r in rows}


def main(fetch_db, write_db, steps):
    f, w = per_kernel(fetch_db, 'FETCH_SIZE'), per_kernel(write_db, 'WRITE_SIZE')
    print('| kernel | dispatches/step | FETCH_SIZE KB/dispatch (raw) | WRITE_SIZE KB/dispatch (raw) | fetch MB/step | write MB/step |')
    print('|---|---|---|---|---|---|')
    names = sorted(f, key=lambda k: -(f[k][2] + w.get(k, (0, 0, 0))[2]))
    for k in names[:30]:
        short = k.replace('(anonymous namespace)::', '').replace('void ', '')[:70]
        wk = w.get(k, (0, 0.0, 0.0))
        print(f'| `{short}` | {f[k][0] / steps:.1f} | {f[k][1]:.1f} | {wk[1]:.1f} | {f[k][2] / steps / 1024:.2f} | {wk[2] / steps / 1024:.2f} |')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]))
