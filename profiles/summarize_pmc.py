#!/usr/bin/env python
"""Per-kernel FETCH_SIZE / WRITE_SIZE (KB per dispatch) from two rocprofv3 --pmc passes."""
import sqlite3
import sys


def per_kernel(db, counter):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, count(*), avg(counter_value), sum(counter_value) from pmc_events "
                       "where counter_name=? group by name", (counter,)).fetchall()
    return {r[0]: (r[1], r[2], r[3]) for r in rows}


def pairs_in(db, counter):
    """Scan pairs the profiled process ran = dispatches of a once-per-pair kernel (export_result_kernel)."""
    cur = sqlite3.connect(db).cursor()
    return cur.execute("select count(*) from pmc_events where counter_name=? and name like '%export_result_kernel%'", (counter,)).fetchone()[0]


def main(fetch_db, write_db, steps):
    f, w = per_kernel(fetch_db, 'FETCH_SIZE'), per_kernel(write_db, 'WRITE_SIZE')
    if steps <= 0:  # count them (round 2 passed a constant that did not match the run: 14 against 10 pairs)
        steps = pairs_in(fetch_db, 'FETCH_SIZE')
        assert steps == pairs_in(write_db, 'WRITE_SIZE'), 'the two passes ran different numbers of pairs'
    print(f'pairs in each pass: {steps}; whole run per pair: FETCH_SIZE {sum(v[2] for v in f.values()) / steps / 1024:.1f} MB raw '
          f'(x2 on gfx950 for wide reads = {2 * sum(v[2] for v in f.values()) / steps / 1024:.1f} MB), WRITE_SIZE '
          f'{sum(v[2] for v in w.values()) / steps / 1024:.1f} MB\n')
    print('| kernel | dispatches/pair | FETCH_SIZE KB/dispatch (raw) | WRITE_SIZE KB/dispatch (raw) | fetch MB/pair | write MB/pair |')
    print('|---|---|---|---|---|---|')
    names = sorted(f, key=lambda k: -(f[k][2] + w.get(k, (0, 0, 0))[2]))
    for k in names[:30]:
        short = k.replace('(anonymous namespace)::', '').replace('void ', '')[:70]
        wk = w.get(k, (0, 0.0, 0.0))
        print(f'| `{short}` | {f[k][0] / steps:.1f} | {f[k][1]:.1f} | {wk[1]:.1f} | {f[k][2] / steps / 1024:.2f} | {wk[2] / steps / 1024:.2f} |')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]))
