#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (kernel trace) into a per-kernel table (markdown)."""
import re
import sqlite3
import sys


def main(db_path, steps):
    cur = sqlite3.connect(db_path).cursor()
    rows = cur.execute('select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) '
                       'from kernels group by name order by 3 desc').fetchall()
    tot = sum(r[2] for r in rows)
    if steps <= 0:  # pairs the process ran = dispatches of a once-per-pair kernel
        steps = max(1, sum(r[1] for r in rows if 'nms_kernel' in r[0]))
    print(f'total kernel time {tot / 1e6:.3f} ms over {steps} steps = {tot / steps / 1e6:.3f} ms/step\n')
    print('| kernel | calls/step | total ms | avg us | min us | max us | % |')
    print('|---|---|---|---|---|---|---|')
    for name, calls, total, avg, mn, mx in rows:
        short = name.replace('(anonymous namespace)::', '').replace('void ', '')
        short = re.sub(r'\((?:[^()]|\([^()]*\))*\)$', '', short)[:90]
        print(f'| `{short}` | {calls / steps:.1f} | {total / 1e6:.3f} | {avg / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100 * total / tot:.1f} |')


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1)
