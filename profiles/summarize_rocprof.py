#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (kernel trace) into a per-kernel table (markdown)."""
import re
import sqlite3
import sys


def short_name(name):
    """demangled name without namespace and argument list; a lock-step grouped launch (rdm::grouped_kernel<body, ...>, which
    rocprofv3 leaves mangled) as `grouped <body><template arguments>`"""
    m = re.search(r'grouped_kernel.*?N_1\d+([a-z0-9_]+_(?:body|entry))((?:I(?:Li\d+E|Lb[01]E)+E)?)', name)
    if m:
        args = re.findall(r'L[ib](\d+)E', m.group(2))
        return 'grouped ' + m.group(1) + ('<' + ', '.join(args) + '>' if args else '')
    short = name.replace('(anonymous namespace)::', '').replace('void ', '')
    return re.sub(r'\((?:[^()]|\([^()]*\))*\)$', '', short)[:90]


def main(db_path, steps):
    cur = sqlite3.connect(db_path).cursor()
    raw = cur.execute('select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) '
                      'from kernels group by name order by 3 desc').fetchall()
    merged = {}
    for name, calls, total, avg, mn, mx in raw:  # (argument packs of a grouped launch differ in their mangled tails only)
        k = short_name(name)
        c = merged.setdefault(k, [k, 0, 0, 0.0, mn, mx])
        c[1] += calls; c[2] += total; c[4] = min(c[4], mn); c[5] = max(c[5], mx)
    rows = sorted(([k, c, t, t / c, mn, mx] for k, c, t, _, mn, mx in merged.values()), key=lambda r: -r[2])
    tot = sum(r[2] for r in rows)
    if steps <= 0:  # pairs the process ran = dispatches of a once-per-pair kernel
        steps = max(1, sum(r[1] for r in rows if 'export_result_kernel' in r[0]))
    print(f'total kernel time {tot / 1e6:.3f} ms over {steps} steps = {tot / steps / 1e6:.3f} ms/step\n')
    print('| kernel | calls/step | total ms | avg us | min us | max us | % |')
    print('|---|---|---|---|---|---|---|')
    for name, calls, total, avg, mn, mx in rows:
        short = name
        print(f'| `{short}` | {calls / steps:.1f} | {total / 1e6:.3f} | {avg / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100 * total / tot:.1f} |')


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1)
