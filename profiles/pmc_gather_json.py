#!/usr/bin/env python
"""HBM traffic per KPConv neighbourhood-kernel dispatch (kpconv_gather*, kpconv_fused* and kpconv_tile*) from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) -> json."""
import json
import sqlite3
import sys


FORMS = {'one_kernel_layers': ("(name like '%kpconv_fused%' or (name like '%kpconv_tile%' and name not like '%Lb1E%' and name not like '%, true>%'))"),
         'gather_kernel_layers': ("(name like '%kpconv_gather%' or (name like '%kpconv_tile%' and (name like '%Lb1E%' or name like '%, true>%')))")}


def total_form(db, counter, grouped, form):
    """per form: one-kernel layers (c_in 1 / 32 / 64: kpconv_fused*, kpconv_tile<C, false>) and gather-only
    layers (c_in >= 128: kpconv_gather_kernel<*>, kpconv_tile<64, true>)"""
    cur = sqlite3.connect(db).cursor()
    return cur.execute("select count(*), sum(counter_value) from pmc_events where counter_name=? and " + FORMS[form] + " and name " +
                       ("like" if grouped else "not like") + " '%grouped_kernel%'", (counter,)).fetchall()[0]


def total(db, counter, grouped):
    """grouped: the lock-step grouped dispatches only (rdm::grouped_kernel<kpconv_*_body, ...>: one dispatch serves the pairs of a
    group) -- the launches of bench.py's timed region since round 5; otherwise the one-pair dispatches."""
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select count(*), sum(counter_value) from pmc_events where counter_name=? and (name like '%kpconv_gather%' or name like '%kpconv_fused%' or name like '%kpconv_tile%') "
                       "and name " + ("like" if grouped else "not like") + " '%grouped_kernel%'", (counter,)).fetchall()
    return rows[0]


def main(fetch_db, write_db, source, grouped=False):
    nf, f = total(fetch_db, 'FETCH_SIZE', grouped)
    nw, w = total(write_db, 'WRITE_SIZE', grouped)
    fetch_kb, write_kb = f / nf, w / nw
    out = {'source': source,
           'kernel': ('lock-step grouped dispatches (rdm::grouped_kernel<body>: one dispatch = the 4 pairs of a group) of ' if grouped else '') +
                     'kpconv_fused_c1_kernel + kpconv_tile_kernel<32|64> / kpconv_fused_kernel<64> + kpconv_gather_kernel<*> (14 dispatches per pair' + (' group)' if grouped else ')'),
           'dispatches': nf, 'fetch_kb_per_dispatch_raw': fetch_kb, 'write_kb_per_dispatch_raw': write_kb,
           'correction': 'MI355X_MICROARCH.md §HBM: FETCH_SIZE under-reports wide streaming reads by 2x on gfx950 -> reads '
                         'doubled; WRITE_SIZE uncalibrated, taken as is',
           'traffic_bytes_per_dispatch': (2 * fetch_kb + write_kb) * 1024, 'by_form': {}}
    for form in FORMS:
        (n1, f1), (n2, w1) = total_form(fetch_db, 'FETCH_SIZE', grouped, form), total_form(write_db, 'WRITE_SIZE', grouped, form)
        if n1 and n2:
            out['by_form'][form] = {'dispatches': n1, 'fetch_kb_per_dispatch_raw': f1 / n1, 'write_kb_per_dispatch_raw': w1 / n2,
                                    'traffic_bytes_per_dispatch': (2 * f1 / n1 + w1 / n2) * 1024}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], sys.argv[3], len(sys.argv) > 4 and sys.argv[4] == 'grouped')
