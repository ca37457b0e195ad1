"""Markdown table from the rocprofv3 databases of tools/pmc_kpconv_forms.sh."""
import glob, os, sqlite3, sys
root = sys.argv[1]
shapes = ['L0->L0 32->32 (32 000 queries)', 'L0->L1 32->32 strided (10 961)', 'L1->L1 64->64 (10 961)', 'L1->L2 64->64 strided (3 879)']
print('# KPConv as one kernel: lock-step form (every (query, neighbour) row fetched from L2) against the LDS-tile form (16 cell-ordered')
print('# queries per workgroup, union of their rows staged once), rocprofv3 --kernel-trace --pmc <one group per pass>, 5 launches per form,')
print('# per-launch averages; MI355X, r04.  TCP_TCC_READ_REQ = read requests a CU\'s L1 sends to L2 (64 B each); TCC_REQ = all L2 requests;')
print('# FETCH_SIZE / WRITE_SIZE in KB (FETCH raw: x2 for wide reads on gfx950); SQ_* summed over the chip (WAVE_CYCLES etc. in quad-cycles).')
for si, sname in enumerate(shapes):
    vals = {}
    dur = {}
    for db in glob.glob(os.path.join(root, f's{si}_p*', '**', '*.db'), recursive=True):
        cur = sqlite3.connect(db).cursor()
        try:
            rows = cur.execute('select name, counter_name, avg(counter_value) from pmc_events group by name, counter_name').fetchall()
        except sqlite3.Error:
            continue
        for name, cn, v in rows:
            form = 'tile' if 'kpconv_tile_kernel' in name else ('lock' if 'kpconv_fused_kernel' in name else None)
            if form:
                vals.setdefault(cn, {})[form] = v
        for name, n, t in cur.execute('select name, count(*), avg(end-start) from kernels group by name'):
            form = 'tile' if 'kpconv_tile_kernel' in name else ('lock' if 'kpconv_fused_kernel' in name else None)
            if form:
                dur.setdefault(form, []).append(t)
    print(f'\n## {sname}\n')
    print('| counter | lock-step | LDS tile | tile / lock-step |')
    print('|---|---|---|---|')
    if dur:
        a, b = (sum(dur.get(f, [0])) / max(len(dur.get(f, [])), 1) / 1e3 for f in ('lock', 'tile'))
        print(f'| kernel duration under the profiler, us | {a:.1f} | {b:.1f} | {b / a if a else 0:.2f} |')
    for cn in sorted(vals):
        a, b = vals[cn].get('lock', 0.0), vals[cn].get('tile', 0.0)
        print(f'| {cn} | {a:,.0f} | {b:,.0f} | {b / a if a else 0:.2f} |')
