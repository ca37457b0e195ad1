"""Per-kernel-class GPU time per ORIGINAL pair from rocprofv3 kernel traces of tools/batch_lab.py runs:
python tools/batch_lab_classes.py DB_a B_a [DB_b B_b]"""
import re, sqlite3, sys

CLASSES = [('tiled GEMMs', r'gemm_kernel|splitk_reduce'), ('KPConv one-kernel', r'kpconv_tile|kpconv_fused'), ('KPConv gathers', r'kpconv_gather'),
           ('GroupNorm', r'gn_'), ('shortcut pools / upsample', r'gather_max|upsample_concat'), ('radius search', r'rn_'),
           ('grid subsampling', r'grid_subsample|gs_'), ('superpoint stages (not comparable: B x the tokens in one problem)',
            r'attention|gemm_small|linear_ln|rope|layernorm|nms|p2n|coarse|topk|select_nodes|compact|sinkhorn|lgr|gather_rows|vote|sigmoid|l2_norm')]


def per_class(db, B):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute('select name, count(*), sum(end-start) from kernels group by name').fetchall()
    runs = max(1, sum(r[1] for r in rows if 'nms_kernel' in r[0]))  # one NMS per engine run
    out, other = {}, 0.0
    for name, _, total in rows:
        for cls, pat in CLASSES:
            if re.search(pat, name):
                out[cls] = out.get(cls, 0.0) + total
                break
        else:
            other += total
    out['other'] = other
    return {k: v / 1e3 / runs / B for k, v in out.items()}, runs


if __name__ == '__main__':
    a, ra = per_class(sys.argv[1], int(sys.argv[2]))
    cols = [(f'B = {sys.argv[2]} ({ra} runs)', a)]
    if len(sys.argv) > 4:
        b, rb = per_class(sys.argv[3], int(sys.argv[4]))
        cols.append((f'B = {sys.argv[4]} ({rb} runs)', b))
    print('| kernel class | ' + ' | '.join(f'us per original pair, {n}' for n, _ in cols) + (' | ratio |' if len(cols) == 2 else ' |'))
    print('|---|' + '---|' * (len(cols) + (1 if len(cols) == 2 else 0)))
    for cls in [c for c, _ in CLASSES] + ['other']:
        vals = [c.get(cls, 0.0) for _, c in cols]
        line = f'| {cls} | ' + ' | '.join(f'{v:.1f}' for v in vals)
        if len(cols) == 2:
            line += f' | {vals[1] / vals[0]:.2f}' if vals[0] > 0 else ' | -'
        print(line + ' |')
    tot = [sum(v for k, v in c.items() if not k.startswith('superpoint')) for _, c in cols]
    print('| **linear classes, total** | ' + ' | '.join(f'{v:.1f}' for v in tot) + (f' | {tot[1] / tot[0]:.2f} |' if len(cols) == 2 else ' |'))
