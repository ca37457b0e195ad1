"""Refit of the GEMM dispatch model (gemm.hip: gemm_dispatch) against a full tile/split sweep
(tools/gemm_sweep_graph.py <json>): a Python replica of the model picks a configuration per shape, the measured time of
that configuration is summed, and a coordinate search over the model's constants minimises the sum.

    python tools/gemm_model_fit.py tools/data/gemm_sweep_r01.json
"""
import itertools, json, math, sys

TILES = {1: (128, 128), 2: (64, 64), 3: (128, 32)}


def choose(m, k, n, P):
    """P = {tile: (fixed, lat, mfma, resident)}, P['red'] = (fixed, bytes_per_us)."""
    best = (1e30, 2, 1)
    for t, (bm, bn) in TILES.items():
        if t == 3 and n > 64: continue
        if t != 3 and n <= 32: continue
        if t == 1 and n < 128: continue
        fixed, lat, mfma, res, alpha = P[t]
        tiles = math.ceil(m / bm) * math.ceil(n / bn)
        max_s = min(16, k // 128) if k >= 256 else 1
        for sp in (1, 2, 3, 4, 6, 8, 12, 16):  # the swept split factors (the dispatch uses the same set)
            if sp > max_s: break
            if sp > 1 and m * n * sp * 4 > min(m * n * 16, 1024 * 128 * 128) * 4: break
            blocks = tiles * sp
            k_per = k / sp / 64.0
            per_round = 256 * res
            full, rem = divmod(blocks, per_round)
            per = lambda r: max(lat * (1.0 + alpha * (r - 1)), mfma * r)  # r blocks sharing a CU
            tt = full * (fixed + k_per * per(res))
            if rem: tt += fixed + k_per * per(math.ceil(rem / 256))
            if sp > 1: tt += P['red'][0] + m * n * sp * 8.0 / P['red'][1]
            if tt < best[0]: best = (tt, t, sp)
    return best[1], best[2]


def total(table, P):
    s = 0.0
    for name, e in table.items():
        t, sp = choose(e['m'], e['k'], e['n'], P)
        meas = {(c[0], c[1]): c[2] for c in e['configs']}
        us = meas.get((t, sp)) or meas.get((t, 0 if sp == 1 else sp)) or meas.get((t, 1 if sp == 1 else sp))
        if us is None:  # configuration not swept (e.g. 5, 7 splits): nearest swept split count
            cand = [(abs(c[1] - sp), c[2]) for c in e['configs'] if c[0] == t]
            us = min(cand)[1] if cand else 1e3
        s += us
    return s


if __name__ == '__main__':
    table = json.load(open(sys.argv[1]))
    P = {1: [4.0, 2.6, 3.4, 3, 0.0], 2: [3.0, 2.6, 0.86, 4, 0.0], 3: [3.0, 1.9, 0.86, 3, 0.0], 'red': [5.0, 2.5e6]}
    best = sum(min(c[2] for c in e['configs']) for e in table.values())
    print(f'current constants: {total(table, P):.0f} us; per-shape optimum {best:.0f} us')
    grids = {(2, 0): [1.5, 2.0, 2.5, 3.0, 3.5], (2, 1): [1.6, 1.9, 2.2, 2.6, 3.0], (2, 2): [0.6, 0.75, 0.86, 1.0],
             (3, 0): [1.5, 2.0, 2.5, 3.0], (3, 1): [1.3, 1.6, 1.9, 2.3], (3, 2): [0.6, 0.86, 1.0],
             (2, 4): [0.0, 0.1, 0.2, 0.3, 0.4, 0.6], (3, 4): [0.0, 0.1, 0.2, 0.3, 0.4, 0.6],
             ('red', 0): [3.0, 4.0, 5.0, 6.0, 8.0], ('red', 1): [1.5e6, 2.5e6, 4e6, 6e6]}
    cur = total(table, P)
    for sweep in range(4):
        for (key, idx), vals in grids.items():
            for v in vals:
                old = P[key][idx]
                P[key][idx] = v
                t = total(table, P)
                if t < cur - 1e-9: cur = t
                else: P[key][idx] = old
    print(f'fitted: {cur:.0f} us with', P)
    for name, e in table.items():
        t, sp = choose(e['m'], e['k'], e['n'], P)
        print(f"  {name:6s} -> tile {t}, splits {sp}; best measured {min(e['configs'], key=lambda c: c[2])}")
