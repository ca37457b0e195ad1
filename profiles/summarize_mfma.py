#!/usr/bin/env python
"""Per-kernel matrix-core / VALU utilisation from ONE rocprofv3 --pmc pass with
SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE
(kernel trace in the same pass for names and durations).  Columns:
  TFLOP/s f32   = SQ_INSTS_VALU_MFMA_MOPS_F32 x 512 FLOP / kernel time  (one MOP = 512 FLOP: a 16x16x4 f32 MFMA counts 4, a
                  32x32x2 counts 8; cross-checked on the GEMM rows whose flops follow from the shapes), and its share of the
                  157.3 TFLOP/s fp32 matrix peak (MI355X_MICROARCH.md) = the MFMA utilisation while the kernel runs
  mfma busy %   = SQ_VALU_MFMA_BUSY_CYCLES / (kernel time x 2.4 GHz x 1024 SIMDs): a lower bound (the chip clocks below 2.4 GHz)
  wait %        = SQ_WAIT_ANY / SQ_WAVE_CYCLES: wave-cycles parked at s_waitcnt / barriers
pmc_events holds one row per (dispatch, counter, hardware instance); SQ counters are summed over the instances.
Pairs in the run = dispatches of concat_points_kernel (one per scan pair)."""
import re
import sqlite3
import sys


def main(db, steps=0):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute('select name, counter_name, sum(counter_value) from pmc_events group by name, counter_name').fetchall()
    disp = {r[0]: (r[1], r[2]) for r in cur.execute('select name, count(*), sum(end-start) from kernels group by name').fetchall()}
    pairs = max([v[0] for n, v in disp.items() if 'concat_points_kernel' in n] + [1]) if steps <= 0 else steps
    k = {}
    for name, c, v in rows:
        k.setdefault(name, {})[c] = v
    ttot = sum(v[1] for v in disp.values())
    tot = {c: sum(d.get(c, 0) for d in k.values()) for c in ('SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY',
                                                             'SQ_ACTIVE_INST_VALU', 'SQ_INSTS_VALU_MFMA_MOPS_F32')}
    tf = tot["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512 / max(ttot, 1) / 1e3
    print(f'{pairs} pairs, all kernels: {ttot / pairs / 1e6:.3f} ms/pair; fp32 MFMA {tf:.1f} TFLOP/s over the kernel time = {100 * tf / 157.3:.1f} % of the '
          f'157.3 TFLOP/s peak; matrix pipe busy >= {100 * tot["SQ_VALU_MFMA_BUSY_CYCLES"] / (ttot * 2.4 * 1024):.1f} % of the SIMD-cycles; '
          f'wave-cycles waiting {100 * tot["SQ_WAIT_ANY"] / max(tot["SQ_WAVE_CYCLES"], 1):.1f} %\n')
    print('| kernel | calls/pair | ms/pair | TFLOP/s f32 on the MFMA | % of 157.3 peak | mfma busy % (>=) | VALU active % | wait % |')
    print('|---|---|---|---|---|---|---|---|')
    for name in sorted(k, key=lambda n: -disp.get(n, (0, 0))[1])[:28]:
        d = k[name]
        short = re.sub(r'\((?:[^()]|\([^()]*\))*\)$', '', name.replace('(anonymous namespace)::', '').replace('void ', ''))[:70]
        wc = max(d.get('SQ_WAVE_CYCLES', 0), 1)
        calls, ns = disp.get(name, (0, 1))
        t = d.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0) * 512 / max(ns, 1) / 1e3
        print(f'| `{short}` | {calls / pairs:.1f} | {ns / pairs / 1e6:.3f} | {t:.1f} | {100 * t / 157.3:.1f} | '
              f'{100 * d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (max(ns, 1) * 2.4 * 1024):.1f} | {100 * d.get("SQ_ACTIVE_INST_VALU", 0) / wc:.1f} | '
              f'{100 * d.get("SQ_WAIT_ANY", 0) / wc:.1f} |')


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0)
