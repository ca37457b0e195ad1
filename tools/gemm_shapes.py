"""The dense products of ONE scan pair on rdm_gemm's tiled kernels (gemm_kernel / gemm_small_kernel), per shape: the tile and
split-K factor the dispatch model picks, microseconds with the GPU to itself (HIP-graph replay) and with three other pairs
in flight (three background engines on their own streams), TFLOP/s against the 157.3 TFLOP/s fp32 MFMA peak -- and, joined
from a rocprofv3 --pmc pass of `--pmc-pass` (plain launches in the same order), matrix-pipe busy and wait shares.
    python tools/gemm_shapes.py --out gpurun_out/gemm_shapes.json          # timings
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 -d D -- python tools/gemm_shapes.py --pmc-pass
    python tools/gemm_shapes.py --join gpurun_out/gemm_shapes.json D/.../x.db > profiles/r03_gemm_shapes.md
Shapes: pair 0 of the bench workload (level sizes 32000 / 10961 / 3879 / 1310 / 563); the KPConv contractions with
c_in <= 64 run inside kpconv_fused* since round 3 and are not GEMM launches any more."""
import argparse, ctypes, json, os, sqlite3, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PEAK = 157.3
N = [32000, 10961, 3879, 1310, 563]
# (name, where it runs, M, K, N, rowdiv)
SHAPES = [('KPConv 128->128 L2', N[2], 1920, 128, True), ('KPConv 128->128 L2->L3', N[3], 1920, 128, True),
          ('KPConv 256->256 L3', N[3], 3840, 256, True), ('KPConv 256->256 L3->L4', N[4], 3840, 256, True),
          ('KPConv 512->512 L4', N[4], 7680, 512, True),
          ('unary 64->32 L0', N[0], 64, 32, False), ('unary 32->128 L0', N[0], 32, 128, False), ('shortcut 64->128 L0', N[0], 64, 128, False),
          ('unary 128->32 L0', N[0], 128, 32, False), ('unary 32->128 L1', N[1], 32, 128, False), ('unary 128->64 L1', N[1], 128, 64, False),
          ('unary 64->256 L1', N[1], 64, 256, False), ('shortcut 128->256 L1', N[1], 128, 256, False), ('unary 256->64 L1', N[1], 256, 64, False),
          ('unary 256->128 L2', N[2], 256, 128, False), ('unary 128->512 L2', N[2], 128, 512, False), ('shortcut 256->512 L2', N[2], 256, 512, False),
          ('unary 512->128 L2', N[2], 512, 128, False), ('unary 512->256 L3', N[3], 512, 256, False), ('unary 256->1024 L3', N[3], 256, 1024, False),
          ('shortcut 512->1024 L3', N[3], 512, 1024, False), ('unary 1024->256 L3', N[3], 1024, 256, False), ('unary 1024->512 L4', N[4], 1024, 512, False),
          ('unary 512->2048 L4', N[4], 512, 2048, False), ('shortcut 1024->2048 L4', N[4], 1024, 2048, False), ('unary 2048->512 L4', N[4], 2048, 512, False),
          ('decoder4 1281->1024 L3', N[3], 1284, 1024, False), ('decoder3 1536->512 L2', N[2], 1536, 512, False), ('decoder2 768->257 L1', N[1], 768, 257, False),
          ('in_proj 2048->128 L4', N[4], 2048, 128, False), ('qkv 128->384 L4', N[4], 128, 384, False), ('out_proj 128->256 L4', N[4], 128, 256, False)]
# launches of each shape per pair (encoder blocks repeat: backbone.py:27-70; transformer products: 2 x 4 layers)
PER_PAIR = {'KPConv 128->128 L2': 2, 'KPConv 256->256 L3': 2, 'KPConv 512->512 L4': 2, 'unary 128->32 L0': 1, 'unary 32->128 L0': 1,
            'unary 256->64 L1': 2, 'unary 64->256 L1': 3, 'unary 512->128 L2': 2, 'unary 128->512 L2': 3, 'unary 1024->256 L3': 2, 'unary 256->1024 L3': 3,
            'unary 2048->512 L4': 2, 'unary 512->2048 L4': 3, 'in_proj 2048->128 L4': 1, 'qkv 128->384 L4': 8, 'out_proj 128->256 L4': 2}


def operands(torch, m, k, n, rd):
    a = torch.randn(m, k, device='cuda')
    b = torch.randn(k, (n + 3) // 4 * 4, device='cuda')
    return a, b, (torch.ones(m, device='cuda') if rd else None), (None if rd else torch.randn(n, device='cuda'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'gemm_shapes.json'))
    ap.add_argument('--pmc-pass', action='store_true')
    ap.add_argument('--join', nargs=2, metavar=('JSON', 'DB'))
    args = ap.parse_args()
    if args.join:
        return join(*args.join)
    import numpy as np
    import torch
    from rdmnet_amd import _lib, config, engine, ops, weights
    L = _lib.lib()

    def plan():
        out = (ctypes.c_int * 4)()
        L.rdm_gemm_last_plan(ctypes.addressof(out))
        return list(out)
    if args.pmc_pass:  # four plain launches per shape, in SHAPES order (the join segments the trace by this order)
        for name, m, k, n, rd in SHAPES:
            a, b, rowdiv, bias = operands(torch, m, k, n, rd)
            for _ in range(4):
                ops.gemm(a, b, k, n, rowdiv=rowdiv, bias=bias)
            torch.cuda.synchronize()
        return

    def timed(fn, reps=20):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s, capture_error_mode='thread_local'):
                for _ in range(reps):
                    fn()
            g.replay(); torch.cuda.synchronize()

        def measure():
            with torch.cuda.stream(s):  # (replay launches on the current stream)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(s)
                for _ in range(5):
                    g.replay()
                e1.record(s); e1.synchronize()
            return e0.elapsed_time(e1) / (5 * reps) * 1e3
        return measure

    # three pairs in flight beside the measured product
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'synthetic_pairs.npz'))
    cfg = config.make_cfg()
    state = weights.synthetic_state_dict(cfg, seed=0)
    pairs = [(torch.from_numpy(z[f'ref{i}']).cuda(), torch.from_numpy(z[f'src{i}']).cuda()) for i in (0, 1)]
    engines = [engine.Engine(cfg, state) for _ in range(3)]
    stop, go = threading.Event(), threading.Event()

    idle = [threading.Event() for _ in engines]

    def background(eng, k):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            while not stop.is_set():
                if go.is_set():
                    idle[k].clear()
                    eng.run(*pairs[k % 2])
                else:
                    idle[k].set()
                    go.wait(0.01)
    threads = [threading.Thread(target=background, args=(e, k)) for k, e in enumerate(engines)]
    for t in threads:
        t.start()
    rows = []
    try:
        for name, m, k, n, rd in SHAPES:
            a, b, rowdiv, bias = operands(torch, m, k, n, rd)
            ops.gemm(a, b, k, n, rowdiv=rowdiv, bias=bias)
            p = plan()
            measure = timed(lambda: ops.gemm(a, b, k, n, rowdiv=rowdiv, bias=bias))
            alone = min(measure() for _ in range(3))
            go.set()
            import time
            time.sleep(0.05)
            shared = float(np.median([measure() for _ in range(7)]))
            go.clear()
            for ev in idle:  # (a stream capture must not meet another thread's synchronisation)
                ev.clear()
            for ev in idle:
                ev.wait()
            torch.cuda.synchronize()
            fl = 2.0 * m * k * n
            rows.append({'name': name, 'm': m, 'k': k, 'n': n, 'tile': f'{p[0]}x{p[1]}x{p[2]}', 'split': p[3], 'us_alone': alone,
                         'us_shared': shared, 'tf_alone': fl / alone / 1e6, 'tf_shared': fl / shared / 1e6, 'per_pair': PER_PAIR.get(name, 1)})
            print(f'{name:28s} M={m:6d} K={k:5d} N={n:5d} {rows[-1]["tile"]:>10s} s{p[3]:<2d} alone {alone:7.1f} us {rows[-1]["tf_alone"]:6.1f} TF | '
                  f'3 pairs beside {shared:7.1f} us {rows[-1]["tf_shared"]:6.1f} TF', flush=True)
    finally:
        stop.set(); go.set()
        for t in threads:
            t.join()
    json.dump(rows, open(args.out, 'w'), indent=1)


def join(json_path, db):
    rows = json.load(open(json_path))
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute('pragma table_info(pmc_events)').fetchall()]
    kcol = 'name' if 'name' in cols else 'kernel_name'
    order = 'start' if 'start' in cols else ('dispatch_id' if 'dispatch_id' in cols else 'rowid')
    disp_col = next((c for c in ('dispatch_id', 'event_id', 'id') if c in cols), None)
    ev = cur.execute(f"select {kcol}, counter_name, counter_value, {order}, {disp_col or order}, duration from pmc_events where {kcol} like '%gemm_%' or {kcol} like '%splitk%' "
                     f'order by {order}').fetchall()
    # dispatches in order: key = (order value, dispatch id); counters summed over hardware instances
    disp, seq = {}, []
    for name, c, v, o, d, dur in ev:
        key = (o, d)
        if key not in disp:
            disp[key] = {'name': name, 'ns': dur}
            seq.append(key)
        disp[key][c] = disp[key].get(c, 0) + v
    main_disp = [disp[k] for k in seq if 'splitk' not in disp[k]['name']]  # the tiled / small kernel launches: 4 per shape
    assert len(main_disp) == 4 * len(rows), (len(main_disp), len(rows), cols)
    print('| product (pair 0) | M | K | N | x/pair | tile | split-K | us alone | TFLOP/s alone | of 157.3 | us, 3 pairs beside | TFLOP/s | MFMA busy % | waiting % | lost us/pair vs peak |')
    print('|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|')
    out, tot_fl, tot_alone, tot_shared = [], 0.0, 0.0, 0.0
    for i, r in enumerate(rows):
        d = main_disp[4 * i + 1:4 * i + 4]  # (the first launch of a shape pays cold caches)
        busy = sum(x.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) for x in d) / len(d)
        ns = sum(x['ns'] for x in d) / len(d)  # the kernel's duration in the counter pass
        wc = sum(x.get('SQ_WAVE_CYCLES', 0) for x in d) / len(d)
        wait = sum(x.get('SQ_WAIT_ANY', 0) for x in d) / len(d)
        fl = 2.0 * r['m'] * r['k'] * r['n']
        lost = (r['us_shared'] - fl / PEAK / 1e6) * r['per_pair']
        tot_fl += fl * r['per_pair']; tot_alone += r['us_alone'] * r['per_pair']; tot_shared += r['us_shared'] * r['per_pair']
        out.append((lost, f"| {r['name']} | {r['m']} | {r['k']} | {r['n']} | {r['per_pair']} | {r['tile']} | {r['split']} | {r['us_alone']:.1f} | {r['tf_alone']:.1f} | "
                          f"{r['tf_alone'] / PEAK:.2f} | {r['us_shared']:.1f} | {r['tf_shared']:.1f} | {100 * busy / (ns * 2.4 * 1024):.0f} | "
                          f"{100 * wait / max(wc, 1):.0f} | {lost:.0f} |"))
    for _, line in sorted(out, key=lambda t: -t[0]):
        print(line)
    print(f'\nAll listed products of a pair: {tot_fl / 1e9:.1f} GFLOP, {tot_alone:.0f} us alone = {tot_fl / tot_alone / 1e6:.1f} TFLOP/s '
          f'({tot_fl / tot_alone / 1e6 / PEAK:.2f} of the fp32 MFMA peak), {tot_shared:.0f} us with three pairs beside = {tot_fl / tot_shared / 1e6:.1f} TFLOP/s '
          f'({tot_fl / tot_shared / 1e6 / PEAK:.2f}).  Rows ranked by the time a pair loses against the peak in the shared regime.')


if __name__ == '__main__':
    main()
