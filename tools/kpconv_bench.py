"""Graph-replayed timings of the KPConv neighbourhood kernels on the real pyramid of a bench pair: the one-kernel form
(rdm_kpconv_fused, c_in = 1 / 32 / 64) against gather + weight GEMM, and the stand-alone gather of the deeper levels;
GB/s are SURVEY 8d bytes M*H*(8 + 12 + 4*C_in) (padded slots) over the kernel time.   python tools/kpconv_bench.py [pair]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rdmnet_amd import config, engine, ops, weights


def timed(fn, reps=10):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(5):
            g.replay()
        e1.record(s); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * reps) * 1e3


if __name__ == '__main__':
    pid = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'synthetic_pairs.npz'))
    cfg = config.make_cfg()
    eng = engine.Engine(cfg, weights.synthetic_state_dict(cfg, seed=0))
    dd = eng.collate(torch.from_numpy(z[f'ref{pid}']).cuda(), torch.from_numpy(z[f'src{pid}']).cuda())
    g = torch.Generator().manual_seed(0)
    kp = (torch.randn(15, 3, generator=g) * 0.3).cuda()
    # (query level, support level, table, c_in, c_out) of the 14 layers' distinct shapes
    cases = [(0, 0, 'neighbors', 1, 64), (0, 0, 'neighbors', 32, 32), (1, 0, 'subsampling', 32, 32), (1, 1, 'neighbors', 64, 64),
             (2, 1, 'subsampling', 64, 64), (2, 2, 'neighbors', 128, 128), (3, 2, 'subsampling', 128, 128),
             (3, 3, 'neighbors', 256, 256), (4, 3, 'subsampling', 256, 256), (4, 4, 'neighbors', 512, 512)]
    print('| layer | M | H | real fill | kernel | us | GB/s (padded) | of 8 TB/s |')
    print('|---|---|---|---|---|---|---|---|')
    for ql, sl, key, cin, cout in cases:
        q, s = dd['points'][ql], dd['points'][sl]
        idx = dd[key][sl if key == 'subsampling' else ql]
        M, H = idx.shape
        fill = float((idx < s.shape[0]).float().mean())
        feats = ops.feat_empty(s.shape[0], cin, 'cuda')
        feats.copy_(torch.ones(s.shape[0], 1) if cin == 1 else torch.randn(s.shape[0], cin, generator=g))
        pos = ops.row_positive(feats)
        sigma = cfg.backbone.init_sigma * 2 ** sl
        nbytes = M * H * (8 + 12 + 4 * cin)
        W = (torch.randn(15, cin, cout, generator=g) / np.sqrt(15 * cin)).numpy()
        bias = torch.randn(cout, generator=g).cuda()
        rows = []
        if ops.kpconv_fused_supported(cin, cout):
            packed = torch.from_numpy(ops.kpconv_pack_weights(W)).cuda()
            t = timed(lambda: ops.kpconv_fused(q, s, feats, pos, idx, kp, sigma, packed, bias, cout, want_partials=True, form=1))
            rows.append(('one kernel, lock-step (r03)', t))
            rec = ops.radius_grid_records(q, dd['lengths'][ql], cfg.backbone.init_radius * 2 ** ql)
            if cin > 1:
                t = timed(lambda: ops.kpconv_fused(q, s, feats, pos, idx, kp, sigma, packed, bias, cout, want_partials=True, form=2))
                rows.append(('one kernel, LDS tile, row order', t))
                t = timed(lambda: ops.kpconv_fused(q, s, feats, pos, idx, kp, sigma, packed, bias, cout, want_partials=True, form=2, order=rec))
                rows.append(('one kernel, LDS tile, cell order', t))
                for f, tag in ((3, 'block ids in dispatch order (round 4)'),):
                    t = timed(lambda: ops.kpconv_fused(q, s, feats, pos, idx, kp, sigma, packed, bias, cout, want_partials=True, form=f, order=rec))
                    rows.append((f'  lab form {f}: {tag}', t))
        rec_q = ops.radius_grid_records(q, dd['lengths'][ql], cfg.backbone.init_radius * 2 ** ql)
        t = timed(lambda: ops.kpconv_gather(q, s, feats, pos, idx, kp, sigma, order=rec_q, form=1))
        rows.append(('gather alone (one wavefront per query and slice, cell order)', t))
        if cin >= 128:
            t = timed(lambda: ops.kpconv_gather(q, s, feats, pos, idx, kp, sigma, order=rec_q, form=2))
            rows.append(('gather alone, LDS tile (round 5)', t))
        for name, t in rows:
            print(f'| L{sl}->L{ql} {cin}->{cout} | {M} | {H} | {fill:.2f} | {name} | {t:.1f} | {nbytes / t / 1e3:.0f} | {nbytes / t / 1e3 / 8000:.2f} |')
