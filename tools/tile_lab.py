"""Phase timing of kpconv_tile_kernel (s_memtime stamps of thread 0 of every workgroup, in units of 100 ticks of that counter; a phase ends at a workgroup barrier, so it includes the wait for the slowest wavefront):
    make -C rdmnet_amd/csrc timing && RDM_LIB_PATH=$PWD/rdmnet_amd/librdmnet_hip_timing.so python tools/tile_lab.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rdmnet_amd import _lib, config, engine, ops, weights

z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'synthetic_pairs.npz'))
cfg = config.make_cfg()
eng = engine.Engine(cfg, weights.synthetic_state_dict(cfg, seed=0))
dd = eng.collate(torch.from_numpy(z['ref0']).cuda(), torch.from_numpy(z['src0']).cuda())
L = _lib.lib()
handle = ctypes.CDLL(_lib.LIB_PATH)
handle.rdm_dbg_tile_timing.argtypes = [ctypes.c_void_p]
handle.rdm_dbg_tile_wave_timing.argtypes = [ctypes.c_void_p]
g = torch.Generator().manual_seed(0)
kp = (torch.randn(15, 3, generator=g) * 0.3).cuda()
names = ['idx issue + hash clear', 'hash inserts', 'slot numbering', 'tile load', 'aggregation', 'park + contraction', 'epilogue']
for ql, sl, key, c in [(0, 0, 'neighbors', 32), (1, 0, 'subsampling', 32), (1, 1, 'neighbors', 64), (2, 1, 'subsampling', 64)]:
    q, s = dd['points'][ql], dd['points'][sl]
    idx = dd[key][sl if key == 'subsampling' else ql]
    feats = ops.feat_empty(s.shape[0], c, 'cuda'); feats.copy_(torch.randn(s.shape[0], c, generator=g))
    pos = ops.row_positive(feats)
    W = (torch.randn(15, c, c, generator=g) / np.sqrt(15 * c)).numpy()
    packed = torch.from_numpy(ops.kpconv_pack_weights(W)).cuda()
    bias = torch.randn(c, generator=g).cuda()
    rec = ops.radius_grid_records(q, dd['lengths'][ql], cfg.backbone.init_radius * 2 ** ql)
    nblk = (idx.shape[0] + 15) // 16
    buf = torch.zeros((nblk, 8), dtype=torch.int64, device='cuda')
    wbuf = torch.zeros((nblk, 16, 8), dtype=torch.int64, device='cuda')
    for order, tag in ((None, 'row order'), (rec, 'cell order')):
        handle.rdm_dbg_tile_timing(None)
        for _ in range(3):
            ops.kpconv_fused(q, s, feats, pos, idx, kp, 0.6 * 2 ** sl, packed, bias, c, want_partials=True, form=2, order=order)
        handle.rdm_dbg_tile_timing(buf.data_ptr())
        handle.rdm_dbg_tile_wave_timing(wbuf.data_ptr())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.kpconv_fused(q, s, feats, pos, idx, kp, 0.6 * 2 ** sl, packed, bias, c, want_partials=True, form=2, order=order); e1.record()
        torch.cuda.synchronize()
        t = buf.cpu().numpy().astype(np.float64)
        raw7 = buf.cpu().numpy()[:, 7]
        slots = (raw7 & 0xffff).astype(np.float64)
        rt_ns = (raw7 >> 16).astype(np.float64) * 10.0  # wall_clock64: 100 MHz
        uniq = []
        print(f'--- C={c} L{sl}->L{ql} M={idx.shape[0]} H={idx.shape[1]} {tag}: kernel {e0.elapsed_time(e1) * 1e3:.1f} us; workgroups {nblk}; '
              f'distinct rows per workgroup mean {slots.mean():.0f} p90 {np.percentile(slots, 90):.0f} max {slots.max():.0f} '
              f'(real (query, neighbour) pairs per workgroup {float((idx < s.shape[0]).sum()) / nblk:.0f})')
        print('    phase, x100 ticks (mean over workgroups / p90): ' + '; '.join(f'{n} {t[:, k].mean() / 100:.2f} / {np.percentile(t[:, k], 90) / 100:.2f}' for k, n in enumerate(names))
              + f'; total {t[:, :7].sum(1).mean() / 100:.2f} = {rt_ns.mean() / 1e3:.2f} us per workgroup on the 100 MHz clock -> '
                f'{t[:, :7].sum(1).mean() / rt_ns.mean():.2f} ticks per ns')
        w = wbuf.cpu().numpy().astype(np.float64)
        nw = 16 if c == 32 else 8
        w = w[:, :nw]
        trips = np.ceil(w[..., 2] / 16)
        ok = trips > 0
        print(f'    per wavefront, first query: prologue (flags, positive count) {w[..., 0][ok].mean():.0f} ticks; neighbour loop {w[..., 1][ok].mean():.0f} ticks for '
              f'{trips[ok].mean():.2f} trips of 16 neighbours = {(w[..., 1][ok] / trips[ok]).mean():.0f} per trip; queries on the global fall-back path {w[..., 3][ok].mean():.3f}')
        if w[..., 4:].sum() > 0:  # build with TIMING_EXTRA=-DRDM_TILE_TRIP_STAMPS: serialised stages of a trip
            print('    serialised stages per trip (ticks): ' + '; '.join(f'{n} {(w[..., 4 + k][ok] / trips[ok]).mean():.0f}' for k, n in
                  enumerate(['slot codes (LDS)', 'points + feature rows (LDS)', 'influence (VALU)', 'MFMA chain'])))
