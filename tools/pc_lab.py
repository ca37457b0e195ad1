"""Phase timing of kpconv_fused_pc_kernel (shader-clock sums of workgroup 0's sixteen wavefronts): builds of kpconv_fused.hip
with -DRDM_PC_TIMING export rdm_dbg_pc_timing(buffer).   RDM_LIB_PATH=.../librdmnet_hip_pctiming.so python tools/pc_lab.py
The kernel is the lab copy tools/lab/kpconv_fused_pc.hip (docs/EXPERIMENTS.md 5d: measured a wash, not in the library): to reproduce, put it
in place of rdmnet_amd/csrc/kpconv_fused.hip, compile that file with -DRDM_PC_TIMING and link it with the other objects of
rdmnet_amd/csrc/build into librdmnet_hip_pctiming.so."""
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
handle.rdm_dbg_pc_timing.argtypes = [ctypes.c_void_p]
buf = torch.zeros((16, 8), dtype=torch.int64, device='cuda')
handle.rdm_dbg_pc_timing(buf.data_ptr())
g = torch.Generator().manual_seed(0)
kp = (torch.randn(15, 3, generator=g) * 0.3).cuda()
for ql, sl, key, c in [(0, 0, 'neighbors', 32), (1, 1, 'neighbors', 64), (2, 1, 'subsampling', 64)]:
    q, s = dd['points'][ql], dd['points'][sl]
    idx = dd[key][sl if key == 'subsampling' else ql]
    feats = ops.feat_empty(s.shape[0], c, 'cuda'); feats.copy_(torch.randn(s.shape[0], c, generator=g))
    pos = ops.row_positive(feats)
    W = (torch.randn(15, c, c, generator=g) / np.sqrt(15 * c)).numpy()
    packed = torch.from_numpy(ops.kpconv_pack_weights(W)).cuda()
    bias = torch.randn(c, generator=g).cuda()
    for _ in range(3):
        ops.kpconv_fused(q, s, feats, pos, idx, kp, 0.6 * 2 ** sl, packed, bias, c, want_partials=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.kpconv_fused(q, s, feats, pos, idx, kp, 0.6 * 2 ** sl, packed, bias, c, want_partials=True); e1.record(); torch.cuda.synchronize()
    t = buf.cpu().numpy().astype(np.float64)
    npr = 14 if c == 32 else 12
    print(f'--- C={c} M={idx.shape[0]} H={idx.shape[1]}: kernel {e0.elapsed_time(e1) * 1e3:.1f} us (events); clock ticks of workgroup 0 (s_memtime, 100 MHz: x10 ns)')
    print('producers: total | issue A/B | wait B + stage | trips+mfma | wait slot | park | queries')
    for w in range(npr):
        print('   ', ' '.join(f'{v:9.0f}' for v in t[w, :7]))
    print('consumers: total | wait block | contraction | epilogue | - | - | blocks')
    for w in range(npr, 16):
        print('   ', ' '.join(f'{v:9.0f}' for v in t[w, :7]))
