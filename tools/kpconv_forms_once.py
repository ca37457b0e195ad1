"""One shape of the KPConv layer table through both one-kernel forms (lock-step = form 1, LDS tile in cell order = form 2), a few
launches each -- the workload of tools/pmc_kpconv_forms.sh (rocprofv3 --pmc attributes counters per kernel name, and two layers
share a kernel: one shape per process).   python tools/kpconv_forms_once.py <shape 0..3> [launches]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rdmnet_amd import config, engine, ops, weights

shape, reps = int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 5
z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'synthetic_pairs.npz'))
cfg = config.make_cfg()
eng = engine.Engine(cfg, weights.synthetic_state_dict(cfg, seed=0))
dd = eng.collate(torch.from_numpy(z['ref0']).cuda(), torch.from_numpy(z['src0']).cuda())
g = torch.Generator().manual_seed(0)
kp = (torch.randn(15, 3, generator=g) * 0.3).cuda()
ql, sl, key, c = [(0, 0, 'neighbors', 32), (1, 0, 'subsampling', 32), (1, 1, 'neighbors', 64), (2, 1, 'subsampling', 64)][shape]
q, s = dd['points'][ql], dd['points'][sl]
idx = dd[key][sl if key == 'subsampling' else ql]
feats = ops.feat_empty(s.shape[0], c, 'cuda'); feats.copy_(torch.randn(s.shape[0], c, generator=g))
pos = ops.row_positive(feats)
W = (torch.randn(15, c, c, generator=g) / np.sqrt(15 * c)).numpy()
packed = torch.from_numpy(ops.kpconv_pack_weights(W)).cuda()
bias = torch.randn(c, generator=g).cuda()
rec = ops.radius_grid_records(q, dd['lengths'][ql], cfg.backbone.init_radius * 2 ** ql)
torch.cuda.synchronize()
for _ in range(reps):
    ops.kpconv_fused(q, s, feats, pos, idx, kp, 0.6 * 2 ** sl, packed, bias, c, want_partials=True, form=1)
    torch.cuda.synchronize()
    ops.kpconv_fused(q, s, feats, pos, idx, kp, 0.6 * 2 ** sl, packed, bias, c, want_partials=True, form=2, order=rec)
    torch.cuda.synchronize()
print(f'shape {shape}: L{sl}->L{ql} C={c} M={idx.shape[0]} H={idx.shape[1]} real pairs {int((idx < s.shape[0]).sum())} bytes/row {4 * c}')
