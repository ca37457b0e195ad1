import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rdmnet_amd import collate, config, model, weights
tag = sys.argv[1] if len(sys.argv) > 1 else 'synth3'
g = np.load(os.path.join(ROOT, 'tests/golden', f'forward_{tag}.npz'))
cfg = config.make_cfg()
net = model.create_model(cfg).cuda(); net.load_state_dict(weights.synthetic_state_dict(cfg, seed=int(g['weight_seed'])))
data = collate.collate_pair(g['ref_points_in'], g['src_points_in'], cfg, exact_shapes=True)
taps = {}
out = net(data, taps)
npy = lambda x: x.detach().cpu().numpy()
ref_pairs = list(zip(g['out/ref_node_corr_indices'].tolist(), g['out/src_node_corr_indices'].tolist()))
hip_pairs = list(zip(npy(out['ref_node_corr_indices']).tolist(), npy(out['src_node_corr_indices']).tolist()))
print('symdiff', len(set(ref_pairs) ^ set(hip_pairs)))
pos = {p: i for i, p in enumerate(ref_pairs)}
perm = np.array([pos[p] for p in hip_pairs])
for side in ('ref', 'src'):
    a = npy(out[f'{side}_node_corr_knn_points']); b = g[f'out/{side}_node_corr_knn_points'][perm]
    bad = np.nonzero((a != b).any(axis=(1, 2)))[0]
    print(side, 'patches differing', bad)
    for p in bad[:4]:
        rows = np.nonzero((a[p] != b[p]).any(1))[0]
        print('  patch', p, 'rows', rows, 'node', hip_pairs[p])
        print('   hip', a[p][rows][:4]); print('   ref', b[p][rows][:4])
        sa = a[p][np.lexsort(a[p].T)]; sb = b[p][np.lexsort(b[p].T)]
        print('   equal as sets:', np.array_equal(sa, sb))
        node = npy(out[f'{side}_points_c'])[hip_pairs[p][0 if side == 'ref' else 1]].astype(np.float64)
        gn = g[f'out/{side}_points_c'][hip_pairs[p][0 if side == 'ref' else 1]].astype(np.float64)
        print('   node hip', node, 'ref', gn)
        for r in rows[:4]:
            print('   d2 hip-order', ((a[p][r] - node) ** 2).sum(), ((b[p][r] - node) ** 2).sum())
