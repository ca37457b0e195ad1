"""Developer probe (GPU box): how close the bf16-attention path (BASELINE configs[3]) lands to the oracle's bf16
restatement and to the fp32 path on the discrete outputs -- the numbers behind the assertions of
tests/test_full_size_configs_gpu.py::test_config3_bf16_attention_full_size."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import forward as ofw  # noqa: E402
from rdmnet_amd import collate, config, model, weights  # noqa: E402

z = np.load(os.path.join(ROOT, 'tests', 'golden', 'synthetic_pairs.npz'))
for pid in (0, 1):
    ref, src = z[f'ref{pid}'], z[f'src{pid}']
    res = {}
    for bf16 in (False, True):
        cfg = config.make_cfg()
        cfg.thdroformer.attention_bf16 = bf16
        state = weights.synthetic_state_dict(cfg, seed=0)
        odata = ofw.pyramid(np.concatenate([ref, src]), np.array([len(ref), len(src)], np.int64), cfg)
        otaps = {}
        oout = ofw.forward(ofw.to_torch(state), cfg, odata, otaps)
        net = model.create_model(cfg).cuda()
        net.load_state_dict(state)
        taps = {}
        out = net(collate.collate_pair(ref, src, cfg, exact_shapes=True), taps)
        res[bf16] = (oout, otaps, out, taps)
        hp = set(zip(out['ref_node_corr_indices'].tolist(), out['src_node_corr_indices'].tolist()))
        op = set(zip(oout['ref_node_corr_indices'].tolist(), oout['src_node_corr_indices'].tolist()))
        nms_eq = int((taps['nms_mask'].cpu().bool() != otaps['nms_mask']).sum())
        rre, rte = ofw.rre_rte(out['estimated_transform'].cpu().double().numpy(), oout['estimated_transform'].numpy())
        hc = {tuple(np.round(r, 5)) for r in torch.cat([out['ref_corr_points'], out['src_corr_points']], 1).cpu().numpy().tolist()}
        oc = {tuple(np.round(r, 5)) for r in torch.cat([oout['ref_corr_points'], oout['src_corr_points']], 1).numpy().tolist()}
        print(f'pair {pid} bf16={bf16}: nms mask differs at {nms_eq} of {len(otaps["nms_mask"])}; superpoint pairs common {len(hp & op)}/256; '
              f'correspondences hip {len(hc)} oracle {len(oc)} common {len(hc & oc)}; pose vs oracle rre {rre:.3e} deg rte {rte:.3e} m')
        for k in ('t1_ref', 't2_ref', 'vote_xyz', 'vote_feats', 'decoder'):
            a, b = taps[k].cpu().double(), otaps[k].double()
            print(f'    {k}: rel {((a - b).abs().max() / b.abs().max()).item():.3e}')
    T32, T16 = (res[b][2]['estimated_transform'].cpu().double().numpy() for b in (False, True))
    print(f'pair {pid}: hip bf16 pose vs hip fp32 pose: rre %.3e deg rte %.3e m' % ofw.rre_rte(T16, T32))
    print(f'pair {pid}: oracle bf16 pose vs oracle fp32 pose: rre %.3e deg rte %.3e m' % ofw.rre_rte(res[True][0]['estimated_transform'].numpy(), res[False][0]['estimated_transform'].numpy()))
