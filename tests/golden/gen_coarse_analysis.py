"""Writes tests/golden/coarse_order_analysis.json: how fragile the ORDER of the reference's top-256 superpoint
correspondences is (superpoint_matching.py:14-83 restated in oracle/forward.py:coarse_matching, which reproduces the
reference's indices exactly -- tests/golden/oracle_vs_reference.json teacher_forced.coarse/indices_equal).

For both golden cases the restated formula is evaluated on identical features in fp32 (= the reference) and in fp64,
and on features perturbed by 1e-6 relative noise (what any other fp32 implementation of the encoder -- another BLAS,
another thread count, a GPU -- produces).  Runs on the CPU, oracle only; no reference import needed."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
from oracle import forward as ofw  # noqa: E402
from rdmnet_amd import config, weights  # noqa: E402


def dual_scores(rf, sf, dt):
    rf, sf = rf.to(dt), sf.to(dt)
    s = torch.exp(-(2 - 2 * rf @ sf.t()))
    return (s / s.sum(1, keepdim=True)) * (s / s.sum(0, keepdim=True))


def main():
    cfg = config.make_cfg()
    W = ofw.to_torch(weights.synthetic_state_dict(cfg, seed=0))
    report = {}
    for tag in ('small', 'pair04'):
        g = np.load(os.path.join(HERE, f'forward_{tag}.npz'))
        rp, sp = g['ref_points_in'], g['src_points_in']
        data = ofw.pyramid(np.concatenate([rp, sp]), np.array([len(rp), len(sp)], np.int64), cfg)
        taps = {}
        out = ofw.forward(W, cfg, data, taps)
        rf, sf = out['ref_feats_c'][taps['ref_node_masks']], out['src_feats_c'][taps['src_node_masks']]
        k = out['ref_node_corr_indices'].shape[0]
        s32, s64 = dual_scores(rf, sf, torch.float32).reshape(-1), dual_scores(rf, sf, torch.float64).reshape(-1)
        v64, i64 = s64.topk(min(k + 1, s64.numel()))
        i32 = s32.topk(k)[1]
        gaps = ((v64[:-1] - v64[1:]) / v64[:-1]).numpy()
        gen = torch.Generator().manual_seed(0)
        noisy = []
        for _ in range(8):
            a = rf * (1 + 1e-6 * torch.randn(rf.shape, generator=gen))
            b = sf * (1 + 1e-6 * torch.randn(sf.shape, generator=gen))
            j = dual_scores(a, b, torch.float32).reshape(-1).topk(k)[1]
            noisy.append((len(set(j.tolist()) ^ set(i32.tolist())), float((j == i32).float().mean())))
        report[tag] = {
            'k': int(k), 'candidates': int(s32.numel()),
            'fp32_vs_fp64_same_inputs': {'set_symmetric_difference': len(set(i32.tolist()) ^ set(i64[:k].tolist())),
                                         'same_position_fraction': float((i32 == i64[:k]).float().mean())},
            'relative_gap_between_neighbouring_scores': {'min': float(gaps[:k - 1].min()) if k > 1 else None,
                                                         'median': float(np.median(gaps[:k - 1])) if k > 1 else None,
                                                         'below_1e-6': int((gaps[:k - 1] < 1e-6).sum()),
                                                         'below_1e-5': int((gaps[:k - 1] < 1e-5).sum()),
                                                         'at_the_cut': float(gaps[k - 1]) if len(gaps) >= k else None},
            'features_perturbed_by_1e-6_relative(8 draws)': {'set_symmetric_difference_max': max(n[0] for n in noisy),
                                                             'same_position_fraction_min': min(n[1] for n in noisy)}}
        print(tag, json.dumps(report[tag]))
    with open(os.path.join(HERE, 'coarse_order_analysis.json'), 'w') as f:
        json.dump(report, f, indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
