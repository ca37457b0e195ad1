"""CPU: pins oracle/forward.py against golden vectors captured from the reference's own forward
(tests/golden/gen_golden.py).  The end-to-end replay tolerates the fp32 noise the reference shows
against itself (tests/golden/oracle_vs_reference.json: 8-thread vs 1-thread runs); discrete outputs
must match exactly on the full bundled pair."""
import json
import os

import numpy as np
import pytest
import torch

from sampling import compact_scores, expand_scores, sample


@pytest.fixture(scope='module')
def setup(oracle_native):
    from oracle import forward as ofw
    from rdmnet_amd import config, weights
    cfg = config.make_cfg()
    W = ofw.to_torch(weights.synthetic_state_dict(cfg, seed=0))
    return ofw, cfg, W


def load(golden_dir, tag):
    return np.load(os.path.join(golden_dir, f'forward_{tag}.npz'))


def close(a, b, rtol=2e-5, atol=2e-5):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = max(1.0, float(np.abs(b).max())) if b.size else 1.0
    err = np.abs(a - b).max() if a.size else 0.0
    assert err <= atol * scale + rtol * scale, f'max err {err} (scale {scale})'


TAGS = ['small', 'crop9', 'pair04', 'pair07', 'pair04_seed1', 'synth0', 'synth3', 'lowoverlap', 'dense20k']


@pytest.mark.parametrize('tag', TAGS)
def test_forward_replays_reference_goldens(setup, golden_dir, tag):
    ofw, cfg, W = setup
    g = load(golden_dir, tag)
    if int(g['weight_seed']) != 0:
        from rdmnet_amd import weights
        W = ofw.to_torch(weights.synthetic_state_dict(cfg, seed=int(g['weight_seed'])))
    rp, sp = g['ref_points_in'], g['src_points_in']
    data = ofw.pyramid(np.concatenate([rp, sp]), np.array([len(rp), len(sp)], np.int64), cfg)
    for i in range(5):
        assert np.array_equal(data['lengths'][i].numpy(), g[f'lengths{i}'])
    taps = {}
    out = ofw.forward(W, cfg, data, taps)
    for k in g.files:
        if k.startswith('tap/encoder.'):
            close(sample(taps[k[4:]].numpy()), g[k])
    for k in ('t1_ref', 't1_src', 'vote_feats', 'decoder'):
        close(sample(taps[k].numpy()), g['tap/' + k])
    close(taps['vote_xyz'].numpy(), g['tap/vote_xyz'])
    assert np.array_equal(taps['nms_mask'].numpy(), g['tap/nms_mask'])
    for k in ('ref_points_c', 'src_points_c', 'ref_n2p_scores_c', 'src_n2n_scores_c'):
        close(out[k].numpy(), g['out/' + k])
    close(sample(out['ref_feats_f'].numpy()), g['out/ref_feats_f'])
    close(sample(out['src_feats_c'].numpy()), g['out/src_feats_c'])
    # superpoint pairs: the same set on every case (the reference agrees with itself on it, `self/*` entries)
    pairs = set(zip(out['ref_node_corr_indices'].tolist(), out['src_node_corr_indices'].tolist()))
    assert pairs == set(zip(g['out/ref_node_corr_indices'].tolist(), g['out/src_node_corr_indices'].tolist()))
    # point correspondences as a set (their order follows the order of near-tied superpoint scores); the reference's own
    # 8- and 1-thread runs differ by `self/corr_symmetric_difference` rows (2 on synth0, else 0)
    def rows(rc, sc):
        return set(map(tuple, np.concatenate([np.asarray(rc), np.asarray(sc)], 1).tolist()))
    diff = rows(out['ref_corr_points'], out['src_corr_points']) ^ rows(g['out/ref_corr_points'], g['out/src_corr_points'])
    assert len(diff) <= int(g['self/corr_symmetric_difference'])
    if tag == 'pair04':  # here the ORDER is stable too
        assert np.array_equal(out['ref_node_corr_indices'].numpy(), g['out/ref_node_corr_indices'])
        assert np.array_equal(out['src_node_corr_indices'].numpy(), g['out/src_node_corr_indices'])
        assert np.array_equal(out['ref_corr_points'].numpy(), g['out/ref_corr_points'])
        assert np.array_equal(out['src_corr_points'].numpy(), g['out/src_corr_points'])
        close(out['corr_scores'].numpy(), g['out/corr_scores'], atol=1e-4)
        ms = compact_scores(out['matching_scores'].numpy(), out['ref_node_corr_knn_masks'].numpy(),
                            out['src_node_corr_knn_masks'].numpy())
        close(ms, g['out/matching_scores'], rtol=1e-6, atol=1e-6)
    # pose, tie-aware: one of the poses the reference returns from the hypotheses within one inlier of its best
    # (gen_golden.py `lgr/alt_transforms`; on 0<->7 the reference's own 8- and 1-thread runs pick different ones)
    # -- except where the reference's pose is not reproducible by the reference itself (0<->7: a handful of inliers at
    # the 0.6 m acceptance radius; its two runs return poses 121 deg apart): there the stage-wise pins below are the bar
    errs = [ofw.rre_rte(out['estimated_transform'].numpy(), A) for A in g['lgr/alt_transforms']]
    rre, rte = min(errs)
    self_rre, self_rte = ofw.rre_rte(g['self/transform_1_thread'], g['out/estimated_transform'])
    assert (self_rre <= 1e-3 and self_rte <= 1e-4) == (tag != 'pair07')
    if tag != 'pair07':
        assert rre < 1e-3 and rte < 1e-4, (errs,)


@pytest.mark.parametrize('tag', TAGS)
def test_lgr_stage_teacher_forced_is_exact(setup, golden_dir, tag):
    """Fed the reference's own Sinkhorn output, the restated LGR returns the reference's
    correspondences bit-exactly and its pose within RRE <= 1e-3 deg, RTE <= 1e-3 cm."""
    ofw, cfg, W = setup
    g = load(golden_dir, tag)
    rm, sm = g['out/ref_node_corr_knn_masks'], g['out/src_node_corr_knn_masks']
    ms = expand_scores(g['out/matching_scores'], rm, sm)
    rc, sc, cs, T, info = ofw.lgr(torch.from_numpy(g['out/ref_node_corr_knn_points']),
                                  torch.from_numpy(g['out/src_node_corr_knn_points']), torch.from_numpy(rm),
                                  torch.from_numpy(sm), torch.from_numpy(ms), cfg)
    assert np.array_equal(rc.numpy(), g['out/ref_corr_points'])
    assert np.array_equal(sc.numpy(), g['out/src_corr_points'])
    assert np.array_equal(cs.numpy(), g['out/corr_scores'])
    rre, rte = ofw.rre_rte(T.numpy(), g['out/estimated_transform'])
    assert rre <= 1e-3 and rte <= 1e-5, (rre, rte)
    # the recorded near-tie alternatives are what the restatement returns when forced onto those hypotheses
    assert int(info['best']) == int(g['lgr/best']) and np.array_equal(info['inlier_counts'].numpy(), g['lgr/inlier_counts'])
    for i, A in zip(g['lgr/alt_hypotheses'], g['lgr/alt_transforms']):
        Ti = ofw.lgr(torch.from_numpy(g['out/ref_node_corr_knn_points']), torch.from_numpy(g['out/src_node_corr_knn_points']),
                     torch.from_numpy(rm), torch.from_numpy(sm), torch.from_numpy(ms), cfg, force_best=int(i))[3]
        assert ofw.rre_rte(Ti.numpy(), A) <= (1e-3, 1e-5)


def test_generation_report_shows_stage_pins(golden_dir):
    rep = json.load(open(os.path.join(golden_dir, 'oracle_vs_reference.json')))
    for tag in TAGS:
        tf = rep[tag]['teacher_forced']
        assert tf['coarse/indices_equal'] and tf['lgr/corr_points_equal']
        assert tf['point_to_node/ref_masks_equal'] and tf['point_to_node/src_knn_points_equal']
        assert tf['sinkhorn/matching_scores']['max_abs'] == 0.0
        assert rep[tag]['tap/nms_mask_equal'] and rep[tag]['pyramid/points']
