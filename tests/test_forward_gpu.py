"""GPU parity of the full inference path (HIP) against the oracle.

Strategy: discrete decisions amplify 1-ulp differences, so every stage is checked
TEACHER-FORCED (fed the oracle's exact stage inputs: integer outputs bit-exact, float outputs within
the stated tolerance) and the end-to-end run is checked on the pose and on agreement rates."""
import os

import numpy as np
import pytest
import torch

from sampling import sample

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.numel() == 0:
        return 0.0
    return ((a - b).abs().max() / b.abs().max().clamp(min=1e-30)).item()


@pytest.fixture(scope='module')
def ctx(oracle_native, golden_dir):
    from oracle import forward as ofw
    from rdmnet_amd import collate, config, model, ops, weights
    cfg = config.make_cfg()
    state = weights.synthetic_state_dict(cfg, seed=0)
    W = ofw.to_torch(state)
    net = model.create_model(cfg).cuda()
    net.load_state_dict(state)
    net._prepare()
    g = np.load(os.path.join(golden_dir, 'forward_small.npz'))
    rp, sp = g['ref_points_in'], g['src_points_in']
    odata = ofw.pyramid(np.concatenate([rp, sp]), np.array([len(rp), len(sp)], np.int64), cfg)
    otaps = {}
    oout = ofw.forward(W, cfg, odata, otaps)
    return dict(ofw=ofw, cfg=cfg, W=W, net=net, rp=rp, sp=sp, odata=odata, otaps=otaps, oout=oout, ops=ops,
                collate=collate, golden=g)


def test_collate_matches_oracle_pyramid_bit_exact(ctx):
    d = ctx['collate'].collate_pair(ctx['rp'], ctx['sp'], ctx['cfg'], exact_shapes=True)
    o = ctx['odata']
    for key in ('points', 'lengths', 'neighbors', 'subsampling', 'upsampling'):
        for a, b in zip(d[key], o[key]):
            assert torch.equal(a.cpu(), b), key
    assert torch.equal(d['features'].cpu(), o['features'])


def test_transformer_stage(ctx):
    net, o, ops = ctx['net'], ctx['otaps'], ctx['ops']
    n_c = int(ctx['odata']['lengths'][-1][0])
    pts = ctx['odata']['points'][-1].cuda()
    fc = o['feats_c_enc'].contiguous().cuda()
    buf = ops.feat_empty(pts.shape[0], 256, 'cuda')
    p4 = net._pts4(pts)
    net._thdroformer('transformer', p4, fc, n_c, 4, buf)
    assert rel_err(buf[:n_c], o['t1_ref']) <= 2e-5 and rel_err(buf[n_c:], o['t1_src']) <= 2e-5


def test_nms_and_grouping_are_bit_exact(ctx):
    ofw, cfg, o, ops = ctx['ofw'], ctx['cfg'], ctx['otaps'], ctx['ops']
    L = ctx['odata']['lengths'][-1]
    shifted = o['vote_xyz'].cuda()
    flags = torch.zeros(8, dtype=torch.int32, device='cuda')
    idx = ops.radius_search_device(shifted, shifted, L.cuda(), L.cuda(), cfg.Vote.NMS_radius, cfg.neighbor_limits[-1], flags)
    w = min(cfg.neighbor_limits[-1], int(flags[0]))
    assert torch.equal(idx[:, :w].cpu(), o['nms_idx'])
    keep = ops.nms(idx, flags)
    assert torch.equal(keep.cpu().bool(), o['nms_mask'])
    # grouping, teacher-forced with the oracle's surviving nodes
    n_f = int(ctx['odata']['lengths'][1][0])
    pts_f = ctx['odata']['points'][1]
    for side, lo, hi in (('ref', 0, n_f), ('src', n_f, None)):
        nodes = ctx['oout'][f'{side}_points_c']
        nm, knn, km = ops.point_to_node(pts_f[lo:hi].cuda().contiguous(), nodes.cuda().contiguous(), 128, flags[4:])
        assert torch.equal(nm.cpu().bool(), o[f'{side}_node_masks'])
        assert torch.equal(km.cpu().bool(), o[f'{side}_knn_masks'])
        assert torch.equal(knn.cpu(), o[f'{side}_knn'])
    assert int(flags[4]) == 0


def test_coarse_matching_stage(ctx):
    """Teacher-forced with the oracle's superpoint features: the selected pairs equal the oracle's, and a pair sits at
    another position only inside a group of scores tied to 1e-5 relative (the oracle evaluates the scores in fp32 with
    this host's BLAS: neighbouring scores are as close as 2e-6 relative here, 4e-7 on the full pair,
    tests/golden/coarse_order_analysis.json, closer than fp32 sums in another order resolve).  The ORDER is pinned
    where it can be -- against the reference's own captured indices, fed the reference's captured features, on all seven
    golden cases (test_reference_goldens_gpu.py::test_coarse_matching_reproduces_reference_indices_teacher_forced).
    The stage itself runs in fp64 (rdm_coarse_matching_features)."""
    o, oo, ops = ctx['otaps'], ctx['oout'], ctx['ops']
    rf, sf = oo['ref_feats_c'].cuda(), oo['src_feats_c'].cuda()
    ri, si, sc, cnt = ops.coarse_matching_features(rf, sf, o['ref_node_masks'].cuda().to(torch.uint8),
                                                   o['src_node_masks'].cuda().to(torch.uint8), 256)
    k = int(cnt)
    assert k == oo['ref_node_corr_indices'].shape[0]
    want_list = list(zip(oo['ref_node_corr_indices'].tolist(), oo['src_node_corr_indices'].tolist()))
    got_list = list(zip(ri[:k].cpu().tolist(), si[:k].cpu().tolist()))
    assert set(got_list) == set(want_list) and len(set(got_list)) == k
    pos = {p: i for i, p in enumerate(want_list)}
    perm = torch.tensor([pos[p] for p in got_list])
    os_ = o['node_corr_scores'].double()
    assert float((os_[perm] - os_).abs().max() / os_.max()) <= 1e-5  # moved only inside near-tie groups
    assert rel_err(sc[:k], o['node_corr_scores'][perm]) <= 3e-6  # the oracle's scores carry their own fp32 rounding
    # the fp32 pipeline (GEMM, then rdm_coarse_matching) selects the same set up to near-ties at the cut
    sim = ops.gemm(rf, sf, 256, sf.shape[0], trans_b=True)
    ri2, si2, sc2, cnt2 = ops.coarse_matching(sim, o['ref_node_masks'].cuda().to(torch.uint8),
                                              o['src_node_masks'].cuda().to(torch.uint8), 256)
    assert int(cnt2) == k and rel_err(sc2[:k], o['node_corr_scores']) <= 1e-5
    got = set(zip(ri2[:k].cpu().tolist(), si2[:k].cpu().tolist()))
    want = set(zip(oo['ref_node_corr_indices'].tolist(), oo['src_node_corr_indices'].tolist()))
    assert len(got ^ want) <= 4


def test_sinkhorn_stage(ctx):
    o, oo, ops, W, cfg = ctx['otaps'], ctx['oout'], ctx['ops'], ctx['W'], ctx['cfg']
    ms = ops.sinkhorn(o['patch_scores'].cuda().contiguous(), oo['ref_node_corr_knn_masks'].cuda().to(torch.uint8).contiguous(),
                      oo['src_node_corr_knn_masks'].cuda().to(torch.uint8).contiguous(), W['optimal_transport.alpha'].cuda(), 100)
    ref = oo['matching_scores']
    valid = ref > -1e11
    assert torch.equal((ms.cpu() > -1e11), valid)
    # log-scores of magnitude ~1.6e2 after 100 iterations: 3e-6 of the maximum (the bound of the reference-golden test)
    assert (ms.cpu()[valid] - ref[valid]).abs().max().item() <= 3e-6 * ref[valid].abs().max().item()
    assert torch.equal(ms.cpu()[~valid], ref[~valid])                 # fl(-1e12) exactly


def test_lgr_stage_teacher_forced(ctx):
    """Identical Sinkhorn output in -> correspondences bit-exact, pose within RRE 1e-3 deg / RTE 1e-3 cm."""
    oo, ops, cfg, ofw = ctx['oout'], ctx['ops'], ctx['cfg'], ctx['ofw']
    fm = cfg.fine_matching
    rc, sc, cs, T, counts = ops.lgr(oo['matching_scores'].cuda().contiguous(), oo['ref_node_corr_knn_points'].cuda().contiguous(),
                                    oo['src_node_corr_knn_points'].cuda().contiguous(),
                                    oo['ref_node_corr_knn_masks'].cuda().to(torch.uint8).contiguous(),
                                    oo['src_node_corr_knn_masks'].cuda().to(torch.uint8).contiguous(), fm.acceptance_radius,
                                    fm.correspondence_threshold, fm.num_refinement_steps)
    C = int(counts[0])
    assert C == oo['corr_scores'].shape[0]
    assert torch.equal(rc[:C].cpu(), oo['ref_corr_points']) and torch.equal(sc[:C].cpu(), oo['src_corr_points'])
    assert rel_err(cs[:C], oo['corr_scores']) <= 1e-6
    assert int(counts[1]) == len(ctx['otaps']['lgr']['chunks'])
    rre, rte = ofw.rre_rte(T.cpu().numpy(), oo['estimated_transform'].numpy())
    assert rre <= 1e-3 and rte <= 1e-5, (rre, rte)  # degrees, metres (= 1e-3 cm)


@pytest.mark.parametrize('full', [True, False])
def test_lgr_with_more_correspondences_than_the_refinement_stages_in_lds(ctx, full):
    """Trained weights give confident Sinkhorn outputs: up to 256 x (128 + 128) correspondences
    (local_global_registration.py:145-202), where random weights give a few hundred.  Synthetic confident
    `matching_scores` for 256 patches -> 16-33 k correspondences, beyond the 4 608 that lgr_refine_kernel stages in LDS
    (its global-memory branch) -- teacher-forced against oracle.forward.lgr: correspondences bit-exact and in
    torch.nonzero order, scores to 1e-6, the same hypothesis, pose within 1e-3 deg / 1e-3 cm of the float64 Procrustes of
    the final inliers and 1e-3 deg / 1e-2 cm of the oracle's fp32 pose (tens of thousands of fp32 terms in its sums)."""
    ops, cfg, ofw = ctx['ops'], ctx['cfg'], ctx['ofw']
    fm = cfg.fine_matching
    g = torch.Generator().manual_seed(11 + int(full))
    B, K = 256, 128
    ang = 0.3
    R = torch.tensor([[np.cos(ang), -np.sin(ang), 0.0], [np.sin(ang), np.cos(ang), 0.0], [0.0, 0.0, 1.0]], dtype=torch.float32)
    t = torch.tensor([4.0, -2.5, 0.3])
    centres = (torch.rand(B, 1, 3, generator=g) - 0.5) * torch.tensor([60.0, 60.0, 4.0])
    ref_knn = centres + (torch.rand(B, K, 3, generator=g) - 0.5) * 5.0
    perm = torch.stack([torch.randperm(K, generator=g) for _ in range(B)])  # src slot of ref point i: perm[b, i]
    src_true = (ref_knn - t) @ R  # R^T (ref - t) row-wise
    src_knn = torch.zeros(B, K, 3)
    src_knn.scatter_(1, perm[:, :, None].expand(-1, -1, 3), src_true + 0.03 * torch.randn(B, K, 3, generator=g))
    n_valid = torch.full((B,), K) if full else torch.randint(2, K + 1, (B,), generator=g)
    n_valid[3] = 2  # below correspondence_threshold: no hypothesis from this patch
    ref_mask = torch.arange(K)[None] < n_valid[:, None]
    src_mask = torch.zeros(B, K, dtype=torch.bool).scatter_(1, perm, ref_mask)
    # every eighth patch is an outlier patch: its src points are somewhere else (other hypotheses, fewer inliers)
    bad = torch.arange(B) % 8 == 5
    src_knn[bad] += torch.tensor([7.0, 3.0, 0.0])
    # log-scores: the true match dominates its row and column, everything else and the dustbins are small; a tenth of
    # the rows prefer the dustbin (no correspondence from them); masked entries as the Sinkhorn leaves them
    logit = torch.full((B, K + 1, K + 1), -12.0) + 0.5 * torch.randn(B, K + 1, K + 1, generator=g)
    logit[:, :-1, :-1].scatter_(2, perm[:, :, None], -0.2 + 0.05 * torch.randn(B, K, 1, generator=g))
    logit[:, :-1, -1] = torch.where(torch.rand(B, K, generator=g) < 0.1, torch.tensor(-0.05), torch.tensor(-4.0))
    logit[:, -1, :-1] = -4.0
    full_r = torch.cat([ref_mask, torch.ones(B, 1, dtype=torch.bool)], 1)
    full_s = torch.cat([src_mask, torch.ones(B, 1, dtype=torch.bool)], 1)
    logit = torch.where(full_r[:, :, None] & full_s[:, None, :], logit, torch.tensor(-1e12))
    o_rc, o_sc, o_cs, o_T, info = ofw.lgr(ref_knn, src_knn, ref_mask, src_mask, logit, cfg)
    C = o_cs.shape[0]
    assert C > 2 * 4608 and len(info['chunks']) >= 200
    rc, sc, cs, T, counts = ops.lgr(logit.cuda().contiguous(), ref_knn.cuda().contiguous(), src_knn.cuda().contiguous(),
                                    ref_mask.cuda().to(torch.uint8).contiguous(), src_mask.cuda().to(torch.uint8).contiguous(),
                                    fm.acceptance_radius, fm.correspondence_threshold, fm.num_refinement_steps)
    assert int(counts[0]) == C and int(counts[1]) == len(info['chunks'])
    assert torch.equal(rc[:C].cpu(), o_rc) and torch.equal(sc[:C].cpu(), o_sc)
    assert rel_err(cs[:C], o_cs) <= 1e-6
    assert int(counts[2]) == info['best']  # the FIRST argmax: most hypotheses tie exactly here (every clean patch explains every clean patch)
    Tn = T.cpu().double().numpy()
    res = np.linalg.norm(o_rc.double().numpy() - (o_sc.double().numpy() @ Tn[:3, :3].T + Tn[:3, 3]), axis=1)
    T64, _ = ofw.procrustes_fp64(o_sc.double().numpy(), o_rc.double().numpy(), o_cs.double().numpy() * (res < fm.acceptance_radius))
    rre, rte = ofw.rre_rte(Tn, T64)
    assert rre <= 1e-3 and rte <= 1e-5, (rre, rte)
    rre, rte = ofw.rre_rte(Tn, o_T.numpy())
    assert rre <= 1e-3 and rte <= 1e-4, (rre, rte)


def test_forward_end_to_end(ctx):
    net, ofw, oo, o = ctx['net'], ctx['ofw'], ctx['oout'], ctx['otaps']
    data = ctx['collate'].collate_pair(ctx['rp'], ctx['sp'], ctx['cfg'])
    taps = {}
    out = net(data, taps)
    for k in ('t1_ref', 't1_src', 'vote_feats', 'decoder'):
        assert rel_err(taps[k], o[k]) <= 1e-4, k
    assert rel_err(taps['vote_xyz'], o['vote_xyz']) <= 1e-6
    assert torch.equal(taps['nms_mask'].cpu().bool(), o['nms_mask'])
    for k in ('ref_points_c', 'src_points_c'):
        assert rel_err(out[k], oo[k]) <= 1e-6, k
    for k in ('ref_feats_c', 'src_feats_c', 'ref_feats_f', 'src_feats_f', 'ref_n2p_scores_c', 'src_n2n_scores_c',
              'ref_p2p_scores_c'):
        assert rel_err(out[k], oo[k]) <= 2e-4, k
    assert set(out.keys()) == set(oo.keys())
    rre, rte = ofw.rre_rte(out['estimated_transform'].cpu().numpy(), oo['estimated_transform'].numpy())
    # end to end: the discrete outputs equal the oracle's -- the same superpoint pairs (at another position only inside
    # groups of scores tied to 1e-5: the oracle ranks fp32 scores of this host's BLAS), the same point correspondences as
    # a set -- and the pose meets the north-star bound (1e-3 deg, 1e-3 cm); the crop's best LGR hypothesis leads by 7 inliers
    import tie_aware
    got = list(zip(out['ref_node_corr_indices'].cpu().tolist(), out['src_node_corr_indices'].cpu().tolist()))
    want = list(zip(oo['ref_node_corr_indices'].tolist(), oo['src_node_corr_indices'].tolist()))
    tie_aware.pair_permutation(got, want, o['node_corr_scores'].numpy())
    assert set(tie_aware.corr_rows(out['ref_corr_points'].cpu(), out['src_corr_points'].cpu())) == \
        set(tie_aware.corr_rows(oo['ref_corr_points'], oo['src_corr_points']))
    assert rre <= 1e-3 and rte <= 1e-5, (rre, rte)
