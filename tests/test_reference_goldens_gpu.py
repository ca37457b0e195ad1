"""GPU: the HIP path against vectors captured from the REFERENCE ITSELF (tests/golden/forward_*.npz, produced by
tests/golden/gen_golden.py importing /root/reference: experiments/model_infer.py:109-354 behind the reference's own
collate) -- no oracle in between.  Seven cases: both bundled pairs of the reference's `infer` subset at full size
(0<->4: 20 524 + 19 085 points, 0<->7; kitti/dataset.py:56-64 = BASELINE.json configs[0]), pair 0<->4 with a second
weight seed, two 2 x 16 k-point synthetic pairs (configs[1]'s workload), the 10 m crop of 0<->4 (`small`) and the
9 m crop (`crop9`, the documented near-tie case).

Integer / index outputs must be EQUAL (NMS mask, superpoint correspondences, patch masks, point correspondences as a
set -- except where the golden file records that the reference differs from ITSELF between its 8- and 1-thread CPU
runs, `self/*` entries: synth0's correspondences differ by one row there, and then at most that many rows may differ
here); float outputs within the stated relative bounds.  Pose: with random weights the reference's local-to-global
registration picks among hypotheses whose inlier counts tie within one (local_global_registration.py:204-221; margins
recorded per case in oracle_vs_reference.json: 0-2 everywhere but `small`, and on 0<->7 the reference's own two runs
return poses 121 deg apart).  The assertion is therefore tie-aware: the HIP pose must match ONE of the poses the
reference returns from the hypotheses within one inlier of its best (`lgr/alt_transforms`, entry of the reference's own
choice first) -- within the north star's RRE <= 1e-3 deg / RTE <= 1e-3 cm on the crops, RRE <= 1e-3 deg / RTE <= 1e-2 cm
at full size (64-80 m coordinates: one fp32 ulp is 8e-4 cm, the reference against itself moves by 2e-3 ... 5e-3 cm).
On `small` (margin 7) only the reference's own hypothesis is accepted.  The measured deviations are written to
gpurun_out/hip_vs_reference.json; a copy of a run is tracked as tests/golden/hip_vs_reference.json."""
import json
import os

import numpy as np
import pytest
import torch

import tie_aware
from sampling import expand_scores, sample

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_report = {}


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)) if a.size else 0.0


def npy(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


rre_rte = tie_aware.rre_rte


TAGS = ['pair04', 'pair07', 'pair04_seed1', 'synth0', 'synth3', 'small', 'crop9', 'lowoverlap', 'dense20k']


@pytest.fixture(scope='module')
def setup():
    from rdmnet_amd import collate, config, engine, model, weights
    cfg = config.make_cfg()
    built = {}

    def for_seed(seed):  # the weights the golden case was generated with (`weight_seed` in the file)
        if seed not in built:
            state = weights.synthetic_state_dict(cfg, seed=seed)
            net = model.create_model(cfg).cuda()
            net.load_state_dict(state)
            eng = engine.Engine(cfg, state)
            eng.keep_taps(True)
            built[seed] = (net, eng)
        return built[seed]

    yield cfg, for_seed, collate
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', 'hip_vs_reference.json'), 'w') as f:
        json.dump(_report, f, indent=1, sort_keys=True)


@pytest.mark.parametrize('tag', TAGS)
def test_hip_forward_matches_reference_goldens(setup, golden_dir, tag):
    cfg, for_seed, collate = setup
    g = np.load(os.path.join(golden_dir, f'forward_{tag}.npz'))
    net, eng = for_seed(int(g['weight_seed']))
    full_size = tag not in ('small', 'crop9')
    rp, sp = g['ref_points_in'], g['src_points_in']
    data = collate.collate_pair(rp, sp, cfg, exact_shapes=True)
    for i in range(5):
        assert np.array_equal(npy(data['lengths'][i]), g[f'lengths{i}'])
    taps = {}
    out = net(data, taps)
    rep = _report.setdefault(tag, {})

    # ---- float stages against the reference's captured tensors (row/column samples of the same rule, tests/sampling.py)
    for k in g.files:
        if k.startswith('tap/encoder.'):
            rep[k] = rel(sample(npy(taps[k[4:]])), g[k])
            assert rep[k] <= 2e-5, (k, rep[k])
    for k in ('t1_ref', 't1_src', 't2_ref', 't2_src', 'vote_feats', 'decoder'):
        rep['tap/' + k] = rel(sample(npy(taps[k])), g['tap/' + k])
        assert rep['tap/' + k] <= 2e-5, (k, rep['tap/' + k])
    rep['tap/vote_xyz'] = rel(npy(taps['vote_xyz']), g['tap/vote_xyz'])
    assert rep['tap/vote_xyz'] <= 1e-6

    # ---- discrete decisions: equal
    assert np.array_equal(npy(taps['nms_mask']).astype(bool), g['tap/nms_mask'])
    for k in ('ref_points_c', 'src_points_c', 'ori_ref_points_c', 'ori_src_points_c', 'shifted_ref_points_c',
              'shifted_src_points_c'):
        rep['out/' + k] = rel(npy(out[k]), g['out/' + k])
        assert rep['out/' + k] <= 1e-6, k
    for k in ('ref_n2p_scores_c', 'src_n2p_scores_c', 'ref_n2n_scores_c', 'src_n2n_scores_c'):
        rep['out/' + k] = rel(npy(out[k]), g['out/' + k])
        assert rep['out/' + k] <= 2e-5, k
    for k in ('ref_feats_f', 'src_feats_f', 'ref_p2p_scores_c', 'src_p2p_scores_c', 'ref_feats_c', 'src_feats_c'):
        rep['out/' + k] = rel(sample(npy(out[k])), g['out/' + k])
        assert rep['out/' + k] <= 2e-5, k
    # ---- superpoint correspondences (superpoint_matching.py:14-83).  The reference's ORDER among its top-256 scores is
    # decided by its own fp32 rounding: relative gaps between neighbouring scores go down to 3e-7, and the reference
    # formula evaluated in fp64 on the reference's own features already disagrees with its fp32 order at 2-3 % of
    # the positions (tests/golden/coarse_order_analysis.json).  So: the SET of pairs must be equal, and a pair may
    # only sit at a position whose reference score equals its own reference score within 1e-5 (i.e. inside a near-tie
    # group); everything per patch is then compared through that permutation, and correspondences as a set.
    ref_pairs = list(zip(g['out/ref_node_corr_indices'].tolist(), g['out/src_node_corr_indices'].tolist()))
    hip_pairs = list(zip(npy(out['ref_node_corr_indices']).tolist(), npy(out['src_node_corr_indices']).tolist()))
    rep['node_corr_set_symmetric_difference'] = len(set(ref_pairs) ^ set(hip_pairs))
    rep['node_corr_same_position_fraction'] = float(np.mean([a == b for a, b in zip(ref_pairs, hip_pairs)]))
    assert int(g['self/node_corr_symmetric_difference']) == 0  # the reference agrees with itself on every case
    rs = g['tap/node_corr_scores'].astype(np.float64)
    perm, rep['node_corr_max_tie_gap'] = tie_aware.pair_permutation(hip_pairs, ref_pairs, rs)  # HIP position -> reference's
    rep['tap/node_corr_scores'] = rel(npy(taps['node_corr_scores']), rs[perm])
    assert rep['tap/node_corr_scores'] <= 1e-5
    # ---- patches: the same points per patch.  Two points of a patch swap places where they are equidistant from the node
    # within the cancellation noise of the reference's |x|^2 - 2xy + |y|^2 (synth3: gaps of 5e-4 m^2 at |x|^2 = 4000 m^2,
    # node positions differing in the last bit); rows / columns are matched by point, then everything must agree.
    rmask, smask = g['out/ref_node_corr_knn_masks'][perm], g['out/src_node_corr_knn_masks'][perm]
    rpts, spts = g['out/ref_node_corr_knn_points'][perm], g['out/src_node_corr_knn_points'][perm]
    assert np.array_equal(npy(out['ref_node_corr_knn_masks']).astype(bool), rmask)
    assert np.array_equal(npy(out['src_node_corr_knn_masks']).astype(bool), smask)
    hr_pts, hs_pts = npy(out['ref_node_corr_knn_points']), npy(out['src_node_corr_knn_points'])
    gold_ms = expand_scores(g['out/matching_scores'], g['out/ref_node_corr_knn_masks'], g['out/src_node_corr_knn_masks'])[perm]
    hip_ms = npy(out['matching_scores'])
    swapped = 0
    for b in range(len(perm)):
        pr = tie_aware.patch_permutation(hr_pts[b], rmask[b], rpts[b], rmask[b])
        pc = tie_aware.patch_permutation(hs_pts[b], smask[b], spts[b], smask[b])
        swapped += int((pr != np.arange(len(pr))).any() or (pc != np.arange(len(pc))).any())
        gold_ms[b] = gold_ms[b][np.r_[pr, len(pr)]][:, np.r_[pc, len(pc)]]
    rep['patches_with_swapped_points'] = swapped
    assert swapped <= len(perm) // 16
    valid = gold_ms > -1e11
    assert np.array_equal(hip_ms > -1e11, valid)
    rep['out/matching_scores'] = rel(hip_ms[valid], gold_ms[valid])
    # (log-scores of magnitude 1e2 ... 1e3 after 100 Sinkhorn iterations; the reference against itself, 8 vs 1 thread,
    # moves by 2e-7 ... 4e-7 of the maximum where that can be measured)
    assert rep['out/matching_scores'] <= 3e-6
    # ---- point correspondences: the same set of (ref point, src point) rows, scores attached; where the reference's own
    # 8- and 1-thread runs differ in k rows (synth0: k = 2, one correspondence at the top-1-versus-dustbin threshold), at
    # most k rows may differ here
    hs = tie_aware.corr_rows(npy(out['ref_corr_points']), npy(out['src_corr_points']), npy(out['corr_scores']))
    gs = tie_aware.corr_rows(g['out/ref_corr_points'], g['out/src_corr_points'], g['out/corr_scores'])
    rep['corr_points_symmetric_difference'] = len(set(hs) ^ set(gs))
    rep['n_corr'] = [len(hs), len(gs)]
    assert rep['corr_points_symmetric_difference'] <= int(g['self/corr_symmetric_difference']), rep['n_corr']
    common = sorted(set(hs) & set(gs))
    rep['out/corr_scores'] = rel([hs[k] for k in common], [gs[k] for k in common])
    assert rep['out/corr_scores'] <= 2e-5

    # ---- pose (tie-aware, see the module docstring)
    T = npy(out['estimated_transform'])
    margin = int(np.sort(g['lgr/inlier_counts'])[::-1][:2] @ [1, -1])
    own = int(np.nonzero(g['lgr/alt_hypotheses'] == g['lgr/best'])[0][0])
    errs = [rre_rte(T, A) for A in g['lgr/alt_transforms']]
    pick = int(np.argmin([e[0] for e in errs]))
    rre, rte = errs[pick]
    rre_own, rte_own = rre_rte(T, g['out/estimated_transform'])
    rep['pose'] = {'rre_deg': rre, 'rte_m': rte, 'reference_inlier_margin': margin, 'n_near_tie_hypotheses': len(errs),
                   'matched_the_references_own_hypothesis': pick == own,
                   'vs_reference_pose': {'rre_deg': rre_own, 'rte_m': rte_own},
                   'reference_8_vs_1_thread': dict(zip(('rre_deg', 'rte_m'), rre_rte(g['self/transform_1_thread'],
                                                                                    g['out/estimated_transform'])))}
    # 0<->7 is the one case whose pose the reference does not reproduce itself (a handful of inliers at the 0.6 m
    # acceptance radius, hypothesis margin 0: its 8- and 1-thread runs return poses 121 deg apart); the bar there is the
    # next block
    self_pose = rep['pose']['reference_8_vs_1_thread']
    assert (self_pose['rre_deg'] <= 1e-3 and self_pose['rte_m'] <= 1e-4) == (tag != 'pair07')
    bound_t = 1e-4 if full_size else 1e-5
    if tag != 'pair07':
        assert rre <= 1e-3 and rte <= bound_t, (rre, rte, margin)
    # on every case the HIP pose IS one of the two poses the reference itself returned (8- or 1-thread run)
    e1 = rre_rte(T, g['self/transform_1_thread'])
    rep['pose']['vs_reference_1_thread_pose'] = {'rre_deg': e1[0], 'rte_m': e1[1]}
    # (component-wise: RRE AND RTE of the SAME candidate pose -- ADVICE r3: a tuple comparison would ignore RTE)
    assert (rre_own <= 1e-3 and rte_own <= bound_t) or (e1[0] <= 1e-3 and e1[1] <= bound_t), ((rre_own, rte_own), e1)
    if margin >= 3:
        assert pick == own and len(errs) == 1
    # the pose is the reference's local-to-global registration OF THIS RUN'S OWN matching scores: the restated LGR
    # (oracle.forward.lgr, bit-exact against the reference on all seven golden cases, test_oracle_forward.py) fed the
    # HIP patch points / masks / Sinkhorn output returns the same correspondences, and the HIP pose is the pose it returns
    # from one of the hypotheses within one inlier of its best
    from oracle import forward as ofw
    (orc, osc, _), alts, own_margin = tie_aware.lgr_alternatives(
        ofw, cfg, out['ref_node_corr_knn_points'].cpu(), out['src_node_corr_knn_points'].cpu(),
        out['ref_node_corr_knn_masks'].cpu().bool(), out['src_node_corr_knn_masks'].cpu().bool(), out['matching_scores'].cpu())
    assert torch.equal(orc, out['ref_corr_points'].cpu()) and torch.equal(osc, out['src_corr_points'].cpu())
    e2 = min((rre_rte(T, A) for _, A in alts), key=lambda e: max(e[0] / 1e-3, e[1] / bound_t))  # both components of one pose
    rep['pose']['vs_reference_lgr_on_own_scores'] = {'rre_deg': e2[0], 'rte_m': e2[1], 'inlier_margin': own_margin,
                                                      'n_near_tie_hypotheses': len(alts)}
    assert e2[0] <= 1e-3 and e2[1] <= bound_t, (e2, own_margin)

    # ---- the native engine (what bench.py measures) returns the same result bit for bit at this size
    eng.run(torch.from_numpy(rp).cuda(), torch.from_numpy(sp).cuda())
    assert np.array_equal(eng.transform(), npy(out['estimated_transform']))
    rc, sc, cs = eng.corr()
    assert torch.equal(rc, out['ref_corr_points']) and torch.equal(sc, out['src_corr_points']) and torch.equal(cs, out['corr_scores'])
    assert torch.equal(eng.tensor('nms_mask')[:, 0], taps['nms_mask'])
    assert torch.equal(eng.tensor('ref_node_corr_indices')[:, 0], out['ref_node_corr_indices'])


@pytest.mark.parametrize('tag', TAGS)
def test_coarse_matching_reproduces_reference_indices_teacher_forced(golden_dir, tag):
    """Fed the REFERENCE's own superpoint features and non-empty-node masks (the un-sampled `full/*` entries of the
    golden file), the HIP stage returns the reference's captured ref/src_node_corr_indices (superpoint_matching.py:14-83)
    -- the same 256 pairs, and a pair at another position only where the reference's own scores are tied to 2e-6
    relative (neighbouring scores are as close as 4e-7, tests/golden/coarse_order_analysis.json: the order inside such a
    group is the reference's fp32 rounding, which its own 1-thread run does not reproduce either on synth0); on 0<->4
    (both crops and the full pair) the order is the reference's at all 256 positions."""
    from rdmnet_amd import ops
    g = np.load(os.path.join(golden_dir, f'forward_{tag}.npz'))
    ri, si, sc, cnt = ops.coarse_matching_features(torch.from_numpy(g['full/ref_feats_c']).cuda(),
                                                   torch.from_numpy(g['full/src_feats_c']).cuda(),
                                                   torch.from_numpy(g['full/ref_node_masks']).cuda().to(torch.uint8),
                                                   torch.from_numpy(g['full/src_node_masks']).cuda().to(torch.uint8), 256)
    k = int(cnt)
    assert k == g['out/ref_node_corr_indices'].shape[0]
    want = list(zip(g['out/ref_node_corr_indices'].tolist(), g['out/src_node_corr_indices'].tolist()))
    got = list(zip(npy(ri)[:k].tolist(), npy(si)[:k].tolist()))
    perm, gap = tie_aware.pair_permutation(got, want, g['tap/node_corr_scores'], tie=2e-6)
    same = float(np.mean(perm == np.arange(k)))
    rep = _report.setdefault(tag, {})
    rep['teacher_forced_coarse_same_position_fraction'], rep['teacher_forced_coarse_max_tie_gap'] = same, gap
    if tag in ('pair04', 'small', 'crop9'):
        assert same == 1.0
    err = rel(npy(sc)[:k], g['tap/node_corr_scores'][perm])
    _report.setdefault(tag, {})['teacher_forced_coarse_scores'] = err
    assert err <= 4e-6  # the reference's scores carry their own fp32 rounding (measured: <= 2.2e-6 on seven cases, 3.02e-6 on `lowoverlap`)


@pytest.mark.parametrize('tag', ['synth0', 'synth3', 'dense20k'])
def test_bf16_attention_deviation_from_the_fp32_reference_goldens(setup, golden_dir, tag):
    """BASELINE configs[3] (bf16 operands in QK^T and PV, fp32 softmax / accumulators / pose solve) has no counterpart in
    the reference; test_full_size_configs_gpu.py compares it with the oracle's bf16 restatement.  Here the same mode runs on
    the two bench-workload goldens and its deviation from the REFERENCE's fp32 run is measured and bounded: what a user
    who switches the mode on gives up against the reference, not against our own restatement.  Recorded under `bf16/*`
    in hip_vs_reference.json.  Bounds (measured values in tests/golden/hip_vs_reference.json, asserted with a margin):
    encoder taps untouched (<= 2e-5), transformer / vote taps <= 2e-3 of the tensor maximum, NMS mask Hamming distance
    <= 2 % of the superpoints, >= 90 % of the superpoint pairs and >= 80 % of the point correspondences in common, pose
    within 2e-2 deg / 1 cm of the reference's."""
    from rdmnet_amd import config, model, weights
    cfg, _for_seed, collate = setup
    cfg16 = config.make_cfg()
    cfg16.thdroformer.attention_bf16 = True
    g = np.load(os.path.join(golden_dir, f'forward_{tag}.npz'))
    net = model.create_model(cfg16).cuda()
    net.load_state_dict(weights.synthetic_state_dict(cfg16, seed=int(g['weight_seed'])))
    data = collate.collate_pair(g['ref_points_in'], g['src_points_in'], cfg16, exact_shapes=True)
    taps = {}
    out = net(data, taps)
    rep = _report.setdefault(tag, {}).setdefault('bf16', {})
    for k in g.files:
        if k.startswith('tap/encoder.'):
            assert rel(sample(npy(taps[k[4:]])), g[k]) <= 2e-5, k
    for k in ('t1_ref', 't1_src', 't2_ref', 't2_src', 'vote_feats', 'decoder'):
        rep['tap/' + k] = rel(sample(npy(taps[k])), g['tap/' + k])
        assert rep['tap/' + k] <= 2e-3, (k, rep['tap/' + k])
    mask, gmask = npy(taps['nms_mask']).astype(bool), g['tap/nms_mask']
    rep['nms_mask_hamming'] = int((mask != gmask).sum())
    rep['n_superpoints'] = int(gmask.size)
    assert rep['nms_mask_hamming'] <= 0.02 * gmask.size, rep
    # superpoint pairs are indices into the NMS survivors of each cloud: translated to the indices of the coarse points they
    # came from (k-th survivor -> k-th set bit of the cloud's part of the mask), which both runs share
    n_ref_c = int(g['lengths4'][0])

    def pair_keys(m, ri, si):
        r_ids, s_ids = np.nonzero(m[:n_ref_c])[0], np.nonzero(m[n_ref_c:])[0]
        return {(int(r_ids[a]), int(s_ids[b])) for a, b in zip(np.asarray(ri).tolist(), np.asarray(si).tolist())}
    hp = pair_keys(mask, npy(out['ref_node_corr_indices']), npy(out['src_node_corr_indices']))
    gp = pair_keys(gmask, g['out/ref_node_corr_indices'], g['out/src_node_corr_indices'])
    rep['node_corr_in_common'], rep['node_corr_reference'] = len(hp & gp), len(gp)
    assert len(hp & gp) >= 0.9 * len(gp), rep
    hs = set(tie_aware.corr_rows(npy(out['ref_corr_points']), npy(out['src_corr_points'])))
    gs = set(tie_aware.corr_rows(g['out/ref_corr_points'], g['out/src_corr_points']))
    rep['corr_in_common'], rep['corr_reference'], rep['corr_bf16'] = len(hs & gs), len(gs), len(hs)
    assert len(hs & gs) >= 0.8 * max(len(hs), len(gs)), rep
    rre, rte = rre_rte(npy(out['estimated_transform']), g['out/estimated_transform'])
    rep['pose_vs_reference'] = {'rre_deg': rre, 'rte_m': rte}
    assert rre <= 2e-2 and rte <= 1e-2, rep
