"""GPU: callers either side of the path and the two configuration variants of
BASELINE.json (configs[3] bf16 attention, configs[4] vote layer off), through the C-ABI."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, 'harness.npz'))


@pytest.fixture(scope='module')
def infer_set(scans):
    from rdmnet_amd import dataset
    return dataset.ArrayPairDataset([(scans['s000000'], scans['s000004']), (scans['s000000'], scans['s000007'])])


def test_neighbor_histogram_is_bincount(gold):
    from rdmnet_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(0)
    for n, hist_n in ((1, 5), (1000, 64), (123457, 607), (50000, 1024)):
        counts = rng.integers(0, hist_n + 40, n).astype(np.int32)
        c = torch.from_numpy(counts).cuda()
        hist = torch.zeros(hist_n, dtype=torch.int32, device='cuda')
        for _ in range(2):  # accumulates over calls
            _lib.check(L.rdm_neighbor_histogram(c.data_ptr(), n, hist.data_ptr(), hist_n, _lib.stream_ptr()), 'hist')
        assert np.array_equal(hist.cpu().numpy(), 2 * np.bincount(counts, minlength=hist_n)[:hist_n])
    assert L.rdm_neighbor_histogram(c.data_ptr(), n, hist.data_ptr(), 5000, _lib.stream_ptr()) != 0  # > 1024 bins


def test_calibration_matches_reference_limits(gold, infer_set):
    from rdmnet_amd import config, dataset
    b = config.make_cfg().backbone
    for thr, tag in ((2000, 'default'), (10 ** 9, 'all')):
        for ratio, want in zip(gold['calib_keep_ratios'], gold[f'calib_limits_{tag}']):
            got, hists = dataset.calibrate_neighbors_stack_mode(infer_set, None, b.num_stages, b.init_voxel_size, b.init_radius,
                                                                keep_ratio=float(ratio), sample_threshold=thr,
                                                                return_hists=True)
            assert hists.shape == (5, int(gold['calib_hist_n']))
            assert np.array_equal(np.asarray(got), want), (tag, ratio, got, want)
    assert list(got.shape) == [5] and config.make_cfg().neighbor_limits == list(gold['calib_limits_default'][1])


def test_histograms_equal_counts_of_the_collated_tables(infer_set):
    """The count-only path must see the same neighbourhoods the full collate writes."""
    from rdmnet_amd import collate, config, dataset
    cfg = config.make_cfg()
    b = cfg.backbone
    item = infer_set[0]
    hist_n = 607
    hists = dataset.neighbor_histograms(item, b.num_stages, b.init_voxel_size, b.init_radius, hist_n).cpu().numpy()
    cfg.neighbor_limits = [hist_n] * 5
    d = collate.collate_pair(item['ref_points'], item['src_points'], cfg, exact_shapes=True)
    for i in range(5):
        nb = d['neighbors'][i]
        counts = (nb < nb.shape[0]).sum(1).cpu().numpy()
        assert np.array_equal(hists[i], np.bincount(counts, minlength=hist_n)[:hist_n])


def test_pair_stager_delivers_every_pair_in_order(infer_set):
    from rdmnet_amd import dataset
    rng = np.random.default_rng(1)
    pairs = [(rng.normal(size=(100 + 7 * i, 3)).astype(np.float32), rng.normal(size=(90 + 3 * i, 3)).astype(np.float32))
             for i in range(9)]
    ds = dataset.ArrayPairDataset(pairs)
    for depth, workers, idx in ((1, 1, None), (2, 2, None), (3, 4, [8, 1, 5])):
        got = list(dataset.PairStager(ds, idx, depth=depth, workers=workers))
        want = list(range(9)) if idx is None else idx
        assert len(got) == len(want)
        for (item, r, s), i in zip(got, want):
            assert item['ref_frame'] == 2 * i
            torch.cuda.current_stream().synchronize()
            assert np.array_equal(r.cpu().numpy(), pairs[i][0]) and np.array_equal(s.cpu().numpy(), pairs[i][1])

    class Broken(dataset.ArrayPairDataset):
        def __getitem__(self, i):
            if i == 1:
                raise OSError('scan unreadable')
            return super().__getitem__(i)
    with pytest.raises(OSError):
        list(dataset.PairStager(Broken(pairs[:3]), depth=2, workers=1))


def test_tester_writes_reference_outputs(gold, infer_set, tmp_path):
    from rdmnet_amd import config, dataset, infer, weights
    cfg = config.make_cfg()
    t = infer.Tester(cfg, weights.synthetic_state_dict(cfg, seed=0), str(tmp_path))
    recs = t.run(dataset.PairStager(infer_set))
    assert len(recs) == 2
    lines = open(tmp_path / '00_pose').read().splitlines()
    assert len(lines) == 2
    for rec, line in zip(recs, lines):
        parts = line.split(' ')
        assert parts[-1] == '' and len(parts) == 15  # 'ref src ' + 12 x '%.6f ' (trailing blank as in the reference)
        assert [int(parts[0]), int(parts[1])] == [rec['ref_frame'], rec['src_frame']]
        np.testing.assert_allclose(np.array([float(x) for x in parts[2:14]]), rec['transform'].reshape(-1)[:12], atol=5.1e-7)
        z = np.load(tmp_path / f"0_{rec['src_frame']}_{rec['ref_frame']}.npz")
        assert set(gold['npz_keys']) <= set(z.files)
        assert np.array_equal(z['estimated_transform'], rec['transform'])
        assert z['ref_corr_points'].shape == (rec['n_corr'], 3) and z['corr_scores'].shape == (rec['n_corr'],)
        assert z['ref_points'].shape[0] == 20524 and z['ref_points_f'].shape[0] in (8145,)
        assert z['ref_feats_c'].shape == (z['ref_points_c'].shape[0], 256)
        assert z['ref_node_corr_indices'].max() < z['ref_points_c'].shape[0]
        np.testing.assert_allclose(np.linalg.norm(z['src_feats_c'], axis=1), 1.0, atol=1e-5)


def test_tester_with_ground_truth_reports_registration(tmp_path):
    from rdmnet_amd import config, dataset, infer, synthetic, weights
    cfg = config.make_cfg()
    ref, src, T = synthetic.make_pair(3, target_points=4000, tolerance=400)
    t = infer.Tester(cfg, weights.synthetic_state_dict(cfg, seed=0), None)
    rec = t.run(dataset.PairStager(dataset.ArrayPairDataset([(ref, src, T)])))[0]
    assert np.isfinite(rec['r_RRE']) and np.isfinite(rec['r_RTE']) and 'f_IR' in rec
    assert t.summary.lines()[2].startswith('  Registration, RR: ')


# ---- configs[4]: vote layer off (Mulran, infer.py:119-120) -------------------------------------------------------

def test_no_vote_variant_engine_per_op_and_oracle_agree(golden_dir):
    from oracle import forward as ofw
    from rdmnet_amd import collate, config, engine, model, weights
    cfg = config.make_cfg()
    cfg.Vote.inference_use_vote = False
    state = weights.synthetic_state_dict(cfg, seed=0)
    g = np.load(os.path.join(golden_dir, 'forward_small.npz'))
    rp, sp = g['ref_points_in'], g['src_points_in']
    net = model.create_model(cfg).cuda()
    net.load_state_dict(state)
    data = collate.collate_pair(rp, sp, cfg)
    out = net(data)
    eng = engine.Engine(cfg, state)
    eng.keep_taps(True)
    res = eng.run(torch.from_numpy(rp).cuda(), torch.from_numpy(sp).cuda())
    n_c = int(data['lengths'][-1][0])
    assert res.n_ref_nodes == n_c and res.n_src_nodes == data['points'][-1].shape[0] - n_c  # no NMS: every coarse point
    assert torch.equal(eng.tensor('nodes'), data['points'][-1])
    assert torch.equal(eng.tensor('feats_c')[:n_c], out['ref_feats_c'])
    assert torch.equal(eng.tensor('ref_node_corr_indices')[:, 0], out['ref_node_corr_indices'])
    assert np.array_equal(eng.transform(), out['estimated_transform'].cpu().numpy())
    # oracle with the same definition
    odata = ofw.pyramid(np.concatenate([rp, sp]), np.array([len(rp), len(sp)], np.int64), cfg)
    oout = ofw.forward(ofw.to_torch(state), cfg, odata)
    assert 'shifted_ref_points_c' not in out and 'shifted_ref_points_c' not in oout
    assert torch.equal(out['ref_points_c'].cpu(), oout['ref_points_c'])
    f, of = out['ref_feats_c'].cpu(), oout['ref_feats_c']
    assert (f - of).abs().max() <= 1e-4 * of.abs().max()  # fp32 features after 14 conv blocks + 8 attention layers
    rre, rte = ofw.rre_rte(out['estimated_transform'].cpu().numpy(), oout['estimated_transform'].numpy())
    assert rre < 0.05 and rte < 5e-4  # end-to-end bound; 1e-3 deg / 1e-3 cm holds teacher-forced


# ---- configs[3]: bf16 attention -----------------------------------------------------------------------------------

@pytest.mark.parametrize('nq,nk', [(431, 411), (16, 1), (333, 517), (5, 70)])
def test_attention_fp32_and_bf16_match_oracle(nq, nk):
    from oracle import forward as ofw
    from rdmnet_amd import ops
    rng = np.random.default_rng(nq)
    heads, d = 4, 128
    q, k, v = (torch.from_numpy(rng.normal(size=(n, d)).astype(np.float32) * 1.5) for n in (nq, nk, nk))

    def split(t):
        return t.view(t.shape[0], heads, d // heads).transpose(0, 1)
    for bf16, tol in ((False, 2e-6), (True, 1.5e-2)):
        want = ofw.dense_attention(split(q), split(k), split(v), bf16).transpose(0, 1).reshape(nq, d)
        got = ops.attention(q.cuda(), k.cuda(), v.cuda(), heads, bf16=bf16).cpu()
        # fp32: summation order only.  bf16: the kernel rounds exp(s - running max), the oracle
        # exp(s - final max) / sum; both are within bf16's 2^-9 of the exact probabilities.
        assert (got - want).abs().max() <= tol * want.abs().max(), (bf16, float((got - want).abs().max()))
    exact = ofw.dense_attention(split(q), split(k), split(v), False).transpose(0, 1).reshape(nq, d)
    assert (got - exact).abs().max() > 1e-5 * exact.abs().max() or nk == 1  # the bf16 path really rounds


def test_bf16_attention_engine_equals_per_op_path_and_stays_close_to_fp32(golden_dir):
    from rdmnet_amd import collate, config, engine, model, weights
    g = np.load(os.path.join(golden_dir, 'forward_small.npz'))
    rp, sp = g['ref_points_in'], g['src_points_in']
    outs = {}
    for bf16 in (False, True):
        cfg = config.make_cfg()
        cfg.thdroformer.attention_bf16 = bf16
        state = weights.synthetic_state_dict(cfg, seed=0)
        net = model.create_model(cfg).cuda()
        net.load_state_dict(state)
        taps = {}
        out = net(collate.collate_pair(rp, sp, cfg), taps)
        eng = engine.Engine(cfg, state)
        eng.keep_taps(True)
        eng.run(torch.from_numpy(rp).cuda(), torch.from_numpy(sp).cuda())
        n_c = taps['t1_ref'].shape[0]
        assert torch.equal(eng.tensor('t1')[:n_c], taps['t1_ref'])
        assert np.array_equal(eng.transform(), out['estimated_transform'].cpu().numpy())
        outs[bf16] = taps['t1_ref'].cpu()
    diff = (outs[True] - outs[False]).abs().max() / outs[False].abs().max()
    assert 1e-6 < diff < 5e-2, float(diff)  # 8 bf16 attention layers vs fp32: percent-level, not bit-equal


def test_low_overlap_pair_collate_is_bit_exact_and_pose_matches_oracle(oracle_native):
    """configs[4]: Mulran-shaped pair (70 deg of the second scan's field of view missing, >= 10 m apart,
    arbitrary yaw), vote layer off.  Index tensors bit-exact against the oracle's collate; pose and
    correspondences against the oracle's forward."""
    from oracle import forward as ofw
    from rdmnet_amd import collate, config, engine, synthetic, weights
    cfg = config.make_cfg()
    cfg.Vote.inference_use_vote = False
    ref, src, _ = synthetic.make_low_overlap_pair(1, target_points=5000, tolerance=400)
    assert src.shape[0] < 0.9 * ref.shape[0]
    odata = ofw.pyramid(np.concatenate([ref, src]), np.array([len(ref), len(src)], np.int64), cfg)
    data = collate.collate_pair(ref, src, cfg, exact_shapes=True)
    for key in ('points', 'neighbors', 'subsampling', 'upsampling'):
        for a, b in zip(data[key], odata[key]):
            assert torch.equal(a.cpu(), b), key
    state = weights.synthetic_state_dict(cfg, seed=0)
    eng = engine.Engine(cfg, state)
    r1 = eng.run(torch.from_numpy(ref).cuda(), torch.from_numpy(src).cuda())
    T1, n1 = eng.transform(), r1.n_correspondences
    r2 = eng.run(torch.from_numpy(ref).cuda(), torch.from_numpy(src).cuda())
    assert np.array_equal(T1, eng.transform()) and n1 == r2.n_correspondences  # deterministic
    oout = ofw.forward(ofw.to_torch(state), cfg, odata)
    # float stages against the oracle; the discrete decisions downstream (top-256 node pairs, per-patch
    # hypotheses, argmax of inlier counts) amplify 1-ulp feature noise with random weights, so -- as in
    # test_forward_gpu -- the pose bound is asserted teacher-forced on the oracle's Sinkhorn output.
    from rdmnet_amd import model, ops
    net = model.create_model(cfg).cuda()
    net.load_state_dict(state)
    out = net(collate.collate_pair(ref, src, cfg))
    assert np.array_equal(out['estimated_transform'].cpu().numpy(), T1)  # engine == per-op path
    for k in ('ref_feats_c', 'src_feats_c', 'ref_feats_f', 'src_feats_f'):
        assert (out[k].cpu() - oout[k]).abs().max() <= 2e-4 * oout[k].abs().max(), k
    fm = cfg.fine_matching
    rc, sc, cs, T, counts = ops.lgr(oout['matching_scores'].cuda().contiguous(), oout['ref_node_corr_knn_points'].cuda().contiguous(),
                                    oout['src_node_corr_knn_points'].cuda().contiguous(),
                                    oout['ref_node_corr_knn_masks'].cuda().to(torch.uint8).contiguous(),
                                    oout['src_node_corr_knn_masks'].cuda().to(torch.uint8).contiguous(), fm.acceptance_radius,
                                    fm.correspondence_threshold, fm.num_refinement_steps)
    C = int(counts[0])
    assert C == oout['corr_scores'].shape[0] and torch.equal(rc[:C].cpu(), oout['ref_corr_points'])
    # Low overlap + random weights leave a handful of inliers, here all but collinear: the covariance of the
    # final Procrustes is rank-deficient (sigma2/sigma1 ~ 1e-6), so the reference's fp32 torch.svd pose is
    # rounding noise (0.07 deg away from the float64 solution of the SAME inliers).  The bound that means
    # something is against that float64 solution; against the oracle only when the problem is conditioned.
    T_gpu = T.cpu().double().numpy()
    rcp, scp, w = (oout[k].double().numpy() for k in ('ref_corr_points', 'src_corr_points', 'corr_scores'))
    res = np.linalg.norm(rcp - (scp @ T_gpu[:3, :3].T + T_gpu[:3, 3]), axis=1)
    assert np.abs(res - fm.acceptance_radius).min() > 1e-3  # no inlier decision on the edge
    T64, S = ofw.procrustes_fp64(scp, rcp, w * (res < fm.acceptance_radius))
    rre, rte = ofw.rre_rte(T_gpu, T64)
    assert rre <= 1e-3 and rte <= 1e-5, (rre, rte)  # degrees, metres (= 1e-3 cm)
    if S[1] / S[0] > 1e-3:
        rre, rte = ofw.rre_rte(T_gpu, oout['estimated_transform'].numpy())
        assert rre <= 1e-3 and rte <= 1e-5, (rre, rte)
