"""Generates tests/golden/forward_*.npz by running the REFERENCE (imported from /root/reference with
the shims of ref_import.py) on CPU.  Build container only; the fixtures travel, the reference not.

For each case the reference's own collate (`registration_collate_fn_stack_mode`) and
`RDMNet.forward` run with the seeded synthetic state dict of rdmnet_amd.weights (loaded with
strict=True, which also proves the schema equals the reference checkpoint layout).  Stage outputs
are captured with forward hooks.  Large tensors are stored as row-strided samples.

The script also replays the oracle restatement (oracle/forward.py) on the same inputs and writes the
observed deviations to tests/golden/oracle_vs_reference.json -- the evidence that pins the oracle.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)
import ref_import  # noqa: E402
sys.path.insert(0, os.path.join(REPO, 'tests'))
from sampling import compact_scores, sample  # noqa: E402



def crop(points, radius):
    return points[np.linalg.norm(points[:, :2], axis=1) < radius]


def run_reference(cfg, model, ref_pts, src_pts):
    from geotransformer.utils.data import registration_collate_fn_stack_mode
    item = {'seq_id': 0, 'ref_frame': 0, 'src_frame': 1,
            'ref_points': ref_pts, 'src_points': src_pts,
            'ref_feats': np.ones((ref_pts.shape[0], 1), np.float32),
            'src_feats': np.ones((src_pts.shape[0], 1), np.float32)}
    data = registration_collate_fn_stack_mode([item], cfg.backbone.num_stages, cfg.backbone.init_voxel_size,
                                              cfg.backbone.init_radius, cfg.neighbor_limits)
    for key in ('neighbors', 'subsampling', 'upsampling'):
        data[key] = [x.contiguous() for x in data[key]]
    data['testing'] = True
    taps = {}
    hooks = []

    def tap(name):
        def fn(_m, _i, o):
            taps[name] = o
        return fn

    for n, m in model.named_modules():
        if n.startswith('encoder.encoder') and n.count('.') == 1:
            hooks.append(m.register_forward_hook(tap(n)))
    for n in ('transformer', 'transformer2', 'vote', 'nms', 'optimal_transport', 'coarse_matching', 'decoder'):
        hooks.append(getattr(model, n).register_forward_hook(tap(n)))
    with torch.no_grad():
        out = model(data)
    for h in hooks:
        h.remove()
    return data, out, taps


def corr_rows(out):
    a = np.concatenate([np_(out['ref_corr_points']), np_(out['src_corr_points'])], 1).astype(np.float64)
    return a[np.lexsort(a.T[::-1])]


def corr_symmetric_difference(out_a, out_b):
    a, b = (set(map(tuple, corr_rows(o).tolist())) for o in (out_a, out_b))
    return len(a ^ b)


def np_(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


def only_has_no(tag):
    """True when the command line names cases and `tag` is not one of them (its ~20 s of ray casting are skipped)."""
    return len(sys.argv) > 1 and tag not in sys.argv[1:]


def main():
    cfg = ref_import.make_cfg()
    from model_infer import create_model
    from rdmnet_amd import config as my_config, weights
    from oracle import forward as ofw

    my_cfg = my_config.make_cfg()
    cfg.neighbor_limits = list(my_cfg.neighbor_limits)
    torch.manual_seed(0)
    np.random.seed(0)
    model = create_model(cfg)
    model.eval()

    def load_seed(seed):
        state = weights.synthetic_state_dict(my_cfg, seed=seed)
        ref_sd = model.state_dict()
        assert list(ref_sd.keys()) == list(state.keys()), 'schema order differs from the reference state_dict'
        for k in ref_sd:
            assert tuple(ref_sd[k].shape) == tuple(state[k].shape), k
        model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
        return ofw.to_torch(state)

    scans = np.load(os.path.join(HERE, 'scans.npz'))
    synth = np.load(os.path.join(HERE, 'synthetic_pairs.npz'))
    # tag -> (ref scan, src scan, weight seed).  The reference's `infer` subset is the two bundled pairs 0<->4 and
    # 0<->7 (rdmnet/datasets/registration/kitti/dataset.py:56-64).  `small` is the 10 m crop of 0<->4: its best local
    # hypothesis leads the runner-up by >= 5 inliers in the reference's own LGR (8 and 1 thread), so summation-order
    # noise upstream cannot change the pose.  `crop9` (the 9 m crop, rounds 1-2's `small`) is kept as the documented
    # near-tie case: best and second-best hypothesis are 6 vs 5 inliers; the file carries the poses of every hypothesis
    # within one inlier of the best (`lgr/alt_transforms`) for a tie-aware assertion.
    cases = {
        'small': (crop(scans['s000000'], 10.0), crop(scans['s000004'], 10.0), 0),
        'crop9': (crop(scans['s000000'], 9.0), crop(scans['s000004'], 9.0), 0),
        'pair04': (scans['s000000'], scans['s000004'], 0),
        'pair07': (scans['s000000'], scans['s000007'], 0),
        'pair04_seed1': (scans['s000000'], scans['s000004'], 1),
        'synth0': (synth['ref0'], synth['src0'], 0),   # BASELINE configs[1] shape; the reference's own 8- and 1-thread
        'synth3': (synth['ref3'], synth['src3'], 0),   # runs differ in a few point correspondences on 0, in nothing on 3
    }
    # BASELINE configs[4] WORKLOAD (Mulran-shaped low overlap: 70 deg of the second scan's field of view removed, >= 10 m
    # apart, arbitrary yaw) through the reference with the vote layer ON (model_infer.py:179-246 runs in that mode; only
    # `inference_use_vote=False`, infer.py:119-120, breaks it).  The scans are stored in the fixture itself.
    if not only_has_no('lowoverlap'):
        from rdmnet_amd import synthetic
        lo_ref, lo_src, _ = synthetic.make_low_overlap_pair(0)
        cases['lowoverlap'] = (lo_ref, lo_src, 0)
    # BASELINE configs[3] WORKLOAD (KITTI-360 / Apollo-shaped: denser scans, ~20 k points each) through the reference in fp32 --
    # the reference has no bf16 switch; the bf16 mode's deviation from THIS run is measured by test_reference_goldens_gpu.py
    if not only_has_no('dense20k'):
        from rdmnet_amd import synthetic
        d_ref, d_src, _ = synthetic.make_pair(40, target_points=20000)
        cases['dense20k'] = (d_ref, d_src, 0)
    for i in os.environ.get('RDM_GOLDEN_EXTRA_SYNTH', '').split():  # probing other synthetic pairs, not committed
        cases[f'synth{i}'] = (synth[f'ref{i}'], synth[f'src{i}'], 0)
    only = [a for a in sys.argv[1:] if a in cases]
    assert len(only) == len(sys.argv) - 1, sys.argv
    report = {}
    rep_path = os.path.join(HERE, 'oracle_vs_reference.json')
    if only and os.path.exists(rep_path):
        report = json.load(open(rep_path))
    loaded_seed = None
    for tag, (rp, sp, seed) in cases.items():
        if only and tag not in only:
            continue
        if seed != loaded_seed:
            W = load_seed(seed)
            loaded_seed = seed
        torch.set_num_threads(8)
        data, out, taps = run_reference(cfg, model, rp, sp)
        fx = {'ref_points_in': rp, 'src_points_in': sp, 'weight_seed': np.int64(seed)}
        for i in range(5):
            fx[f'lengths{i}'] = np_(data['lengths'][i])
        for n, v in taps.items():
            if n.startswith('encoder.'):
                fx[f'tap/{n}'] = sample(np_(v))
        fx['tap/t1_ref'], fx['tap/t1_src'] = sample(np_(taps['transformer'][0][0])), sample(np_(taps['transformer'][1][0]))
        fx['tap/t2_ref'], fx['tap/t2_src'] = sample(np_(taps['transformer2'][0][0])), sample(np_(taps['transformer2'][1][0]))
        fx['tap/vote_xyz'], fx['tap/vote_feats'] = np_(taps['vote'][0]), sample(np_(taps['vote'][1]))
        fx['tap/nms_mask'] = np_(taps['nms'])
        fx['tap/decoder'] = sample(np_(taps['decoder'][0]))
        fx['tap/node_corr_scores'] = np_(taps['coarse_matching'][2])
        for k, v in out.items():
            a = np_(v)
            if k in ('ref_points', 'src_points', 'ref_points_f', 'src_points_f'):
                continue  # inputs / pyramid levels
            if k in ('ref_feats_f', 'src_feats_f', 'ref_p2p_scores_c', 'src_p2p_scores_c', 'ref_feats_c', 'src_feats_c'):
                a = sample(a)
            if k == 'matching_scores':  # only the un-masked blocks carry information
                a = compact_scores(a, np_(out['ref_node_corr_knn_masks']), np_(out['src_node_corr_knn_masks']))
            fx['out/' + k] = a
        # un-sampled inputs of the coarse matching stage (superpoint_matching.py:14-83), for a teacher-forced check of the
        # top-k ORDER: L2-normalised superpoint features and the non-empty-node masks of the reference's own grouping
        from geotransformer.modules.ops import point_to_node_partition
        n_f0 = int(data['lengths'][1][0])
        fx['full/ref_feats_c'], fx['full/src_feats_c'] = np_(out['ref_feats_c']), np_(out['src_feats_c'])
        fx['full/ref_node_masks'] = np_(point_to_node_partition(data['points'][1][:n_f0], out['ref_points_c'],
                                                                my_cfg.model.num_points_in_patch)[1])
        fx['full/src_node_masks'] = np_(point_to_node_partition(data['points'][1][n_f0:], out['src_points_c'],
                                                                my_cfg.model.num_points_in_patch)[1])

        # ---- pin the oracle: replay the restatement on the same inputs, record deviations
        odata = ofw.pyramid(np.concatenate([rp, sp]), np.array([len(rp), len(sp)], np.int64), my_cfg)
        rep = {}
        for key in ('points', 'lengths', 'neighbors', 'subsampling', 'upsampling'):
            rep['pyramid/' + key] = bool(all(torch.equal(a, b) for a, b in zip(odata[key], data[key])))
        otaps = {}
        oout = ofw.forward(W, my_cfg, odata, otaps)

        def dev(a, b):
            a, b = np_(a).astype(np.float64), np_(b).astype(np.float64)
            if a.shape != b.shape:
                return {'shape_mismatch': [list(a.shape), list(b.shape)]}
            if a.size == 0:
                return {'max_abs': 0.0}
            fin = np.isfinite(a) & np.isfinite(b) & (np.abs(a) < 1e11)
            return {'max_abs': float(np.abs(a - b)[fin].max()) if fin.any() else 0.0,
                    'max_ref': float(np.abs(a[fin]).max()) if fin.any() else 0.0}

        for n, v in taps.items():
            if n.startswith('encoder.'):
                rep['tap/' + n] = dev(v, otaps[n])
        rep['tap/t1_ref'] = dev(taps['transformer'][0][0], otaps['t1_ref'])
        rep['tap/t2_ref'] = dev(taps['transformer2'][0][0], otaps['t2_ref'])
        rep['tap/vote_xyz'] = dev(taps['vote'][0], otaps['vote_xyz'])
        rep['tap/nms_mask_equal'] = bool(np.array_equal(np_(taps['nms']), np_(otaps['nms_mask'])))
        for k, v in out.items():
            a, b = np_(v), np_(oout[k])
            if a.dtype.kind in 'ib':
                rep['out/' + k + '_equal'] = bool(a.shape == b.shape and np.array_equal(a, b))
            else:
                rep['out/' + k] = dev(a, b)
        # teacher-forced stage pins: each oracle stage fed the REFERENCE's own stage inputs
        n_f = int(data['lengths'][1][0])
        k_pts = my_cfg.model.num_points_in_patch
        tf = {}
        for side, lo, hi in (('ref', 0, n_f), ('src', n_f, None)):
            _, nm, knn, km = ofw.point_to_node(data['points'][1][lo:hi], out[f'{side}_points_c'], k_pts)
            sel = out[f'{side}_node_corr_indices']
            tf[f'point_to_node/{side}_masks_equal'] = bool(torch.equal(km[sel], out[f'{side}_node_corr_knn_masks']))
            pad = torch.cat([data['points'][1][lo:hi], torch.zeros(1, 3)], 0)
            tf[f'point_to_node/{side}_knn_points_equal'] = bool(torch.equal(pad[knn[sel]], out[f'{side}_node_corr_knn_points']))
            tf[f'_{side}_nm'] = nm
        ri, si, sc = ofw.coarse_matching(out['ref_feats_c'], out['src_feats_c'], tf.pop('_ref_nm'), tf.pop('_src_nm'),
                                         my_cfg.coarse_matching.num_correspondences)
        tf['coarse/indices_equal'] = bool(torch.equal(ri, out['ref_node_corr_indices']) and torch.equal(si, out['src_node_corr_indices']))
        tf['coarse/scores'] = dev(sc, taps['coarse_matching'][2])
        rf_f, sf_f = out['ref_feats_f'], out['src_feats_f']
        r_idx = ofw.point_to_node(data['points'][1][:n_f], out['ref_points_c'], k_pts)[2][out['ref_node_corr_indices']]
        s_idx = ofw.point_to_node(data['points'][1][n_f:], out['src_points_c'], k_pts)[2][out['src_node_corr_indices']]
        pscores = torch.einsum('bnd,bmd->bnm', torch.cat([rf_f, torch.zeros(1, rf_f.shape[1])], 0)[r_idx],
                               torch.cat([sf_f, torch.zeros(1, sf_f.shape[1])], 0)[s_idx]) / rf_f.shape[1] ** 0.5
        ms = ofw.sinkhorn(pscores, out['ref_node_corr_knn_masks'], out['src_node_corr_knn_masks'],
                          W['optimal_transport.alpha'], my_cfg.model.num_sinkhorn_iterations)
        tf['sinkhorn/matching_scores'] = dev(ms, out['matching_scores'])
        lgr_in = (out['ref_node_corr_knn_points'], out['src_node_corr_knn_points'], out['ref_node_corr_knn_masks'],
                  out['src_node_corr_knn_masks'], out['matching_scores'], my_cfg)
        rc, sc2, cs, T, linfo = ofw.lgr(*lgr_in)
        # margin of the reference's own hypothesis selection (local_global_registration.py:204-221), and the poses it
        # would return if a near-tie (within one inlier of the best) fell the other way
        if 'inlier_counts' in linfo:
            counts = linfo['inlier_counts'].numpy()
            near = [int(i) for i in np.nonzero(counts >= counts.max() - 1)[0]]
            fx['lgr/inlier_counts'] = counts.astype(np.int64)
            fx['lgr/best'] = np.int64(linfo['best'])
            fx['lgr/alt_hypotheses'] = np.asarray(near, np.int64)
            fx['lgr/alt_transforms'] = np.stack([np_(ofw.lgr(*lgr_in, force_best=i)[3]) for i in near])
            top = np.sort(counts)[::-1]
            tf['lgr/inlier_margin'] = int(top[0] - top[1]) if len(top) > 1 else int(top[0])
        tf['lgr/corr_points_equal'] = bool(torch.equal(rc, out['ref_corr_points']) and torch.equal(sc2, out['src_corr_points']))
        tf['lgr/corr_scores'] = dev(cs, out['corr_scores'])
        rre_t, rte_t = ofw.rre_rte(np_(T), np_(out['estimated_transform']))
        tf['lgr/pose'] = {'rre_deg': rre_t, 'rte_m': rte_t}
        rep['teacher_forced'] = tf
        rre, rte = ofw.rre_rte(np_(oout['estimated_transform']), np_(out['estimated_transform']))
        rep['pose'] = {'rre_deg': rre, 'rte_m': rte, 'n_corr': int(out['corr_scores'].shape[0]),
                       'n_hypotheses': len(otaps['lgr']['chunks'])}
        # the reference against itself at another thread count: its own fp32 noise floor
        torch.set_num_threads(1)
        _, out1, taps1 = run_reference(cfg, model, rp, sp)
        rre1, rte1 = ofw.rre_rte(np_(out1['estimated_transform']), np_(out['estimated_transform']))
        pairs8 = set(zip(np_(out['ref_node_corr_indices']).tolist(), np_(out['src_node_corr_indices']).tolist()))
        pairs1 = set(zip(np_(out1['ref_node_corr_indices']).tolist(), np_(out1['src_node_corr_indices']).tolist()))
        # what the reference decides differently against ITSELF at another thread count bounds what an end-to-end
        # comparison can demand of the discrete outputs of this case (tests read these flags; nothing is skipped silently)
        fx['self/node_corr_symmetric_difference'] = np.int64(len(pairs8 ^ pairs1))
        fx['self/nms_mask_equal'] = np.bool_(np.array_equal(np_(taps['nms']), np_(taps1['nms'])))
        fx['self/transform_1_thread'] = np_(out1['estimated_transform'])
        fx['self/n_corr_1_thread'] = np.int64(out1['corr_scores'].shape[0])
        fx['self/corr_symmetric_difference'] = np.int64(corr_symmetric_difference(out1, out))
        rep['reference_8_vs_1_thread'] = {
            'rre_deg': rre1, 'rte_m': rte1, 'node_corr_symmetric_difference': len(pairs8 ^ pairs1),
            'corr_equal': bool(out1['ref_corr_points'].shape == out['ref_corr_points'].shape
                               and torch.equal(out1['ref_corr_points'], out['ref_corr_points'])),
            'corr_symmetric_difference': corr_symmetric_difference(out1, out),
            'feats_f_max_abs': float((out1['ref_feats_f'] - out['ref_feats_f']).abs().max()),
            'matching_scores': dev(out1['matching_scores'], out['matching_scores'])}
        torch.set_num_threads(8)
        np.savez_compressed(os.path.join(HERE, f'forward_{tag}.npz'), **fx)
        report[tag] = rep
        print(tag, json.dumps(rep['pose']), json.dumps(rep['reference_8_vs_1_thread']))
        print('  teacher-forced:', json.dumps(tf))
    with open(rep_path, 'w') as f:
        json.dump(report, f, indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
