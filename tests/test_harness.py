"""CPU: the callers either side of the path against goldens produced by the
reference's own functions (tests/golden/gen_harness_golden.py): GT-list parser, loader items, pose line
and .npz format, registration / correspondence metrics."""
import os

import numpy as np
import pytest


@pytest.fixture(scope='module')
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, 'harness.npz'))


def test_gt_list_parser_matches_reference(gold, tmp_path):
    from rdmnet_amd import dataset
    (tmp_path / '08').write_text(str(gold['gt_txt']))
    meta = dataset.load_kitti_gt_txt(str(tmp_path), 8)
    assert np.array_equal(np.array([[m['seq_id'], m['frame0'], m['frame1']] for m in meta]), gold['gt_meta_frames'])
    assert np.array_equal(np.stack([m['transform'] for m in meta]), gold['gt_meta_transforms'])
    with pytest.raises(Exception):
        dataset.make_dataset_kitti(str(tmp_path), 'bogus')


def test_infer_dataset_items_match_reference(gold, scans, tmp_path):
    from rdmnet_amd import dataset
    for name, pts in scans.items():
        xyzi = np.concatenate([pts, np.zeros((len(pts), 1), pts.dtype)], 1)  # the .npy scans are [N,4]
        np.save(tmp_path / (name[1:] + '.npy'), xyzi)
    ds = dataset.OdometryKittiPairDataset('.', 'infer', infer_root=str(tmp_path))
    assert len(ds) == int(gold['infer_len'])
    for i in range(len(ds)):
        item = ds[i]
        assert sorted(item.keys()) == list(gold[f'item{i}_keys'])
        assert [item['seq_id'], item['ref_frame'], item['src_frame']] == list(gold[f'item{i}_frames'])
        assert [len(item['ref_points']), len(item['src_points'])] == list(gold[f'item{i}_sizes'])
        sums = [item['ref_points'].astype(np.float64).sum(), item['src_points'].astype(np.float64).sum()]
        assert np.array_equal(np.array(sums), gold[f'item{i}_checksum'])
        assert item['ref_points'].dtype == np.float32 and item['ref_feats'].shape == (len(item['ref_points']), 1)


def test_test_subset_reads_transform_and_point_limit(tmp_path):
    from rdmnet_amd import dataset
    root = tmp_path
    (root / 'icp10').mkdir()
    for seq in (8, 9, 10):
        T = np.eye(4)[:3].reshape(-1)
        (root / 'icp10' / ('%02d' % seq)).write_text('7 3 ' + ' '.join('%.6f' % v for v in T) + '\n')
        d = root / 'downsampled_xyzi' / ('%02d' % seq)
        d.mkdir(parents=True)
        rng = np.random.default_rng(seq)
        np.save(d / '000003.npy', rng.normal(size=(50, 4)).astype(np.float32))
        np.save(d / '000007.npy', rng.normal(size=(40, 4)).astype(np.float32))
    ds = dataset.OdometryKittiPairDataset(str(root), 'test', point_limit=45)
    assert len(ds) == 3
    item = ds[1]
    assert item['seq_id'] == 9 and item['ref_frame'] == 3 and item['src_frame'] == 7
    assert item['transform'].dtype == np.float32 and item['transform'].shape == (4, 4)
    assert item['ref_points'].shape == (45, 3) and item['src_points'].shape == (40, 3)


def test_pose_line_and_npz_format_match_reference(gold, tmp_path):
    from rdmnet_amd import evaluation as ev
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'forward_pair04.npz'))
    T = g['out/estimated_transform']
    assert ev.pose_file_name(8) == str(gold['pose_file_name'])
    assert ev.pose_line(12, 34, T) == str(gold['pose_line'])
    assert ev.npz_file_name(8, 34, 12) == str(gold['npz_file_name'])
    out = {k: np.zeros((3, 3), np.float32) for k in ev.NPZ_KEYS}
    out['estimated_transform'] = T
    out['corr_scores'] = np.zeros(3, np.float32)
    item = {'seq_id': 8, 'ref_frame': 12, 'src_frame': 34}
    name = ev.save_pair_npz(str(tmp_path), item, out)
    assert os.path.basename(name) == str(gold['npz_file_name'])
    keys = set(np.load(name).files)
    assert set(gold['npz_keys']) <= keys          # every key the reference writes ...
    assert keys - set(gold['npz_keys']) == {'corr_scores'}  # ... plus the one eval.py reads but infer.py forgets
    ev.append_pose(str(tmp_path), item, T)
    ev.append_pose(str(tmp_path), item, T)
    assert open(tmp_path / '08_pose').read() == 2 * str(gold['pose_line'])


def test_registration_and_correspondence_metrics_match_reference(gold):
    from rdmnet_amd import evaluation as ev
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'forward_pair04.npz'))
    gt, est = gold['metric_gt_transform'], g['out/estimated_transform'].astype(np.float64)
    np.testing.assert_allclose(np.array(ev.compute_registration_error(gt, est)), gold['metric_registration_error'],
                               rtol=0, atol=1e-12)
    fine = ev.evaluate_correspondences(gold['metric_ref_corr_points'], gold['metric_src_corr_points'], gt, positive_radius=0.6)
    assert sorted(fine.keys()) == list(gold['metric_fine_keys'])
    np.testing.assert_allclose(np.array([float(fine[k]) for k in sorted(fine)]), gold['metric_fine_values'], rtol=0,
                               atol=1e-12)
    coarse = ev.evaluate_sparse_correspondences(g['out/ref_points_c'], g['out/src_points_c'], g['out/ref_node_corr_indices'],
                                                g['out/src_node_corr_indices'], gold['metric_gt_node_corr_indices'])
    np.testing.assert_allclose(np.array([float(coarse[k]) for k in sorted(coarse)]), gold['metric_coarse_values'], rtol=0,
                               atol=1e-12)


def test_summary_follows_eval_py(gold):
    from rdmnet_amd import evaluation as ev
    thr = gold['eval_thresholds']
    s = ev.Summary(*thr)
    gt = gold['metric_gt_transform']
    good = gt.copy()
    bad = gt.copy()
    bad[:3, 3] += [3.0, 0, 0]  # RTE 3 m > 2 m
    r1 = s.update((8, 1, 0), gt, good, gold['metric_ref_corr_points'], gold['metric_src_corr_points'])
    r2 = s.update((8, 2, 0), gt, bad, gold['metric_ref_corr_points'], gold['metric_src_corr_points'])
    assert r1['accepted'] and not r2['accepted'] and s.fail_case == [[8, 2, 0]]
    assert s.mean('recall') == 0.5 and s.mean('rte') == 0.0  # failures do not enter the RRE/RTE means
    assert abs(s.mean('inlier_ratio') - gold['metric_fine_values'][0]) < 1e-12
    lines = s.lines()
    assert lines[2].startswith('  Registration, RR: 0.5000, RRE: 0.000, RTE: 0.000')
    assert lines[1].startswith('  Fine Matching, FMR: 1.0000')
