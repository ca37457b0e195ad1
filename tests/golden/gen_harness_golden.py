"""Golden vectors for the callers either side of the path: neighbour-limit
calibration, the KITTI pair loader, the pose/npz output format and the evaluation metrics.

Runs ONLY in the build container (imports /root/reference through ref_import's shims):

    python tests/golden/gen_harness_golden.py        # writes tests/golden/harness.npz

Everything stored is data produced by the reference's own functions:
  * calibrate_neighbors_stack_mode on the bundled 'infer' set, for several keep ratios
    (geotransformer/utils/data.py:195-220);
  * OdometryKittiPairDataset('infer') items and load_kitti_gt_txt on a synthetic GT file
    (rdmnet/datasets/registration/kitti/dataset.py:16-75,138-191);
  * Tester.after_test_step's pose line and .npz key set (experiments/infer.py:62-110), with the
    Open3D RANSAC call (not under /root/reference) replaced by identity;
  * compute_registration_error / evaluate_correspondences / evaluate_sparse_correspondences
    (geotransformer/utils/registration.py:17-108,175-200,354-402) on the pair-04 golden outputs.
"""
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402


def main():
    ref_import.install()
    import torch
    tb = types.ModuleType('torch.utils.tensorboard')
    tb.SummaryWriter = object
    sys.modules['torch.utils.tensorboard'] = tb
    for name in ('tensorboardX', 'nibabel', 'pykitti'):
        sys.modules.setdefault(name, types.ModuleType(name))
    out = {}

    cfg = ref_import.make_cfg()
    from geotransformer.utils.data import calibrate_neighbors_stack_mode, registration_collate_fn_stack_mode
    from rdmnet.datasets.registration.kitti.dataset import OdometryKittiPairDataset, load_kitti_gt_txt

    ds = OdometryKittiPairDataset(cfg.data.dataset_root if 'data' in cfg else '.', 'infer', point_limit=None)
    out['infer_len'] = np.int64(len(ds))
    for i in range(len(ds)):
        item = ds[i]
        out[f'item{i}_keys'] = np.array(sorted(item.keys()))
        out[f'item{i}_frames'] = np.array([item['seq_id'], item['ref_frame'], item['src_frame']], np.int64)
        out[f'item{i}_sizes'] = np.array([item['ref_points'].shape[0], item['src_points'].shape[0]], np.int64)
        out[f'item{i}_checksum'] = np.array([item['ref_points'].astype(np.float64).sum(),
                                             item['src_points'].astype(np.float64).sum()])
    ratios = [0.5, 0.8, 0.9, 0.99]
    out['calib_keep_ratios'] = np.array(ratios)
    b = cfg.backbone
    for thr, tag in ((2000, 'default'), (10 ** 9, 'all')):
        lim = [calibrate_neighbors_stack_mode(ds, registration_collate_fn_stack_mode, b.num_stages, b.init_voxel_size,
                                              b.init_radius, keep_ratio=r, sample_threshold=thr) for r in ratios]
        out[f'calib_limits_{tag}'] = np.asarray(lim, np.int64)
    out['calib_hist_n'] = np.int64(int(np.ceil(4 / 3 * np.pi * (b.init_radius / b.init_voxel_size + 1) ** 3)))

    # GT list parser
    rng = np.random.default_rng(5)
    lines = []
    for k in range(3):
        T = rng.normal(size=(3, 4))
        lines.append('%d %d ' % (10 * k + 3, 10 * k) + ' '.join('%.9e' % v for v in T.reshape(-1)))
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, '08'), 'w') as f:
            f.write('\n'.join(lines) + '\n')
        meta = load_kitti_gt_txt(d, 8)
    out['gt_txt'] = np.array('\n'.join(lines) + '\n')
    out['gt_meta_frames'] = np.array([[m['seq_id'], m['frame0'], m['frame1']] for m in meta], np.int64)
    out['gt_meta_transforms'] = np.stack([m['transform'] for m in meta])

    # output format (pose line + npz keys) from the reference's own after_test_step
    g = np.load(os.path.join(HERE, 'forward_pair04.npz'))
    keys = ['ref_points', 'src_points', 'ref_points_f', 'src_points_f', 'ref_points_c', 'src_points_c', 'ref_feats_c',
            'src_feats_c', 'ref_node_corr_indices', 'src_node_corr_indices', 'ref_corr_points', 'src_corr_points',
            'estimated_transform']
    rng = np.random.default_rng(6)
    output_dict = {}
    for k in keys:
        if 'out/' + k in g.files and g['out/' + k].ndim == 2 and k.endswith(('corr_points', 'transform', 'points_c')):
            output_dict[k] = torch.from_numpy(g['out/' + k])
        elif k.endswith('indices'):
            output_dict[k] = torch.from_numpy(g['out/' + k])
        else:
            output_dict[k] = torch.from_numpy(rng.normal(size=(7, 3)).astype(np.float32))
    import geotransformer.utils.open3d as ref_o3d
    ref_o3d.registration_with_ransac_from_correspondences = lambda *a, **k: np.eye(4)
    import infer as ref_infer
    with tempfile.TemporaryDirectory() as d:
        fake = types.SimpleNamespace(output_dir=d)
        ref_infer.Tester.after_test_step(fake, 0, {'seq_id': 8, 'ref_frame': 12, 'src_frame': 34}, output_dict, None)
        out['pose_file_name'] = np.array([n for n in sorted(os.listdir(d)) if n.endswith('pose')][0])
        out['pose_line'] = np.array(open(os.path.join(d, '08_pose')).read())
        npz_name = [n for n in os.listdir(d) if n.endswith('.npz')][0]
        out['npz_file_name'] = np.array(npz_name)
        out['npz_keys'] = np.array(sorted(np.load(os.path.join(d, npz_name)).files))

    # metrics on the pair-04 golden correspondences against a perturbed ground truth
    from geotransformer.utils.registration import (compute_registration_error, evaluate_correspondences,
                                                   evaluate_sparse_correspondences)
    import geotransformer.utils.pointcloud as ref_pc
    from scipy.spatial import cKDTree

    class _Tree(cKDTree):  # the reference targets a scipy whose query() still takes n_jobs
        def query(self, x, k=1, n_jobs=None, **kw):
            return super().query(x, k=k, workers=-1 if n_jobs == -1 else 1, **kw)
    ref_pc.cKDTree = _Tree
    est = g['out/estimated_transform'].astype(np.float64)
    from scipy.spatial.transform import Rotation
    dR = Rotation.from_euler('xyz', [0.3, -0.45, 0.7], degrees=True).as_matrix()
    gt = est.copy()
    gt[:3, :3] = dR @ est[:3, :3]
    gt[:3, 3] = est[:3, 3] + np.array([0.11, -0.05, 0.02])
    out['metric_gt_transform'] = gt
    out['metric_registration_error'] = np.array(compute_registration_error(gt, est), np.float64)
    # correspondences with a controlled residual spectrum (random-weight ones are all outliers)
    src_c = g['out/src_corr_points'].astype(np.float64)
    noise = rng.normal(size=src_c.shape) * rng.uniform(0.0, 0.6, size=(src_c.shape[0], 1))
    ref_c = (src_c @ gt[:3, :3].T + gt[:3, 3] + noise).astype(np.float32)
    src_c = src_c.astype(np.float32)
    out['metric_ref_corr_points'], out['metric_src_corr_points'] = ref_c, src_c
    fine = evaluate_correspondences(ref_c, src_c, gt, positive_radius=cfg.eval.acceptance_radius)
    out['metric_fine_keys'] = np.array(sorted(fine.keys()))
    out['metric_fine_values'] = np.array([float(fine[k]) for k in sorted(fine.keys())])
    ref_nodes, src_nodes = g['out/ref_points_c'], g['out/src_points_c']
    ri, si = g['out/ref_node_corr_indices'], g['out/src_node_corr_indices']
    gt_pairs = np.stack([np.concatenate([ri[::3], rng.integers(0, ref_nodes.shape[0], 40)]),
                         np.concatenate([si[::3], rng.integers(0, src_nodes.shape[0], 40)])], 1)
    out['metric_gt_node_corr_indices'] = gt_pairs
    coarse = evaluate_sparse_correspondences(ref_nodes, src_nodes, ri, si, gt_pairs)
    out['metric_coarse_keys'] = np.array(sorted(coarse.keys()))
    out['metric_coarse_values'] = np.array([float(coarse[k]) for k in sorted(coarse.keys())])
    out['eval_thresholds'] = np.array([cfg.eval.acceptance_radius, cfg.eval.inlier_ratio_threshold,
                                       cfg.eval.rre_threshold, cfg.eval.rte_threshold])

    np.savez_compressed(os.path.join(HERE, 'harness.npz'), **out)
    for k, v in out.items():
        print(k, v if v.size < 24 else v.shape)


if __name__ == '__main__':
    main()
