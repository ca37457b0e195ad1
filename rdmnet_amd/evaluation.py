"""Output formats and registration metrics of the inference harness.

Host-side numpy, as in the reference (these run once per pair on a few hundred correspondences):
  * `pose_line`, `pose_file_name`, `npz_file_name`, `save_pair_npz`: experiments/infer.py:62-110
    (`'%02d_pose'` files with `ref src` + 12 floats `%.6f` each followed by a blank, and one
    `{seq}_{src}_{ref}.npz` per pair);
  * `compute_registration_error` & co: geotransformer/utils/registration.py:17-108;
  * `evaluate_correspondences`, `evaluate_sparse_correspondences`: registration.py:175-200, 354-402;
  * `Summary`: the meters and report lines of experiments/eval.py:36-286 (method 'lgr' and 'svd').
"""
import math
import os.path as osp

import numpy as np
from scipy.spatial import cKDTree

NPZ_KEYS = ('ref_points', 'src_points', 'ref_points_f', 'src_points_f', 'ref_points_c', 'src_points_c', 'ref_feats_c',
            'src_feats_c', 'ref_node_corr_indices', 'src_node_corr_indices', 'ref_corr_points', 'src_corr_points',
            'estimated_transform')


def pose_file_name(seq_id):
    return '%02d_pose' % seq_id


def pose_line(ref_frame, src_frame, estimated_transform):
    """infer.py:73-75: first 12 entries of the row-major 4x4, '%.6f', each followed by one blank."""
    m = np.asarray(estimated_transform).reshape(-1)[:12]
    return f'{ref_frame} {src_frame} ' + ''.join(f'{v:.6f} ' for v in m) + '\n'


def npz_file_name(seq_id, src_frame, ref_frame):
    return f'{seq_id}_{src_frame}_{ref_frame}.npz'


def save_pair_npz(output_dir, data_dict, output_dict, estimated_transform_ransac=None, extra=None):
    """infer.py:84-101.  output_dict values may be tensors or arrays.  `estimated_transform_ransac` is
    Open3D's RANSAC result in the reference (not part of this path); identity unless given.
    `corr_scores` (read by eval.py:111 but never written by infer.py) and the ground truth, when the
    loader provided it, are stored as additional keys."""
    def host(v):
        return v.detach().cpu().numpy() if hasattr(v, 'detach') else np.asarray(v)
    arrays = {k: host(output_dict[k]) for k in NPZ_KEYS}
    arrays['estimated_transform_ransac'] = np.eye(4) if estimated_transform_ransac is None else estimated_transform_ransac
    if 'corr_scores' in output_dict:
        arrays['corr_scores'] = host(output_dict['corr_scores'])
    if 'transform' in data_dict:
        arrays['transform'] = host(data_dict['transform'])
    arrays.update(extra or {})
    name = osp.join(output_dir, npz_file_name(data_dict['seq_id'], data_dict['src_frame'], data_dict['ref_frame']))
    np.savez_compressed(name, **arrays)
    return name


def append_pose(output_dir, data_dict, estimated_transform):
    with open(osp.join(output_dir, pose_file_name(data_dict['seq_id'])), 'a') as f:
        f.write(pose_line(data_dict['ref_frame'], data_dict['src_frame'], estimated_transform))


# ---- metrics (geotransformer/utils/registration.py) -------------------------------------------------------------

def apply_transform(points, transform):
    """geotransformer/utils/pointcloud.py apply_transform: p R^T + t."""
    return np.matmul(points, transform[:3, :3].T) + transform[:3, 3]


def compute_relative_rotation_error(gt_rotation, est_rotation):
    x = 0.5 * (np.trace(np.matmul(est_rotation.T, gt_rotation)) - 1.0)
    return 180.0 * np.arccos(np.clip(x, -1.0, 1.0)) / np.pi


def rotation_matrix_to_euler_angles(R):
    """registration.py:36-53 (degrees; the reference divides by a literal pi)."""
    sy = math.sqrt(R[0, 0] * R[0, 0] + R[1, 0] * R[1, 0])
    if sy >= 1e-6:
        x, y, z = math.atan2(R[2, 1], R[2, 2]), math.atan2(-R[2, 0], sy), math.atan2(R[1, 0], R[0, 0])
    else:
        x, y, z = math.atan2(-R[1, 2], R[1, 1]), math.atan2(-R[2, 0], sy), 0
    k = 180.0 / 3.141592653589793
    return x * k, y * k, z * k


def compute_relative_translation_error(gt_translation, est_translation):
    return np.linalg.norm(gt_translation - est_translation)


def compute_registration_error(gt_transform, est_transform):
    """-> (rre_deg, rte, |droll|, |dpitch|, |dyaw|)  (registration.py:91-108)."""
    gt_r, gt_t = gt_transform[:3, :3], gt_transform[:3, 3]
    est_r, est_t = est_transform[:3, :3], est_transform[:3, 3]
    rre = compute_relative_rotation_error(gt_r, est_r)
    g, e = rotation_matrix_to_euler_angles(gt_r), rotation_matrix_to_euler_angles(est_r)
    rte = compute_relative_translation_error(gt_t, est_t)
    return rre, rte, np.absolute(g[0] - e[0]), np.absolute(g[1] - e[1]), np.absolute(g[2] - e[2])


def get_nearest_neighbor(q_points, s_points):
    """registration.py: cKDTree(...).query(k=1, n_jobs=-1).  Same distances; one thread below 50 000 queries -- `workers=-1`
    starts a thread per host CPU, which for the few hundred correspondences of a pair cost 80 ms on a 16-CPU-quota container
    (the harness measured 46 pairs/s with it, every worker thread throttled)."""
    return cKDTree(s_points).query(q_points, k=1, workers=-1 if len(q_points) >= 50000 else 1)[0]


def compute_correspondence_residual(ref_corr_points, src_corr_points, transform):
    src = apply_transform(src_corr_points, transform)
    return np.mean(np.sqrt(((ref_corr_points - src) ** 2).sum(1)))


def compute_inlier_ratio(ref_corr_points, src_corr_points, transform, positive_radius=0.1):
    src = apply_transform(src_corr_points, transform)
    return np.mean(np.sqrt(((ref_corr_points - src) ** 2).sum(1)) < positive_radius)


def compute_overlap(ref_points, src_points, transform=None, positive_radius=0.1):
    if transform is not None:
        src_points = apply_transform(src_points, transform)
    return np.mean(get_nearest_neighbor(ref_points, src_points) < positive_radius)


def evaluate_correspondences(ref_points, src_points, transform, positive_radius=0.1):
    return {
        'overlap': compute_overlap(ref_points, src_points, transform, positive_radius=positive_radius),
        'inlier_ratio': compute_inlier_ratio(ref_points, src_points, transform, positive_radius=positive_radius),
        'inlier_ratio_0.3': compute_inlier_ratio(ref_points, src_points, transform, positive_radius=0.3),
        'inlier_ratio_0.1': compute_inlier_ratio(ref_points, src_points, transform, positive_radius=0.1),
        'residual': compute_correspondence_residual(ref_points, src_points, transform),
        'num_corr': ref_points.shape[0],
    }


def evaluate_sparse_correspondences(ref_points, src_points, ref_corr_indices, src_corr_indices, gt_corr_indices):
    gt = np.zeros((ref_points.shape[0], src_points.shape[0]))
    gt[gt_corr_indices[:, 0], gt_corr_indices[:, 1]] = 1.0
    pred = np.zeros_like(gt)
    pred[ref_corr_indices, src_corr_indices] = 1.0
    pos = gt * pred
    precision = pos.sum() / (pred.sum() + 1e-12)
    recall = pos.sum() / (gt.sum() + 1e-12)
    pos, gt = pos > 0, gt > 0
    ref_hit = np.any(pos, axis=1).sum() / (np.any(gt, axis=1).sum() + 1e-12)
    src_hit = np.any(pos, axis=0).sum() / (np.any(gt, axis=0).sum() + 1e-12)
    return {'precision': precision, 'recall': recall, 'hit_ratio': 0.5 * (ref_hit + src_hit)}


class Summary:
    """Accumulates what eval.py's meters accumulate and prints its report lines.  Thresholds default to
    the reference's config (experiments/config.py:63-67)."""

    def __init__(self, acceptance_radius=0.6, inlier_ratio_threshold=0.05, rre_threshold=5.0, rte_threshold=2.0):
        self.acceptance_radius, self.inlier_ratio_threshold = acceptance_radius, inlier_ratio_threshold
        self.rre_threshold, self.rte_threshold = rre_threshold, rte_threshold
        self.meters = {}
        self.fail_case = []

    def _update(self, name, value):
        self.meters.setdefault(name, []).append(float(value))

    def mean(self, name):
        v = self.meters.get(name, [])
        return float(np.mean(v)) if v else 0.0  # SummaryBoard returns 0 for an empty meter

    def std(self, name):
        v = self.meters.get(name, [])
        return float(np.std(v)) if v else 0.0

    def measure(self, gt_transform, est_transform, ref_corr_points=None, src_corr_points=None, corr_scores=None, nodes=None):
        """The per-pair numbers of `update` as a pure function (no state touched): safe on the worker threads of a
        rdmnet_amd.pipeline.PairPipeline, where the kd-tree of the overlap and the matrix products of several pairs then run
        side by side; `commit` adds them to the meters in dataset order."""
        m = {}
        if nodes is not None:
            m['precision'] = evaluate_sparse_correspondences(*nodes)['precision']
        if ref_corr_points is not None and len(ref_corr_points):
            f = evaluate_correspondences(ref_corr_points, src_corr_points, gt_transform, self.acceptance_radius)
            f['n'] = len(ref_corr_points) if corr_scores is None else corr_scores.shape[0]
            m['fine'] = f
        m['registration'] = compute_registration_error(np.asarray(gt_transform, np.float64), np.asarray(est_transform, np.float64))
        return m

    def commit(self, ids, m):
        """Adds one pair's `measure` result to the meters.  Returns the per-pair dict of `update`."""
        out = {}
        if 'precision' in m:
            c = m['precision']
            out['c_PIR'] = c
            self._update('precision', c)
            for tag, ok in (('PMR>0', c > 0), ('PMR>=0.1', c >= 0.1), ('PMR>=0.3', c >= 0.3), ('PMR>=0.5', c >= 0.5)):
                self._update(tag, float(ok))
        if 'fine' in m:
            f = m['fine']
            for k in ('inlier_ratio', 'inlier_ratio_0.3', 'inlier_ratio_0.1', 'overlap'):
                self._update(k, f[k])
            self._update('fine_recall', float(f['inlier_ratio'] >= self.inlier_ratio_threshold))
            self._update('num_corr', f['n'])
            out.update(f_IR=f['inlier_ratio'], f_OV=f['overlap'], f_RS=f['residual'], f_NU=f['num_corr'])
        rre, rte, rx, ry, rz = m['registration']
        accepted = bool(rre < self.rre_threshold and rte < self.rte_threshold)
        if accepted:
            for k, v in (('rre', rre), ('rte', rte), ('x', rx), ('y', ry), ('z', rz)):
                self._update(k, v)
        else:
            self.fail_case.append(list(ids))
        self._update('recall', float(accepted))
        out.update(r_RRE=rre, r_RTE=rte, accepted=accepted)
        return out

    def update(self, ids, gt_transform, est_transform, ref_corr_points=None, src_corr_points=None, corr_scores=None,
               nodes=None):
        """ids = (seq_id, src_frame, ref_frame).  nodes = (ref_nodes, src_nodes, ref_idx, src_idx,
        gt_node_corr_indices) enables the coarse-matching meters.  Returns the per-pair dict."""
        return self.commit(ids, self.measure(gt_transform, est_transform, ref_corr_points, src_corr_points, corr_scores, nodes))

    def lines(self):
        m = self.mean
        return [
            '  Coarse Matching, PIR: {:.3f}, PMR>0: {:.3f}, PMR>=0.1: {:.3f}, PMR>=0.3: {:.3f}, PMR>=0.5: {:.3f}'.format(
                m('precision'), m('PMR>0'), m('PMR>=0.1'), m('PMR>=0.3'), m('PMR>=0.5')),
            '  Fine Matching, FMR: {:.4f}, IR: {:.3f}, IR_0.3: {:.3f}, IR_0.1: {:.3f}, num_Corr: {:.3f}, OV: {:.3f}, '
            'std: {:.3f}'.format(m('fine_recall'), m('inlier_ratio'), m('inlier_ratio_0.3'), m('inlier_ratio_0.1'),
                                 m('num_corr'), m('overlap'), self.std('fine_recall')),
            '  Registration, RR: {:.4f}, RRE: {:.3f}, RTE: {:.3f}, Rx: {:.3f}, Ry: {:.3f}, Rz: {:.3f}'.format(
                m('recall'), m('rre'), m('rte'), m('x'), m('y'), m('z')),
        ]
