"""Inference harness: the counterpart of experiments/infer.py (`Tester`, :19-110) + experiments/eval.py's
registration report, on the native engine.

    python -m rdmnet_amd.infer --infer-root /path/to/assets/pc --out out/           # the two bundled pairs
    python -m rdmnet_amd.infer --dataset-root /data/kitti --subset test --out out/ --weights rdmnet.pth.tar
    python -m torch.distributed.run --nproc-per-node 8 -m rdmnet_amd.infer ...       # pairs sharded over ranks

Per pair it writes what the reference writes: one line in `<seq>_pose` and one `<seq>_<src>_<ref>.npz`
(evaluation.save_pair_npz).  With ground truth in the loader it also prints eval.py's report lines.
Multi-GPU: rank r takes pairs r, r+W, ... (sharding.pairs_for_rank); the only collective is the final
gather of the per-pair records.  `--dataset mulran` switches the vote layer off as infer.py:119-120 does.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

from . import config, dataset as ds_mod, evaluation, ops, sharding, weights
from .engine import Engine


def load_state(path, cfg, seed=0):
    """`state['model']` of a reference checkpoint (base_tester.py:97-107), or synthetic weights."""
    if path:
        state = torch.load(path, map_location='cpu')
        return state['model'] if 'model' in state else state
    return weights.synthetic_state_dict(cfg, seed=seed)


class Tester:
    """One engine on the current device; `run(stager)` processes this rank's pairs."""

    def __init__(self, cfg, state, output_dir=None, save_npz=True, ransac=True, write_poses=True):
        self.cfg, self.output_dir, self.save_npz, self.ransac = cfg, output_dir, save_npz, ransac
        self.write_poses = write_poses  # False under several ranks: rank 0 writes all poses, in pair order, at the end
        self.engine = Engine(cfg, state)
        if save_npz and output_dir:
            self.engine.keep_taps(True)
        if output_dir:
            os.makedirs(output_dir, exist_ok=True)
        self.summary = evaluation.Summary()
        self.records = []

    def output_dict(self, n_ref):
        """The tensors infer.py:84-101 stores, from the engine's taps of the last run."""
        e, r = self.engine, self.engine.result
        lv0, lv1 = e.tensor('points0'), e.tensor('points1')
        nodes, feats = e.tensor('nodes'), e.tensor('feats_c')
        m_r, n_ref_f = int(r.n_ref_nodes), int(r.level_ref_sizes[1])
        rc, sc, cs = e.corr()
        out = {'ref_points': lv0[:n_ref], 'src_points': lv0[n_ref:],
               'ref_points_f': lv1[:n_ref_f], 'src_points_f': lv1[n_ref_f:],
               'ref_points_c': nodes[:m_r], 'src_points_c': nodes[m_r:],
               'ref_feats_c': feats[:m_r], 'src_feats_c': feats[m_r:],
               'ref_node_corr_indices': e.tensor('ref_node_corr_indices')[:, 0],
               'src_node_corr_indices': e.tensor('src_node_corr_indices')[:, 0],
               'ref_corr_points': rc, 'src_corr_points': sc, 'corr_scores': cs,
               'estimated_transform': torch.from_numpy(e.transform())}
        return out

    def step(self, item, ref_dev, src_dev):
        t0 = time.perf_counter()
        res = self.engine.run(ref_dev.contiguous(), src_dev.contiguous())  # returns with the pose on the host
        ms = (time.perf_counter() - t0) * 1e3
        T = self.engine.transform()
        rec = {'seq_id': item['seq_id'], 'ref_frame': item['ref_frame'], 'src_frame': item['src_frame'],
               'n_corr': int(res.n_correspondences), 'ms': ms, 'transform': T}
        if self.output_dir:
            if self.write_poses:
                evaluation.append_pose(self.output_dir, item, T)
            if self.save_npz:
                od = self.output_dict(item['ref_points'].shape[0])
                T_ransac = None
                if self.ransac:  # infer.py:75-82: distance 0.3, ransac_n 4, 50 000 iterations, on the GPU
                    T_ransac = ops.ransac_correspondences(od['src_corr_points'].contiguous(), od['ref_corr_points'].contiguous(),
                                                          0.3, 4, 50000)[0].cpu().numpy().astype(np.float64)
                evaluation.save_pair_npz(self.output_dir, item, od, estimated_transform_ransac=T_ransac)
        if 'transform' in item:
            rc, sc, cs = self.engine.corr()
            rec.update(self.summary.update((item['seq_id'], item['src_frame'], item['ref_frame']),
                                           np.asarray(item['transform'], np.float64), T, rc.cpu().numpy(),
                                           sc.cpu().numpy(), cs.cpu().numpy()))
        self.records.append(rec)
        return rec

    def run(self, stager, log=None):
        for item, ref_dev, src_dev in stager:
            rec = self.step(item, ref_dev, src_dev)
            if log:
                log('seq_id: {}, id0: {}, id1: {}, nCorr: {}'.format(rec['seq_id'], rec['ref_frame'], rec['src_frame'],
                                                                      rec['n_corr']))
        return self.records


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('--infer-root', default=None, help="directory with %%06d.npy scans of the 'infer' subset (assets/pc)")
    ap.add_argument('--dataset-root', default=None, help='KITTI-style root (icp10/<seq> lists + downsampled_xyzi/)')
    ap.add_argument('--subset', default='test')
    ap.add_argument('--dataset', default='kitti', help="'mulran' disables the vote layer (infer.py:119-120)")
    ap.add_argument('--weights', default=None, help='reference checkpoint (.pth.tar); default: synthetic seed-0 weights')
    ap.add_argument('--out', default=None)
    ap.add_argument('--no-npz', action='store_true')
    ap.add_argument('--neighbor-limits', type=int, nargs=5, default=None, help='skip the calibration')
    ap.add_argument('--bf16-attention', action='store_true')
    args = ap.parse_args(argv)

    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', 0)))
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group('nccl')
    cfg = config.make_cfg()
    if args.dataset == 'mulran':
        cfg.Vote.inference_use_vote = False
    cfg.thdroformer.attention_bf16 = bool(args.bf16_attention)
    b = cfg.backbone
    if args.infer_root:
        data = ds_mod.OdometryKittiPairDataset('.', 'infer', infer_root=args.infer_root)
        calib = data
    elif args.dataset_root:
        data = ds_mod.OdometryKittiPairDataset(args.dataset_root, args.subset)
        calib = data
    else:
        ap.error('give --infer-root or --dataset-root')
    t0 = time.time()
    if args.neighbor_limits:
        cfg.neighbor_limits = list(args.neighbor_limits)
    else:
        cfg.neighbor_limits = [int(x) for x in ds_mod.calibrate_neighbors_stack_mode(
            calib, None, b.num_stages, b.init_voxel_size, b.init_radius)]
    if rank == 0:
        print(f'Data loader created: {time.time() - t0:.3f}s collapsed.')
        print(f'Calibrate neighbors: {cfg.neighbor_limits}.')
    tester = Tester(cfg, load_state(args.weights, cfg), args.out, save_npz=not args.no_npz, write_poses=world == 1)
    mine = sharding.pairs_for_rank(len(data), rank, world)
    stager = ds_mod.PairStager(data, mine)
    records = tester.run(stager, log=print if rank == 0 else None)
    # one gather of fixed-size records: ids, counts, time, errors, the pose (12 floats) and the pair's dataset index
    rec = torch.tensor([[r['seq_id'], r['ref_frame'], r['src_frame'], r['n_corr'], r['ms'], r.get('r_RRE', float('nan')),
                         r.get('r_RTE', float('nan')), *np.asarray(r['transform'], np.float64).reshape(-1)[:12], idx]
                        for r, idx in zip(records, mine)], dtype=torch.float64, device='cuda').reshape(-1, 20)
    allrec = torch.cat(sharding.gather_records(rec, world, dist)).cpu().numpy()
    if rank == 0:
        allrec = allrec[np.argsort(allrec[:, 19], kind='stable')]  # dataset order, whatever the rank count
        if world > 1 and args.out:  # the single-rank file, not a rank-interleaved one (evaluation.pose_line format)
            for row in allrec:
                evaluation.append_pose(args.out, {'seq_id': int(row[0]), 'ref_frame': int(row[1]), 'src_frame': int(row[2])},
                                       row[7:19].astype(np.float32))
        print(f'pairs: {allrec.shape[0]}, mean ms/pair: {allrec[:, 4].mean() if len(allrec) else 0:.2f}')
        if len(allrec) and np.isfinite(allrec[:, 5]).any():
            ok = (allrec[:, 5] < tester.summary.rre_threshold) & (allrec[:, 6] < tester.summary.rte_threshold)
            print('  Registration (all ranks), RR: {:.4f}, RRE: {:.3f}, RTE: {:.3f}'.format(
                ok.mean(), allrec[ok, 5].mean() if ok.any() else 0.0, allrec[ok, 6].mean() if ok.any() else 0.0))
            if world == 1:  # correspondence-level meters (PIR / IR / FMR) are accumulated per rank only
                for line in tester.summary.lines()[1:]:
                    print(line)
    if dist is not None:
        dist.destroy_process_group()
    return 0


if __name__ == '__main__':
    sys.exit(main())
