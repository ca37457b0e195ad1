"""Inference harness: the counterpart of experiments/infer.py (`Tester`, :19-110) + experiments/eval.py's
registration report, on the native engine.

    python -m rdmnet_amd.infer --infer-root /path/to/assets/pc --out out/           # the two bundled pairs
    python -m rdmnet_amd.infer --dataset-root /data/kitti --subset test --out out/ --weights rdmnet.pth.tar
    python -m rdmnet_amd.infer --synthetic 512 --no-npz                              # throughput on synthetic KITTI-shaped pairs
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
from .pipeline import DEFAULT_PAIRS_IN_FLIGHT, PairPipeline, pin_rank


def load_state(path, cfg, seed=0):
    """`state['model']` of a reference checkpoint (base_tester.py:97-107), or synthetic weights."""
    if path:
        state = torch.load(path, map_location='cpu')
        return state['model'] if 'model' in state else state
    return weights.synthetic_state_dict(cfg, seed=seed)


class Tester:
    """`run(stager)` processes this rank's pairs with `pairs_in_flight` of them on the GPU at a time
    (rdmnet_amd.pipeline.PairPipeline: what replaces the one-pair-at-a-time loop of engine/single_tester.py:86-134);
    records, pose lines and the report come out in dataset order whatever the completion order."""

    def __init__(self, cfg, state, output_dir=None, save_npz=True, ransac=True, write_poses=True,
                 pairs_in_flight=DEFAULT_PAIRS_IN_FLIGHT, wait_us=None, lockstep=None):
        self.cfg, self.output_dir, self.save_npz, self.ransac = cfg, output_dir, save_npz, ransac
        self.write_poses = write_poses  # False under several ranks: rank 0 writes all poses, in pair order, at the end
        self.pipeline = PairPipeline(cfg, state, pairs_in_flight=pairs_in_flight, wait_us=wait_us,
                                     keep_taps=bool(save_npz and output_dir), lockstep=lockstep)
        self.engine = self.pipeline.engines[0]  # (the serial entry point `step` runs on this one)
        if output_dir:
            os.makedirs(output_dir, exist_ok=True)
        self.summary = evaluation.Summary()
        self.records = []

    @staticmethod
    def output_dict(e, n_ref):
        """The tensors infer.py:84-101 stores, from engine e's taps of its last run."""
        r = e.result
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

    def _work(self, eng, job):
        """One pair on a worker thread / stream of the pipeline: run it and take everything the harness needs out of the
        engine (the next pair reuses its arena).  The .npz of a pair is an independent file: written here."""
        item, ref_dev, src_dev = job
        t0 = time.perf_counter()
        res = eng.run(ref_dev.contiguous(), src_dev.contiguous())  # returns with pose AND correspondences on the host
        ms = (time.perf_counter() - t0) * 1e3
        T = eng.transform()
        rec = {'seq_id': item['seq_id'], 'ref_frame': item['ref_frame'], 'src_frame': item['src_frame'],
               'n_corr': int(res.n_correspondences), 'ms': ms, 'transform': T}
        if 'transform' in item:  # what the pair's registration / correspondence numbers need (the engine's buffers are reused)
            rec['_measure'] = (np.asarray(item['transform'], np.float64), T) + eng.host_corr()  # (numpy copies)
        if self.output_dir and self.save_npz:
            od = self.output_dict(eng, item['ref_points'].shape[0])
            T_ransac = None
            if self.ransac:  # infer.py:75-82: distance 0.3, ransac_n 4, 50 000 iterations, on the GPU
                T_ransac = ops.ransac_correspondences(od['src_corr_points'].contiguous(), od['ref_corr_points'].contiguous(),
                                                      0.3, 4, 50000)[0].cpu().numpy().astype(np.float64)
            evaluation.save_pair_npz(self.output_dir, item, od, estimated_transform_ransac=T_ransac)
        return rec

    def _commit(self, rec):
        """In dataset order, on the calling thread: pose line, registration meters, record list."""
        if self.output_dir and self.write_poses:
            evaluation.append_pose(self.output_dir, rec, rec['transform'])
        args = rec.pop('_measure', None)
        if args is not None:  # 0.3 ms of host work per pair: here it does not keep a worker's stream idle
            rec.update(self.summary.commit((rec['seq_id'], rec['src_frame'], rec['ref_frame']), self.summary.measure(*args)))
        self.records.append(rec)
        return rec

    def step(self, item, ref_dev, src_dev):
        """One pair, serially, on the calling thread's current stream."""
        return self._commit(self._work(self.engine, (item, ref_dev, src_dev)))

    def run(self, stager, log=None):
        """Every pair of the stager, `pairs_in_flight` at a time, committed in dataset order.  If a pair fails, the pairs before
        it are still committed (pose lines, records) before the error is raised; .npz files of later pairs that had already
        finished on other workers may exist without a pose line."""
        # (tensors_of: the pipeline's workers collate several staged pairs with one sequence of launches, pipeline.PairPipeline.imap)
        for i, rec in enumerate(self.pipeline.imap(stager, self._work, tensors_of=lambda job: (job[1].contiguous(), job[2].contiguous()))):
            # a pair's time: from the moment its worker drew it to its result (in a lock-step group the pairs finish together, and
            # `engine.run` in _work only picks the result up)
            rec['ms'] = self.pipeline.last_stats['latency_ms'].get(i, rec['ms'])
            self._commit(rec)
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
    ap.add_argument('--synthetic', type=int, default=0, metavar='N',
                    help='no dataset: N pairs cycling through the bench workload\'s seeded synthetic KITTI-shaped pairs '
                         '(rdmnet_amd.synthetic; --synthetic-distinct of them, cached under --synthetic-cache)')
    ap.add_argument('--synthetic-distinct', type=int, default=8)
    ap.add_argument('--synthetic-cache', default=os.path.join('gpurun_out', 'bench_pairs'))
    ap.add_argument('--pairs-in-flight', type=int, default=DEFAULT_PAIRS_IN_FLIGHT,
                    help='pairs on the GPU at a time (engines / host threads / HIP streams; rdmnet_amd.pipeline)')
    ap.add_argument('--lockstep', type=int, default=None,
                    help='pairs a stream runs as one lock-step group (identical kernels of the group as one grouped launch; default: '
                         'rdmnet_amd.pipeline.DEFAULT_LOCKSTEP with two or more pairs in flight and no .npz outputs; 1 = one pair per engine call)')
    ap.add_argument('--no-ransac', action='store_true', help='skip the RANSAC estimate stored beside the LGR pose in the .npz')
    ap.add_argument('--quiet', action='store_true', help='no per-pair log line (the reference prints one per iteration, infer.py:62-66)')
    args = ap.parse_args(argv)

    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    local_rank, local_world = int(os.environ.get('LOCAL_RANK', 0)), int(os.environ.get('LOCAL_WORLD_SIZE', world))
    pin_rank(local_rank, local_world)  # the CPUs of this rank's GPU's NUMA node (before the first HIP call)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group('nccl')
    cfg = config.make_cfg()
    if args.dataset == 'mulran':
        cfg.Vote.inference_use_vote = False
    cfg.thdroformer.attention_bf16 = bool(args.bf16_attention)
    b = cfg.backbone
    if args.synthetic > 0:
        from . import synthetic
        fixture = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'synthetic_pairs.npz')
        base = ds_mod.ArrayPairDataset(synthetic.cached_pairs(args.synthetic_distinct, args.synthetic_cache, fixture))
        data = ds_mod.CyclingPairDataset(base, args.synthetic)
        calib = base
    elif args.infer_root:
        data = ds_mod.OdometryKittiPairDataset('.', 'infer', infer_root=args.infer_root)
        calib = data
    elif args.dataset_root:
        data = ds_mod.OdometryKittiPairDataset(args.dataset_root, args.subset)
        calib = data
    else:
        ap.error('give --infer-root, --dataset-root or --synthetic N')
    t0 = time.time()
    if args.neighbor_limits:
        cfg.neighbor_limits = list(args.neighbor_limits)
    else:
        cfg.neighbor_limits = [int(x) for x in ds_mod.calibrate_neighbors_stack_mode(
            calib, None, b.num_stages, b.init_voxel_size, b.init_radius)]
    if rank == 0:
        print(f'Data loader created: {time.time() - t0:.3f}s collapsed.')
        print(f'Calibrate neighbors: {cfg.neighbor_limits}.')
    tester = Tester(cfg, load_state(args.weights, cfg), args.out, save_npz=not args.no_npz, ransac=not args.no_ransac,
                    write_poses=world == 1, pairs_in_flight=args.pairs_in_flight, lockstep=args.lockstep)
    mine = sharding.pairs_for_rank(len(data), rank, world)
    # scans are read and staged (pinned host -> HBM on a side stream) two pairs ahead of every in-flight pair
    stager = ds_mod.PairStager(data, mine, depth=2 * args.pairs_in_flight, workers=max(2, args.pairs_in_flight))
    t_run = time.perf_counter()
    records = tester.run(stager, log=print if rank == 0 and not args.quiet else None)
    torch.cuda.synchronize()
    t_run = time.perf_counter() - t_run
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
        print(f'pairs: {allrec.shape[0]}, mean ms/pair: {allrec[:, 4].mean() if len(allrec) else 0:.2f}, '
              f'{allrec.shape[0] / max(t_run, 1e-9):.1f} pairs/s (rank 0 wall time {t_run:.2f} s: host scans -> staging -> '
              f'{tester.pipeline.n} streams x {tester.pipeline.lockstep} pair(s) per lock-step group in flight -> poses and correspondences on the host)')
        st = tester.pipeline.last_stats
        if st and st['jobs']:
            print('  worker time per pair (ms): drawing + staging the next pair {:.2f}, engine + outputs {:.2f}, waiting for the '
                  'in-order consumer {:.2f}'.format(*(1e3 * st[k] / st['jobs'] for k in ('draw_s', 'work_s', 'window_s'))))
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
