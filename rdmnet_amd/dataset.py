"""Pair loader, neighbour-limit calibration and host->HBM staging.

Mirrors, for the inference path only (no training augmentation, no GT correspondences):
  * `load_kitti_gt_txt`, `make_dataset_kitti`, `OdometryKittiPairDataset`
    (rdmnet/datasets/registration/kitti/dataset.py:16-75, 72-191): metadata lists and the `.npy`
    [N, >=3] scan reader; same item keys.  The reference's non-'infer' branch reads `transform` before
    assigning it (dataset.py:160, an UnboundLocalError); here the metadata transform is used, which is
    what the following line of the reference intends.
  * `calibrate_neighbors_stack_mode` (geotransformer/utils/data.py:195-220): same arguments and result,
    computed on the GPU: the pyramid comes from the HIP grid subsampling, the neighbourhood sizes from
    count-only radius searches, the histogram from rdm_neighbor_histogram.
  * `infer_data_loader` / `test_data_loader` (experiments/dataset.py:86-146): calibrated limits + an
    iterable over staged pairs.  The reference's DataLoader workers run the collate on the CPU; here the
    collate is part of the GPU path, so the loader only has to keep raw scans flowing: `PairStager`
    reads scans on background threads into pinned host buffers and issues the H2D copies on a side
    stream, double-buffered, so the copy of pair i+1 overlaps the kernels of pair i.
"""
import os
import os.path as osp
import queue
import threading

import numpy as np
import torch

from . import _lib, ops


def load_kitti_gt_txt(txt_root, seq):
    """One line per pair: `anc_idx pos_idx` + 12 floats (3x4 row-major).  frame0 = pos, frame1 = anc
    (dataset.py:16-38)."""
    dataset = []
    with open(osp.join(txt_root, '%02d' % seq), 'r') as f:
        for line in f.readlines():
            parts = line.split()
            if not parts:
                continue
            trans = np.array([float(x) for x in parts[2:]]).reshape(3, 4)
            trans = np.vstack([trans, [0, 0, 0, 1]])
            dataset.append({'seq_id': seq, 'frame0': int(parts[1]), 'frame1': int(parts[0]), 'transform': trans})
    return dataset


_SPLITS = {'train': [0, 1, 2, 3, 4, 5], 'val': [6, 7], 'test': [8, 9, 10]}


def make_dataset_kitti(txt_path, mode):
    """dataset.py:41-70.  'infer' is the two bundled pairs (0,4), (0,7) of sequence 0."""
    if mode == 'infer':
        return [{'seq_id': 0, 'frame0': 0, 'frame1': 4}, {'seq_id': 0, 'frame0': 0, 'frame1': 7}]
    if mode not in _SPLITS:
        raise Exception('Invalid mode.')
    dataset = []
    for seq in _SPLITS[mode]:
        dataset += load_kitti_gt_txt(txt_path, seq)
    return dataset


class OdometryKittiPairDataset:
    """Items: seq_id, ref_frame, src_frame, [transform], ref_points, src_points (f32 [N,3]),
    ref_feats, src_feats (f32 ones [N,1]).  `infer_root` is the reference's hard-coded './assets/pc'."""

    def __init__(self, dataset_root, subset, point_limit=None, benchmark_distance=10, infer_root='./assets/pc',
                 metadata=None):
        self.dataset_root = dataset_root
        self.subset = subset
        self.point_limit = point_limit
        self.infer_root = infer_root
        self.metadata = metadata if metadata is not None else make_dataset_kitti(
            osp.join(dataset_root, 'icp%i' % benchmark_distance), subset)

    def _load_point_cloud(self, file_name):
        points = np.load(file_name)
        if self.point_limit is not None and points.shape[0] > self.point_limit:
            indices = np.random.permutation(points.shape[0])[: self.point_limit]
            points = points[indices]
        return points

    def scan_path(self, seq_id, frame):
        if self.subset == 'infer':
            return osp.join(self.infer_root, '%06d.npy' % frame)
        return osp.join(self.dataset_root, 'downsampled_xyzi', '%02d' % seq_id, '%06d.npy' % frame)

    def __getitem__(self, index):
        meta = self.metadata[index]
        d = {'seq_id': meta['seq_id'], 'ref_frame': meta['frame0'], 'src_frame': meta['frame1']}
        if self.subset != 'infer':
            d['transform'] = np.asarray(meta['transform']).astype(np.float32)
        ref = self._load_point_cloud(self.scan_path(d['seq_id'], d['ref_frame']))[:, :3]
        src = self._load_point_cloud(self.scan_path(d['seq_id'], d['src_frame']))[:, :3]
        d['ref_points'] = ref.astype(np.float32)
        d['src_points'] = src.astype(np.float32)
        d['ref_feats'] = np.ones((ref.shape[0], 1), dtype=np.float32)
        d['src_feats'] = np.ones((src.shape[0], 1), dtype=np.float32)
        return d

    def __len__(self):
        return len(self.metadata)


class ArrayPairDataset:
    """Same items from in-memory scans (synthetic pairs, fixtures): pairs = [(ref [N,3], src [M,3]) or
    (ref, src, transform)]."""

    def __init__(self, pairs, seq_id=0):
        self.pairs, self.seq_id = pairs, seq_id

    def __len__(self):
        return len(self.pairs)

    def __getitem__(self, i):
        p = self.pairs[i]
        ref, src = np.asarray(p[0], np.float32), np.asarray(p[1], np.float32)
        d = {'seq_id': self.seq_id, 'ref_frame': 2 * i, 'src_frame': 2 * i + 1, 'ref_points': ref, 'src_points': src,
             'ref_feats': np.ones((ref.shape[0], 1), np.float32), 'src_feats': np.ones((src.shape[0], 1), np.float32)}
        if len(p) > 2:
            d['transform'] = np.asarray(p[2], np.float32)
        return d


class CyclingPairDataset:
    """`length` items that cycle through the pairs of `base` (an ArrayPairDataset of a few distinct synthetic pairs):
    a long pair stream for throughput runs without a dataset on disk.  Frame ids follow the position in the stream."""

    def __init__(self, base, length):
        self.base, self.length = base, int(length)

    def __len__(self):
        return self.length

    def __getitem__(self, i):
        d = dict(self.base[i % len(self.base)])
        d['ref_frame'], d['src_frame'] = 2 * i, 2 * i + 1
        return d


def neighbor_histograms(item, num_stages, voxel_size, search_radius, hist_n, hists=None, device=None):
    """Adds the neighbourhood-size histograms of one pair to `hists` (int32 [num_stages, hist_n] on the
    device).  Level i: counts of the self search with radius r*2^i over the stacked [ref; src] level."""
    L = _lib.lib()
    device = device or torch.device('cuda', torch.cuda.current_device())
    if hists is None:
        hists = torch.zeros((num_stages, hist_n), dtype=torch.int32, device=device)
    pts = torch.cat([torch.as_tensor(item['ref_points']), torch.as_tensor(item['src_points'])]).float().to(device)
    lengths = torch.tensor([item['ref_points'].shape[0], item['src_points'].shape[0]], dtype=torch.int64, device=device)
    radius, voxel = search_radius, voxel_size
    for i in range(num_stages):
        if i > 0:
            voxel *= 2
            cap, lengths = ops.grid_subsample_device(pts, lengths, voxel)
            pts = cap[:int(lengths.sum())]
        n = pts.shape[0]
        counts = torch.empty((max(n, 1),), dtype=torch.int32, device=device)
        flags = torch.zeros((2,), dtype=torch.int32, device=device)
        ws = ops.scratch(device, L.rdm_radius_neighbors_workspace_bytes(n, n, 2))
        _lib.check(L.rdm_radius_neighbors(pts.data_ptr(), n, pts.data_ptr(), n, lengths.data_ptr(), lengths.data_ptr(), 2,
                                          float(radius), 0, 0, counts.data_ptr(), flags.data_ptr(), flags[1:].data_ptr(),
                                          ws.data_ptr(), ws.numel(), _lib.stream_ptr()), 'rdm_radius_neighbors')
        _lib.check(L.rdm_neighbor_histogram(counts.data_ptr(), n, hists[i].data_ptr(), hist_n, _lib.stream_ptr()),
                   'rdm_neighbor_histogram')
        radius *= 2
    return hists


def calibrate_neighbors_stack_mode(dataset, collate_fn=None, num_stages=5, voxel_size=0.3, search_radius=1.275,
                                   keep_ratio=0.8, sample_threshold=2000, return_hists=False):
    """geotransformer/utils/data.py:195-220.  `collate_fn` is accepted for signature compatibility and
    unused: only the neighbourhood sizes are needed, so no index table is materialised."""
    hist_n = int(np.ceil(4 / 3 * np.pi * (search_radius / voxel_size + 1) ** 3))
    hists = None
    for i in range(len(dataset)):
        hists = neighbor_histograms(dataset[i], num_stages, voxel_size, search_radius, hist_n, hists)
        if int(hists.sum(1).min()) > sample_threshold:  # one small D2H per pair, as the reference checks per pair
            break
    neighbor_hists = hists.cpu().numpy() if hists is not None else np.zeros((num_stages, hist_n), np.int32)
    cum_sum = np.cumsum(neighbor_hists.T, axis=0)
    neighbor_limits = np.sum(cum_sum < (keep_ratio * cum_sum[hist_n - 1, :]), axis=0)
    return (neighbor_limits, neighbor_hists) if return_hists else neighbor_limits


class PairStager:
    """Iterates a dataset as (item, ref_dev, src_dev): scans are read by `workers` background threads,
    packed into pinned host buffers and copied to HBM on a side stream; `depth` pairs are staged ahead.
    The consumer's current stream waits on the copy's event, so no host synchronisation is needed; with several
    consumer threads (rdmnet_amd.pipeline: one per in-flight pair, drawing under a lock) each draw makes the DRAWING
    thread's stream wait.  The pinned buffers are a fixed ring of `depth + workers` allocations that are re-used (an
    allocation or release of pinned memory synchronises the device: per pair it capped the harness at ~60 pairs/s).
    `indices` selects this rank's pairs (see sharding.pairs_for_rank)."""

    def __init__(self, dataset, indices=None, device=None, depth=2, workers=2):
        self.dataset = dataset
        self.indices = list(range(len(dataset))) if indices is None else list(indices)
        self.device = device or torch.device('cuda', torch.cuda.current_device())
        self.depth, self.workers = max(1, depth), max(1, workers)

    def __len__(self):
        return len(self.indices)

    def __iter__(self):
        todo = queue.Queue()
        for slot, i in enumerate(self.indices):
            todo.put((slot, i))
        ready = {}
        cv = threading.Condition()
        pool = queue.Queue()  # [pinned tensor or None, event of the last copy out of it or None]
        for _ in range(self.depth + self.workers):
            pool.put([None, None])
        errors = []
        stop = threading.Event()

        def work():
            torch.set_num_threads(1)  # (this thread's OpenMP teams, should a dataset's __getitem__ use torch on the CPU)
            while not stop.is_set():
                # the buffer first, then the slot: whoever holds a buffer takes the LOWEST unclaimed slot, so the slot the
                # consumer waits for can never starve behind later slots that took every buffer
                try:
                    buf = pool.get(timeout=0.1)  # (bounds the pinned memory in flight)
                except queue.Empty:
                    continue
                try:
                    slot, i = todo.get_nowait()
                except queue.Empty:
                    pool.put(buf)
                    return
                try:
                    item = self.dataset[i]
                    ref, src = item['ref_points'], item['src_points']
                    rows = ref.shape[0] + src.shape[0]
                    if buf[1] is not None:
                        buf[1].synchronize()  # the previous copy out of this buffer has finished
                        buf[1] = None
                    if buf[0] is None or buf[0].shape[0] < rows:  # first use, or a larger pair than any before
                        buf[0] = torch.empty((max(rows, 1) * 5 // 4, 3), dtype=torch.float32).pin_memory()
                    host = buf[0][:rows]
                    # (through the numpy view: a plain memcpy.  torch's CPU copy_ opens an OpenMP team per calling thread -- 128
                    # threads each on a large host -- and under a container's CPU quota their spinning throttles the whole
                    # process: 13 instead of 220 pairs/s with two reader threads)
                    hv = host.numpy()
                    hv[:ref.shape[0]] = ref
                    hv[ref.shape[0]:] = src
                    out = (item, host, buf)
                except Exception as e:  # surfaced in the consumer
                    errors.append(e)
                    pool.put(buf)
                    out = None
                with cv:
                    ready[slot] = out
                    cv.notify_all()

        threads = [threading.Thread(target=work, daemon=True) for _ in range(self.workers)]
        for t in threads:
            t.start()
        copy_stream = torch.cuda.Stream(device=self.device)

        def stage(slot):
            with cv:
                while slot not in ready:
                    cv.wait(timeout=0.05)
                    if errors:
                        raise errors[0]
                    if stop.is_set():
                        return None
                got = ready.pop(slot)
            if got is None:
                raise errors[0]
            item, host, buf = got
            with torch.cuda.stream(copy_stream):
                dev = host.to(self.device, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(copy_stream)
            buf[1] = ev
            pool.put(buf)  # (its next user waits for `ev` before it overwrites the buffer)
            return item, dev, ev

        # The copies are issued by a thread of the stager's own, `depth` pairs ahead: a consumer that draws a pair only waits for
        # the hand-off event (issuing the copy in the drawing thread cost each pipeline worker 0.3 ms per pair with its stream idle)
        staged = queue.Queue(maxsize=self.depth)
        n = len(self.indices)

        def stage_all():
            torch.set_num_threads(1)
            try:
                with torch.cuda.device(self.device):
                    for slot in range(n):
                        out = stage(slot)
                        while out is not None and not stop.is_set():
                            try:
                                staged.put(out, timeout=0.05)
                                break
                            except queue.Full:
                                continue
                        if out is None or stop.is_set():
                            return
            except BaseException as e:  # surfaced in the consumer
                while not stop.is_set():
                    try:
                        staged.put(e, timeout=0.05)
                        return
                    except queue.Full:
                        continue

        stager = threading.Thread(target=stage_all, daemon=True)
        stager.start()
        try:
            for _ in range(n):
                got = staged.get()
                if isinstance(got, BaseException):
                    raise got
                item, dev, ev = got
                torch.cuda.current_stream(self.device).wait_event(ev)
                dev.record_stream(torch.cuda.current_stream(self.device))
                n_ref = item['ref_points'].shape[0]
                yield item, dev[:n_ref], dev[n_ref:]
        finally:
            stop.set()


def infer_data_loader(cfg, dataset='kitti', infer_root='./assets/pc', rank=0, world=1):
    """experiments/dataset.py:120-146: -> (stager over the 'infer' pairs, calibrated neighbour limits)."""
    ds = OdometryKittiPairDataset(getattr(cfg, 'dataset_root', '.'), 'infer', infer_root=infer_root)
    b = cfg.backbone
    limits = calibrate_neighbors_stack_mode(ds, None, b.num_stages, b.init_voxel_size, b.init_radius)
    from .sharding import pairs_for_rank
    return PairStager(ds, pairs_for_rank(len(ds), rank, world)), limits


def test_data_loader(cfg, dataset_root, rank=0, world=1, point_limit=None, calibration_subset='train'):
    """experiments/dataset.py:55-88: limits calibrated on the TRAIN subset (the reference does so with
    its random training augmentation switched on, which is not reproducible; here the scans are used as
    stored), pairs from the test subset."""
    b = cfg.backbone
    calib = OdometryKittiPairDataset(dataset_root, calibration_subset, point_limit=point_limit)
    limits = calibrate_neighbors_stack_mode(calib, None, b.num_stages, b.init_voxel_size, b.init_radius)
    ds = OdometryKittiPairDataset(dataset_root, 'test', point_limit=point_limit)
    from .sharding import pairs_for_rank
    return PairStager(ds, pairs_for_rank(len(ds), rank, world)), limits


test_data_loader.__test__ = False  # not a pytest test
