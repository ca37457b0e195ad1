"""GPU: raw-scan voxel down-sampling and correspondence RANSAC through the C-ABI against
oracle/preprocess.py.  Open3D (the reference's implementation of both) is not in the reference tree: parity is
against the oracle's restatement only ("unpinned")."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def raw_scan(seed, n):
    rng = np.random.default_rng(seed)
    r = rng.uniform(2, 80, n) ** 0.8
    az = rng.uniform(-np.pi, np.pi, n)
    xyz = np.stack([r * np.cos(az), r * np.sin(az), rng.normal(-1.2, 0.6, n)], 1)
    return np.concatenate([xyz, rng.uniform(0, 1, (n, 1))], 1).astype(np.float32)


@pytest.mark.parametrize('n,voxel', [(0, 0.3), (1, 0.3), (5000, 0.3), (123457, 0.3), (40000, 1.0), (4097, 0.05)])
def test_voxel_downsample_is_bit_identical_to_the_oracle(n, voxel):
    from oracle import preprocess
    from rdmnet_amd import ops
    pts = raw_scan(n, n)
    want = preprocess.voxel_down_sample(pts, voxel)
    got = ops.voxel_downsample(torch.from_numpy(pts).cuda(), voxel).cpu().numpy()
    assert got.shape == want.shape
    assert np.array_equal(got, want)  # same voxels, same order, same float64 sums
    again = ops.voxel_downsample(torch.from_numpy(pts).cuda(), voxel).cpu().numpy()
    assert np.array_equal(got, again)  # atomics only place list entries; the sums are order-fixed


def test_voxel_downsample_xyz_only_strided_and_bad_input():
    from oracle import preprocess
    from rdmnet_amd import ops
    pts = raw_scan(3, 20000)
    got = ops.voxel_downsample(torch.from_numpy(pts).cuda()[:, :3], 0.3).cpu().numpy()  # row stride 4, 3 channels
    assert np.array_equal(got, preprocess.voxel_down_sample(pts[:, :3], 0.3))
    bad = pts.copy()
    bad[17, 1] = np.nan
    with pytest.raises(RuntimeError):
        ops.voxel_downsample(torch.from_numpy(bad).cuda(), 0.3)


def test_raw_scan_to_pose_pipeline_runs():
    """120 k-point raw scans -> GPU voxel grid -> engine: the path extended to raw KITTI-sized input."""
    from rdmnet_amd import config, engine, ops, weights
    cfg = config.make_cfg()
    eng = engine.Engine(cfg, weights.synthetic_state_dict(cfg, seed=0))
    def dense_scan(seed):  # ~6 returns per 0.3 m voxel, like a raw 64-beam sweep: 120 k points -> ~20 k voxels
        base = raw_scan(seed, 20000)
        rng = np.random.default_rng(seed + 100)
        pts = np.repeat(base, 6, axis=0)
        pts[:, :3] += rng.normal(0, 0.04, (pts.shape[0], 3)).astype(np.float32)
        return pts[rng.permutation(pts.shape[0])]
    a = ops.voxel_downsample(torch.from_numpy(dense_scan(10)).cuda(), 0.3)[:, :3].contiguous()
    b = ops.voxel_downsample(torch.from_numpy(dense_scan(11)).cuda(), 0.3)[:, :3].contiguous()
    assert 15000 < a.shape[0] < 60000
    res = eng.run(a, b)
    assert res.level_sizes[0] == a.shape[0] + b.shape[0] and np.isfinite(eng.transform()).all()


def planted(seed, n, outliers):
    rng = np.random.default_rng(seed)
    src = rng.uniform(-20, 20, (n, 3)).astype(np.float32)
    ang = rng.uniform(-1, 1)
    R = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]])
    t = rng.uniform(-5, 5, 3)
    ref = (src @ R.T + t + rng.normal(0, 0.03, src.shape)).astype(np.float32)
    k = int(outliers * n)
    ref[:k] = rng.uniform(-20, 20, (k, 3)).astype(np.float32)
    return src, ref, R, t


@pytest.mark.parametrize('n,iters,ransac_n', [(438, 50000, 4), (60, 2000, 3), (2000, 5000, 4)])
def test_ransac_matches_the_oracle(n, iters, ransac_n):
    from oracle import preprocess
    from rdmnet_amd import ops
    src, ref, R, t = planted(n, n, 0.6)
    T_o, best_o, inl_o, rmse_o, counts_o = preprocess.ransac_correspondences(src, ref, 0.3, ransac_n, iters, seed=11)
    T, stats, rmse, hyp = ops.ransac_correspondences(torch.from_numpy(src).cuda(), torch.from_numpy(ref).cuda(), 0.3, ransac_n,
                                                     iters, seed=11, return_hypotheses=True)
    hyp = hyp.cpu().numpy()
    # per-iteration inlier counts: same draws, float64 fit; Horn vs SVD differ by round-off, so a correspondence
    # sitting within ~1e-12 m of the threshold may flip -- none is expected, a handful is tolerated
    assert (hyp != counts_o).sum() <= max(2, iters // 5000), int((hyp != counts_o).sum())
    best, inl = (int(x) for x in stats.cpu())
    assert inl == hyp.max() and hyp[best] == inl
    assert inl == inl_o and best == best_o
    assert abs(float(rmse) - rmse_o) <= 1e-6
    assert np.abs(T.cpu().numpy().astype(np.float64) - T_o).max() <= 2e-6 * max(1.0, np.abs(T_o).max())
    assert np.abs(T.cpu().numpy()[:3, :3] - R).max() < 0.03  # and it found the planted motion
    T2, stats2, _ = ops.ransac_correspondences(torch.from_numpy(src).cuda(), torch.from_numpy(ref).cuda(), 0.3, ransac_n, iters,
                                               seed=11)
    assert torch.equal(T, T2) and torch.equal(stats, stats2)  # deterministic


def test_ransac_degenerate_inputs():
    from rdmnet_amd import ops
    src = torch.zeros((2, 3), device='cuda')
    T, stats, rmse = ops.ransac_correspondences(src, src.clone(), 0.3, 4, 100)
    assert torch.equal(T.cpu(), torch.eye(4)) and stats.cpu().tolist() == [-1, 0]
    empty = torch.zeros((0, 3), device='cuda')
    T, stats, rmse = ops.ransac_correspondences(empty, empty.clone(), 0.3, 4, 100)
    assert torch.equal(T.cpu(), torch.eye(4)) and stats.cpu().tolist() == [-1, 0]


def test_engine_grows_its_arena_for_dense_input():
    """2 x ~95 k points do not fit the default 3 GiB activation arena: the engine doubles it and re-runs; an arena
    size fixed by the caller is an error instead."""
    from rdmnet_amd import config, engine, ops, weights
    cfg = config.make_cfg()
    state = weights.synthetic_state_dict(cfg, seed=0)
    a = ops.voxel_downsample(torch.from_numpy(raw_scan(20, 120000)).cuda(), 0.3)[:, :3].contiguous()
    b = ops.voxel_downsample(torch.from_numpy(raw_scan(21, 120000)).cuda(), 0.3)[:, :3].contiguous()
    assert a.shape[0] > 80000
    eng = engine.Engine(cfg, state)
    res = eng.run(a, b)
    assert res.arena_used > (3 << 30) and np.isfinite(eng.transform()).all()
    T1 = eng.transform()
    eng.run(a, b)
    assert np.array_equal(T1, eng.transform())
    fixed = engine.Engine(cfg, state, arena_bytes=1 << 30)
    with pytest.raises(RuntimeError, match='arena exhausted'):
        fixed.run(a, b)
