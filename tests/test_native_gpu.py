"""GPU parity: rdmnet_amd.ext (HIP, through the C-ABI) vs the oracle, bit-exact."""
import os

import numpy as np
import pytest
import torch

from pyramid import build_levels, search_calls, sha

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ext():
    from rdmnet_amd import ext
    return ext


def gpu_grid(ext):
    def f(p, l, v):
        sp, sl = ext.grid_subsampling(torch.from_numpy(p).cuda(), torch.from_numpy(l).cuda(), float(v))
        return sp.cpu().numpy(), sl.cpu().numpy()
    return f


def gpu_radius(ext, q, s, ql, sl, r, **kw):
    return ext.radius_neighbors(torch.from_numpy(q).cuda(), torch.from_numpy(s).cuda(), torch.from_numpy(ql).cuda(),
                                torch.from_numpy(sl).cuda(), float(r), **kw).cpu().numpy()


@pytest.mark.parametrize('pair', [('000000', '000004'), ('000000', '000007')])
def test_pyramid_matches_reference_goldens(ext, scans, golden_dir, pair):
    golden = np.load(os.path.join(golden_dir, 'native_golden.npz'))
    a, b = scans['s' + pair[0]], scans['s' + pair[1]]
    tag = f'{pair[0]}_{pair[1]}'
    P, L = build_levels(gpu_grid(ext), np.concatenate([a, b]), np.array([len(a), len(b)], dtype=np.int64))
    for lvl in range(1, 5):
        assert np.array_equal(L[lvl], golden[f'{tag}/lengths{lvl}'])
        assert np.array_equal(P[lvl], golden[f'{tag}/points{lvl}']), f'level {lvl}: values or order differ'
    for name, q, s, ql, sl, r, lim in search_calls(P, L):
        idx = gpu_radius(ext, q, s, ql, sl, r)
        key = f'{tag}/{name}'
        assert idx.shape[1] == int(golden[key + '/width'])
        assert np.array_equal((idx < s.shape[0]).sum(1), golden[key + '/counts'])
        assert np.array_equal(sha(idx[:, :lim]), golden[key + '/sha'])
        lim_idx = gpu_radius(ext, q, s, ql, sl, r, width=lim)  # direct truncated path
        assert np.array_equal(lim_idx, idx[:, :lim])


def test_random_and_edge_clouds_match_oracle(ext, oracle_native):
    o = oracle_native.restatement()
    rng = np.random.default_rng(1)
    cases = [(1, 1, 1.0), (7, 300, 3.0), (2500, 1800, 15.0), (3000, 1, 30.0), (64, 64, 0.05)]
    for n0, n1, scale in cases:
        pts = (rng.standard_normal((n0 + n1, 3)) * scale).astype(np.float32)
        pts[: min(n0, 3)] = pts[0]  # exact duplicates -> ties, multi-point voxels
        lens = np.array([n0, n1], dtype=np.int64)
        for voxel in (0.6, 2.4):
            po, lo = o.grid_subsampling(pts, lens, np.float32(voxel))
            pg, lg = gpu_grid(ext)(pts, lens, np.float32(voxel))
            assert np.array_equal(lo, lg) and np.array_equal(po, pg), (n0, n1, scale, voxel)
            io = o.radius_neighbors(po, pts, lo, lens, np.float32(voxel * 2.125))
            ig = gpu_radius(ext, po, pts, lo, lens, voxel * 2.125)
            assert np.array_equal(io, ig), (n0, n1, scale, voxel)


@pytest.mark.parametrize('n0,n1,scale,voxel', [(65, 200, 0.02, 0.6), (3000, 9, 0.8, 0.6), (70000, 300, 30.0, 0.6),
                                               (66000, 66000, 0.5, 2.4)])
def test_crowded_voxels_and_long_clouds_match_oracle(ext, oracle_native, n0, n1, scale, voxel):
    """Voxels with more than 8 points are summed by a whole wavefront (up to 64 points; 65 and 200 points in one voxel
    take the serial path), and clouds of more than 65536 points recompute the first-occurrence ballots instead of
    keeping them in LDS; the sums are sequential fp32 adds in point order, so everything is bit-exact."""
    o = oracle_native.restatement()
    rng = np.random.default_rng(n0 + n1)
    pts = (rng.standard_normal((n0 + n1, 3)) * scale).astype(np.float32)
    lens = np.array([n0, n1], dtype=np.int64)
    po, lo = o.grid_subsampling(pts, lens, np.float32(voxel))
    pg, lg = gpu_grid(ext)(pts, lens, np.float32(voxel))
    assert np.array_equal(lo, lg) and np.array_equal(po, pg)


@pytest.mark.parametrize('n0,n1,scale', [(40000, 9000, 40.0), (9000, 30000, 25.0), (10400, 10200, 300.0)])
def test_large_clouds_cover_both_order_replay_paths(ext, oracle_native, n0, n1, scale):
    """The hash-map order replay runs from LDS up to 10304 voxels per cloud and from HBM beyond; each case has one
    cloud on either side (third case: every point its own voxel, right at the boundary)."""
    o = oracle_native.restatement()
    rng = np.random.default_rng(n0)
    pts = (rng.uniform(-1, 1, (n0 + n1, 3)) * np.array([scale, scale, 2.0])).astype(np.float32)
    lens = np.array([n0, n1], dtype=np.int64)
    po, lo = o.grid_subsampling(pts, lens, np.float32(0.6))
    pg, lg = gpu_grid(ext)(pts, lens, np.float32(0.6))
    assert min(lo) <= 10304 < max(lo) or scale == 300.0, lo
    assert np.array_equal(lo, lg) and np.array_equal(po, pg)


def test_empty_cloud_in_batch(ext, oracle_native):
    o = oracle_native.restatement()
    pts = np.random.default_rng(2).standard_normal((50, 3)).astype(np.float32)
    lens = np.array([50, 0], dtype=np.int64)
    po, lo = o.grid_subsampling(pts, lens, np.float32(0.5))
    pg, lg = gpu_grid(ext)(pts, lens, np.float32(0.5))
    assert np.array_equal(lo, lg) and np.array_equal(po, pg)
    assert np.array_equal(o.radius_neighbors(pts, pts, lens, lens, np.float32(1.0)), gpu_radius(ext, pts, pts, lens, lens, 1.0))


def test_synthetic_full_size_properties(ext):
    """BASELINE-size check through size-independent properties: voxel occupancy is idempotent
    (re-subsampling with the same voxel cannot merge further than one point per voxel changes),
    neighbour rows are sorted by distance, symmetric and contain the query itself."""
    from rdmnet_amd import synthetic
    ref, src, _ = synthetic.make_pair(3)
    pts = np.concatenate([ref, src])
    lens = np.array([len(ref), len(src)], dtype=np.int64)
    p1, l1 = gpu_grid(ext)(pts, lens, np.float32(0.6))
    assert l1.sum() == p1.shape[0] and 0 < p1.shape[0] < pts.shape[0]
    idx = gpu_radius(ext, p1, p1, l1, l1, 1.275)
    n = p1.shape[0]
    assert (idx[:, 0] == np.arange(n)).all()  # self first (d2 = 0)
    valid = idx < n
    pp = np.concatenate([p1, np.full((1, 3), 1e6, np.float32)])
    d = ((pp[idx] - p1[:, None]) ** 2).sum(-1)
    both = valid[:, 1:] & valid[:, :-1]
    assert (np.diff(d, axis=1)[both] >= 0).all()  # ascending distance
    assert (valid[:, :-1] | ~valid[:, 1:]).all()  # padding only at the tail
    rows = np.repeat(np.arange(n), valid.sum(1))
    cols = idx[valid]
    fwd = set(zip(rows.tolist()[:20000], cols.tolist()[:20000]))
    allp = set(zip(rows.tolist(), cols.tolist()))
    assert all((c, r) in allp for r, c in fwd)
    # clouds never mix
    assert (cols[rows < l1[0]] < l1[0]).all() and (cols[rows >= l1[0]] >= l1[0]).all()


def test_dense_neighbourhoods_use_the_large_buffer_pass(ext, oracle_native):
    """More than 256 neighbours per query: the second (1024-slot) pass must produce the same rows as the oracle;
    beyond 1024 the row is produced in rounds (radix select over the (d2, index) keys) -- the reference has no limit
    (radius_neighbors_cpu.cpp:36-64), neither has this: every neighbour, in order, full width and truncated."""
    o = oracle_native.restatement()
    rng = np.random.default_rng(7)
    pts = rng.uniform(-1.0, 1.0, size=(3000, 3)).astype(np.float32)
    lens = np.array([1800, 1200], dtype=np.int64)
    for radius in (0.7, 0.9):  # ~100-700 neighbours: rows on both sides of the 256 boundary
        io = o.radius_neighbors(pts, pts, lens, lens, np.float32(radius))
        ig = gpu_radius(ext, pts, pts, lens, lens, radius)
        assert io.shape[1] > 256
        assert np.array_equal(io, ig), radius
        assert np.array_equal(gpu_radius(ext, pts, pts, lens, lens, radius, width=70), io[:, :70])
    for radius in (1.5, 3.0):  # 1.5: ~1000-1500 per query (rows on both sides of 1024); 3.0: every point of the cloud (1800 / 1200)
        io = o.radius_neighbors(pts, pts, lens, lens, np.float32(radius))
        ig = gpu_radius(ext, pts, pts, lens, lens, radius)
        assert io.shape[1] > 1024
        assert np.array_equal(io, ig), radius
        assert np.array_equal(gpu_radius(ext, pts, pts, lens, lens, radius, width=81), io[:, :81])
        assert np.array_equal(gpu_radius(ext, pts, pts, lens, lens, radius, width=1100), io[:, :1100])
    # many points at the SAME distance (a regular lattice): ties are ordered by index through all radix digits
    g = np.stack(np.meshgrid(*[np.arange(14, dtype=np.float32)] * 3, indexing='ij'), -1).reshape(-1, 3) * np.float32(0.25)
    lens = np.array([len(g)], dtype=np.int64)
    io = o.radius_neighbors(g, g, lens, lens, np.float32(2.0))
    assert io.shape[1] > 1024
    assert np.array_equal(io, gpu_radius(ext, g, g, lens, lens, 2.0))


def test_multi_launch_subsampling_equals_single_workgroup_form_and_oracle(scans, oracle_native):
    """rdm_grid_subsample_form: the multi-launch form (keys / ranks / lists / sums spread over the GPU, only the hash-map order
    replay on one workgroup per cloud) returns the SAME points in the SAME order as the one-workgroup-per-cloud kernel and as
    the reference's C++ (grid_subsampling_cpu.cpp:3-75) -- bundled scans at full size, random clouds, crowded voxels (one voxel
    holding thousands of points), a batch with a tiny and an empty cloud, capacity larger than the stacked rows."""
    from rdmnet_amd import ops
    o = oracle_native.reference() or oracle_native.restatement()
    rng = np.random.default_rng(11)
    cases = []
    a, b = scans['s000000'], scans['s000004']
    cases.append((np.concatenate([a, b]), np.array([len(a), len(b)], np.int64), 0.6, 0))
    cases.append((np.concatenate([a, b]), np.array([len(a), len(b)], np.int64), 0.3, 5000))       # capacity rows beyond the clouds
    big = (rng.uniform(-1, 1, (60000, 3)) * np.array([80, 80, 4])).astype(np.float32)
    cases.append((big, np.array([30000, 0, 29993, 7], np.int64), 0.9, 0))                      # empty and tiny clouds in the batch
    crowded = np.concatenate([rng.normal(0, 0.05, (5000, 3)), rng.uniform(-30, 30, (20000, 3))]).astype(np.float32)
    cases.append((crowded[rng.permutation(len(crowded))], np.array([12000, 13000], np.int64), 1.2, 0))
    cases.append(((rng.uniform(-1, 1, (200000, 3)) * np.array([100, 100, 10])).astype(np.float32), np.array([200000], np.int64), 0.45, 0))
    for pts, lens, voxel, extra in cases:
        dev = torch.from_numpy(np.concatenate([pts, np.zeros((extra, 3), np.float32)])).cuda()
        dl = torch.from_numpy(lens).cuda()
        outs = []
        for form in (1, 2, 0):
            p, l = ops.grid_subsample_device(dev, dl, voxel, form=form)
            l = l.cpu().numpy()
            outs.append((p[:int(l.sum())].cpu().numpy(), l))
        # (the reference's C++ reads the first point of an empty cloud: the restatement checks the batch that has one)
        chk = oracle_native.restatement() if (lens == 0).any() else o
        rp, rl = chk.grid_subsampling(pts, lens, np.float32(voxel))
        for p, l in outs:
            assert np.array_equal(l, rl), (lens, voxel)
            assert np.array_equal(p, rp), (lens, voxel)


def test_grid_build_of_a_degenerate_cloud_is_not_quadratic():
    """rn_rank_kernel placed a record by a serial loop over all members of its cell -- fine for voxel-subsampled levels
    (tens of points per cell), 10^9 dependent loads for 32 k points in ONE cell (raw scans, coincident points, a clamped cell
    grid).  Crowded cells are now counted by the whole wavefront: the build of such a cloud stays in the milliseconds and the
    records come out in (cloud, cell, row) order as for any other cloud."""
    import time
    from rdmnet_amd import ops
    g = torch.Generator().manual_seed(3)
    n = 32768
    pts = (torch.rand(2 * n, 3, generator=g) * 0.05).cuda()  # every point of a cloud inside one cell of a 1 m grid
    lengths = torch.tensor([n, n], dtype=torch.int64).cuda()
    ops.radius_grid_records(pts[:64], torch.tensor([32, 32]).cuda(), 1.0)  # (first-call costs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rec = ops.radius_grid_records(pts, lengths, 1.0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rows = rec[:, 3].contiguous().view(torch.int32).cpu().numpy()
    assert np.array_equal(rows, np.arange(2 * n)), 'one cell per cloud: the records are the rows in order'
    assert torch.equal(rec[:, :3], pts)
    assert dt < 0.25, f'grid build of a one-cell cloud took {dt * 1e3:.1f} ms'


def test_grid_records_are_a_function_of_the_points_alone():
    """Round 4: the cell-sorted records of a search grid (rdm_radius_grid_records) also order the queries of the KPConv tile
    kernel, whose GroupNorm partials follow its workgroups -- so the order must not depend on how the build's atomics raced:
    cell segments lie in cell order (prefix over cells) and the points of a cell in the order of their rows.  Two builds give the
    same bytes; the records are a permutation of the rows, grouped by cloud, with ascending rows inside every run of one cell."""
    from rdmnet_amd import ops
    g = torch.Generator().manual_seed(11)
    n0, n1 = 9000, 7000
    pts = torch.cat([torch.randn(n0, 3, generator=g) * torch.tensor([20.0, 20.0, 1.0]),
                     torch.randn(n1, 3, generator=g) * torch.tensor([15.0, 25.0, 1.0]) + 3.0]).cuda()
    pts[100:400] = pts[100]  # a crowded cell (300 coincident points)
    lengths = torch.tensor([n0, n1], dtype=torch.int64).cuda()
    radius = 1.5
    a = ops.radius_grid_records(pts, lengths, radius).clone()
    for _ in range(3):
        torch.cuda.synchronize()
        assert torch.equal(ops.radius_grid_records(pts, lengths, radius), a)
    rows = a[:, 3].contiguous().view(torch.int32).cpu().numpy()
    assert sorted(rows.tolist()) == list(range(n0 + n1))
    assert torch.equal(a[:, :3].cpu(), pts.cpu()[rows])
    assert rows[:n0].max() < n0 and rows[n0:].min() >= n0  # cloud by cloud
    # the order is (cloud, cell, row): cells recomputed here with the kernel's fp32 arithmetic (rn_bbox_kernel / cell_of: one
    # bounding box for both clouds, cell edge 1.001 r, x fastest)
    p32 = pts.cpu().numpy()
    lo, hi = p32.min(0), p32.max(0)
    cell = np.float32(radius) * np.float32(1.001) * np.float32(1.0)
    dim = (np.floor((hi - lo).astype(np.float64) / np.float64(cell)) + 1).astype(np.int64)
    inv = np.float32(1.0) / cell
    xyz = a[:, :3].cpu().numpy()
    c3 = np.clip(np.floor((xyz - lo) * inv).astype(np.int64), 0, dim - 1)
    lin = (c3[:, 2] * dim[1] + c3[:, 1]) * dim[0] + c3[:, 0]
    cloud = (rows >= n0).astype(np.int64)
    key = np.stack([cloud, lin, rows.astype(np.int64)], 1)
    order = np.lexsort((key[:, 2], key[:, 1], key[:, 0]))
    assert np.array_equal(order, np.arange(len(rows))), 'records are not in (cloud, cell, row) order'


def test_one_column_search_is_the_first_column_of_the_full_row():
    """A plain engine run builds its up-sampling tables one column wide (nearest neighbour only); the search then takes the
    smallest (distance, index) key with a wavefront minimum instead of sorting the row.  Same column as the full search,
    also for queries without any neighbour and with equidistant candidates (a lattice)."""
    from rdmnet_amd import ops
    g = torch.Generator().manual_seed(11)
    lattice = torch.stack(torch.meshgrid(*[torch.arange(12.0)] * 3, indexing='ij'), -1).reshape(-1, 3) * 0.5
    for s_pts, radius in ((torch.rand(5000, 3, generator=g) * torch.tensor([20.0, 20.0, 4.0]), 1.1), (lattice, 0.8)):
        q_pts = torch.cat([s_pts[::3] + 0.0, torch.rand(300, 3, generator=g) * 40.0 + 30.0])  # (the last 300: far from every support)
        ql = torch.tensor([q_pts.shape[0] // 2, q_pts.shape[0] - q_pts.shape[0] // 2])
        sl = torch.tensor([s_pts.shape[0] // 2, s_pts.shape[0] - s_pts.shape[0] // 2])
        args = (q_pts.cuda(), s_pts.cuda(), ql.cuda(), sl.cuda(), radius)
        full = ops.radius_search_device(*args, 40, torch.zeros(2, dtype=torch.int32, device='cuda'))
        one = ops.radius_search_device(*args, 1, torch.zeros(2, dtype=torch.int32, device='cuda'))
        assert one.shape[1] == 1 and torch.equal(one[:, 0], full[:, 0])
        assert (one[:, 0] == s_pts.shape[0]).sum().item() >= 150  # queries with no neighbour hold the shadow index
