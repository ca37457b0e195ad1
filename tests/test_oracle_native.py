"""CPU: pins the oracle's native restatement (oracle/oracle_native.cpp) against
(1) golden vectors captured from the reference's own native code and (2) that code itself when
oracle/_ref is present."""
import os

import numpy as np
import pytest

from pyramid import build_levels, search_calls, sha

PAIRS = [('000000', '000004'), ('000000', '000007')]


@pytest.fixture(scope='module')
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, 'native_golden.npz'))


@pytest.mark.parametrize('pair', PAIRS)
def test_restatement_matches_reference_goldens(oracle_native, scans, golden, pair):
    o = oracle_native.restatement()
    a, b = scans['s' + pair[0]], scans['s' + pair[1]]
    tag = f'{pair[0]}_{pair[1]}'
    P, L = build_levels(o.grid_subsampling, np.concatenate([a, b]), np.array([len(a), len(b)], dtype=np.int64))
    for lvl in range(1, 5):  # bit-exact values AND order (libstdc++ unordered_map iteration order)
        assert np.array_equal(L[lvl], golden[f'{tag}/lengths{lvl}'])
        assert np.array_equal(P[lvl], golden[f'{tag}/points{lvl}'])
    for name, q, s, ql, sl, r, lim in search_calls(P, L):
        idx = o.radius_neighbors(q, s, ql, sl, np.float32(r))
        key = f'{tag}/{name}'
        assert idx.shape[1] == int(golden[key + '/width'])
        assert np.array_equal((idx < s.shape[0]).sum(1), golden[key + '/counts'])
        assert np.array_equal(idx[golden[key + '/rows'], :lim], golden[key + '/sample'])
        assert np.array_equal(sha(idx[:, :lim]), golden[key + '/sha'])


def test_restatement_matches_reference_build_on_random_clouds(oracle_native):
    ref = oracle_native.reference()
    if ref is None:
        pytest.skip('oracle/_ref not built (reference tree absent)')
    o = oracle_native.restatement()
    rng = np.random.default_rng(0)
    for n0, n1, scale in [(1, 1, 1.0), (5, 300, 3.0), (2000, 1500, 20.0), (4000, 1, 40.0)]:
        pts = (rng.standard_normal((n0 + n1, 3)) * scale).astype(np.float32)
        pts[: min(n0, 3)] = pts[0]  # duplicates -> exact distance ties
        lens = np.array([n0, n1], dtype=np.int64)
        for voxel in (0.6, 2.4):
            po, lo = o.grid_subsampling(pts, lens, np.float32(voxel))
            pr, lr = ref.grid_subsampling(pts, lens, np.float32(voxel))
            assert np.array_equal(lo, lr) and np.array_equal(po, pr)
            io = o.radius_neighbors(po, pts, lo, lens, np.float32(voxel * 2.125))
            ir = oracle_native.canonicalize_ties(po, pts, ref.radius_neighbors(po, pts, lo, lens, np.float32(voxel * 2.125)))
            assert np.array_equal(io, ir)


def test_empty_and_ragged_inputs(oracle_native):
    o = oracle_native.restatement()
    pts = np.zeros((3, 3), np.float32)
    p, l = o.grid_subsampling(pts, np.array([3, 0], dtype=np.int64), np.float32(0.5))
    assert l.tolist() == [1, 0] and p.shape == (1, 3)
    idx = o.radius_neighbors(pts, pts, np.array([3, 0], dtype=np.int64), np.array([3, 0], dtype=np.int64), np.float32(1.0))
    assert idx.shape == (3, 3) and sorted(idx[0].tolist()) == [0, 1, 2]
