"""CPU: properties of the (parity-unpinned) oracle for raw-scan voxel down-sampling and correspondence RANSAC
(Open3D is not part of the reference tree, so there is no golden to pin against)."""
import numpy as np

from oracle import preprocess


def test_voxel_down_sample_properties():
    rng = np.random.default_rng(0)
    pts = np.concatenate([rng.uniform(-40, 40, (20000, 2)), rng.uniform(-2, 1, (20000, 1)), rng.uniform(0, 1, (20000, 1))],
                         1).astype(np.float32)
    out = preprocess.voxel_down_sample(pts, 0.3)
    assert out.dtype == np.float32 and out.shape[1] == 4 and 0 < out.shape[0] <= pts.shape[0]
    # every input point lies in the voxel of exactly one output centroid; centroids are means -> inside the bbox
    assert (out[:, :3].min(0) >= pts[:, :3].min(0) - 1e-6).all() and (out[:, :3].max(0) <= pts[:, :3].max(0) + 1e-6).all()
    # total mass is preserved: sum_v count_v * centroid_v = sum of points
    lo = pts[:, :3].astype(np.float64).min(0) - 0.15
    key = np.floor((pts[:, :3].astype(np.float64) - lo) / 0.3).astype(np.int64)
    _, inv, cnt = np.unique(key, axis=0, return_inverse=True, return_counts=True)
    assert out.shape[0] == cnt.shape[0]
    # first-occurrence order: the first output voxel is the one of point 0
    assert np.array_equal(np.floor((out[0, :3].astype(np.float64) - lo) / 0.3).astype(np.int64), key[0])
    # idempotent on already isolated points
    far = (np.arange(30, dtype=np.float32)[:, None] * np.array([[1.0, 2.0, 3.0, 0.0]], np.float32))
    assert np.array_equal(preprocess.voxel_down_sample(far, 0.3), far)
    assert preprocess.voxel_down_sample(np.zeros((0, 4), np.float32), 0.3).shape == (0, 4)


def test_ransac_recovers_a_planted_transform():
    rng = np.random.default_rng(1)
    src = rng.uniform(-20, 20, (400, 3)).astype(np.float32)
    ang = 0.4
    R = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]])
    t = np.array([3.0, -1.0, 0.5])
    ref = (src @ R.T + t + rng.normal(0, 0.02, src.shape)).astype(np.float32)
    ref[:240] = rng.uniform(-20, 20, (240, 3)).astype(np.float32)  # 60 % outliers
    T, best, inl, rmse, counts = preprocess.ransac_correspondences(src, ref, 0.3, 4, 3000, seed=7)
    assert best >= 0 and inl >= 150 and rmse < 0.1
    assert np.abs(T[:3, :3] - R).max() < 0.02 and np.abs(T[:3, 3] - t).max() < 0.2
    assert counts.shape == (3000,) and counts[best] == counts.max()
    d = preprocess.ransac_draws(400, 4, 3000, 7)
    assert d.min() >= 0 and d.max() < 400 and len(np.unique(d)) > 350        # uniform cover
    assert not np.array_equal(d, preprocess.ransac_draws(400, 4, 3000, 8))     # seed matters
    T0, b0, i0, _, _ = preprocess.ransac_correspondences(src[:2], ref[:2], 0.3, 4, 10)
    assert b0 == -1 and i0 == 0 and np.array_equal(T0, np.eye(4))              # fewer correspondences than ransac_n
