"""TEST INFRASTRUCTURE ONLY (see oracle/forward.py's header): CPU restatement of the raw-scan preprocessing
 and of the RANSAC estimator over correspondences (rank 4).

PARITY UNPINNED: both are Open3D internals in the reference (`open3d==0.11.2`, requirements.txt:8; call sites
preporcess/downsample_pcd_kitti.py:21-36 and geotransformer/utils/open3d.py:173-203); Open3D is neither under
/root/reference nor installed here.  What is restated is Open3D's published arithmetic; the output order /
random sampling, which Open3D leaves unspecified, are fixed by a documented rule so that the HIP kernels can be
compared bit-for-bit with this file.
"""
import numpy as np


def voxel_down_sample(points, voxel):
    """points f32 [N, >=3] (xyz + extra channels such as intensity).  Open3D VoxelDownSample arithmetic:
    voxel_min_bound = min_bound - voxel/2; index = floor((p - voxel_min_bound)/voxel) in float64; per-voxel
    means in float64.  Voxels in first-occurrence order, sums in ascending point order."""
    pts = np.asarray(points, np.float32)
    if pts.shape[0] == 0:
        return np.zeros((0, pts.shape[1]), np.float32)
    p64 = pts.astype(np.float64)
    lo = p64[:, :3].min(0) - float(voxel) * 0.5
    idx = np.floor((p64[:, :3] - lo) / float(voxel)).astype(np.int64)
    key = idx[:, 0] | (idx[:, 1] << 21) | (idx[:, 2] << 42)
    _, first, inv = np.unique(key, return_index=True, return_inverse=True)
    order = np.argsort(first, kind='stable')          # voxel ids sorted by first occurrence
    rank = np.empty_like(order)
    rank[order] = np.arange(order.shape[0])
    e = rank[inv.reshape(-1)]
    sums = np.zeros((order.shape[0], pts.shape[1]), np.float64)
    np.add.at(sums, e, p64)                            # sequential, ascending point index
    cnt = np.bincount(e, minlength=order.shape[0]).astype(np.float64)
    return (sums / cnt[:, None]).astype(np.float32)


def _mix64(z):
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def ransac_draws(n_corr, ransac_n, iters, seed):
    """Correspondence index of draw j of iteration i: counter-based, with replacement (Open3D 0.11.2 draws with
    replacement from a global generator; the generator itself is unspecified, this rule is the build's)."""
    with np.errstate(over='ignore'):
        ctr = (np.arange(iters, dtype=np.uint64)[:, None] * np.uint64(ransac_n) + np.arange(ransac_n, dtype=np.uint64)[None]
               + np.uint64(1))
        z = _mix64(np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * ctr)
    return (z % np.uint64(n_corr)).astype(np.int64)


def ransac_correspondences(src, ref, distance_threshold, ransac_n=4, iters=50000, seed=0, chunk=4096):
    """Open3D RegistrationRANSACBasedOnCorrespondence with point-to-point estimation (no scaling):
    -> (transform 4x4 f64, best iteration, inliers, rmse, per-iteration inlier counts)."""
    src, ref = np.asarray(src, np.float32).astype(np.float64), np.asarray(ref, np.float32).astype(np.float64)
    n = src.shape[0]
    counts = np.zeros(iters, np.int64)
    rmse = np.zeros(iters)
    Ts = np.zeros((iters, 3, 4))
    if n >= ransac_n:
        draws = ransac_draws(n, ransac_n, iters, seed)
        for a in range(0, iters, chunk):
            d = draws[a:a + chunk]
            s, r = src[d], ref[d]                       # [b, k, 3]
            cs, cr = s.mean(1), r.mean(1)
            H = np.einsum('bka,bkc->bac', s - cs[:, None], r - cr[:, None])
            U, _, Vt = np.linalg.svd(H)
            V = np.swapaxes(Vt, 1, 2)
            det = np.sign(np.linalg.det(V @ np.swapaxes(U, 1, 2)))
            D = np.zeros_like(H)
            D[:, 0, 0] = D[:, 1, 1] = 1.0
            D[:, 2, 2] = det
            R = V @ D @ np.swapaxes(U, 1, 2)
            t = cr - np.einsum('bac,bc->ba', R, cs)
            Ts[a:a + chunk, :, :3], Ts[a:a + chunk, :, 3] = R, t
            diff = np.einsum('bac,nc->bna', R, src) + t[:, None] - ref[None]
            d2 = (diff ** 2).sum(2)
            inl = np.sqrt(d2) < float(np.float32(distance_threshold))
            counts[a:a + chunk] = inl.sum(1)
            e2 = (d2 * inl).sum(1)
            rmse[a:a + chunk] = np.where(inl.sum(1) > 0, np.sqrt(e2 / np.maximum(inl.sum(1), 1)), 0.0)
    best = -1
    for i in np.flatnonzero(counts == counts.max()) if counts.max() > 0 else []:
        if best < 0 or rmse[i] < rmse[best]:
            best = int(i)
    T = np.eye(4)
    if best >= 0:
        T[:3] = Ts[best]
    return T, best, int(counts[best]) if best >= 0 else 0, float(rmse[best]) if best >= 0 else 0.0, counts
