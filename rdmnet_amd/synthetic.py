"""Seeded synthetic KITTI-shaped scan pairs: there is no dataset on the bench box.

A 64-beam spinning lidar is ray-cast against a ground plane, axis-aligned boxes (buildings,
cars) and poles; the second scan sees the same scene from a moved pose, so the ground-truth
transform (src -> ref) is known.  Scans are centroid-voxel-downsampled at 0.3 m like the
reference's preprocessing (preporcess/downsample_pcd_kitti.py:28) and sweeps are
accumulated / thinned until a scan has `target_points` +- `tolerance` points.
"""
import numpy as np

SENSOR_HEIGHT = 1.73
MAX_RANGE = 80.0


def _make_scene(rng):
    boxes = []
    for _ in range(int(rng.integers(30, 61))):  # buildings along the road sides
        side = 1.0 if rng.random() < 0.5 else -1.0
        sx, sy, sz = rng.uniform(5, 30), rng.uniform(5, 30), rng.uniform(4, 15)
        cx = rng.uniform(-90, 110)
        cy = side * (rng.uniform(9, 55) + sy / 2)
        boxes.append([cx - sx / 2, cy - sy / 2, -SENSOR_HEIGHT, cx + sx / 2, cy + sy / 2, -SENSOR_HEIGHT + sz])
    for _ in range(int(rng.integers(15, 36))):  # parked cars
        side = 1.0 if rng.random() < 0.5 else -1.0
        cx, cy = rng.uniform(-60, 80), side * rng.uniform(3.0, 7.0)
        boxes.append([cx - 2.0, cy - 1.0, -SENSOR_HEIGHT, cx + 2.0, cy + 1.0, -SENSOR_HEIGHT + 1.5])
    for _ in range(int(rng.integers(60, 121))):  # shrubs / clutter
        side = 1.0 if rng.random() < 0.5 else -1.0
        cx, cy = rng.uniform(-75, 95), side * rng.uniform(6.0, 50.0)
        r, h = rng.uniform(0.4, 1.5), rng.uniform(0.5, 3.0)
        boxes.append([cx - r, cy - r, -SENSOR_HEIGHT, cx + r, cy + r, -SENSOR_HEIGHT + h])
    for _ in range(int(rng.integers(20, 41))):  # poles / trunks
        side = 1.0 if rng.random() < 0.5 else -1.0
        cx, cy, r = rng.uniform(-70, 90), side * rng.uniform(5.0, 9.0), rng.uniform(0.1, 0.25)
        boxes.append([cx - r, cy - r, -SENSOR_HEIGHT, cx + r, cy + r, -SENSOR_HEIGHT + rng.uniform(3, 8)])
    return np.asarray(boxes, dtype=np.float64)


def _scan(boxes, pose_xy, yaw, n_azimuth, rng, elev_jitter_deg=0.0):
    """Points in the SENSOR frame for a sensor at world (pose_xy, z=0) with heading `yaw`."""
    elev = np.deg2rad(np.linspace(-24.8, 2.0, 64) + elev_jitter_deg)
    azim = np.linspace(-np.pi, np.pi, n_azimuth, endpoint=False) + rng.uniform(0, 2 * np.pi / n_azimuth)
    ce, se = np.cos(elev)[:, None], np.sin(elev)[:, None]
    d_s = np.stack([ce * np.cos(azim)[None], ce * np.sin(azim)[None], np.broadcast_to(se, (64, n_azimuth))], -1)
    d_s = d_s.reshape(-1, 3).astype(np.float32)
    c, s = np.cos(yaw), np.sin(yaw)
    R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]], dtype=np.float32)
    d_w = d_s @ R.T
    o = np.array([pose_xy[0], pose_xy[1], 0.0], dtype=np.float32)
    boxes = boxes.astype(np.float32)
    t_hit = np.full(d_w.shape[0], np.inf, dtype=np.float32)
    dz = d_w[:, 2]
    with np.errstate(divide='ignore', invalid='ignore'):
        tg = (-SENSOR_HEIGHT - o[2]) / dz
        t_hit = np.where((dz < 0) & (tg > 0), tg, t_hit)
        inv = 1.0 / d_w
        for bx in boxes:
            t0 = (bx[:3] - o) * inv
            t1 = (bx[3:] - o) * inv
            tn = np.minimum(t0, t1).max(axis=1)
            tf = np.maximum(t0, t1).min(axis=1)
            ok = (tn <= tf) & (tf > 0) & (tn > 0.5)
            t_hit = np.where(ok & (tn < t_hit), tn, t_hit)
    keep = np.isfinite(t_hit) & (t_hit < MAX_RANGE) & (t_hit > 1.5)
    rng_noise = rng.normal(0.0, 0.02, size=int(keep.sum()))
    return (d_s[keep] * (t_hit[keep] + rng_noise)[:, None]).astype(np.float32)


def voxel_centroids(points, voxel):
    """Centroid voxel downsample (Open3D voxel_down_sample semantics), deterministic order."""
    key = np.floor(points.astype(np.float64) / voxel).astype(np.int64)
    _, inv, cnt = np.unique(key, axis=0, return_inverse=True, return_counts=True)
    inv = inv.reshape(-1)
    out = np.zeros((cnt.shape[0], 3), dtype=np.float64)
    np.add.at(out, inv, points.astype(np.float64))
    return (out / cnt[:, None]).astype(np.float32)


def _scan_to_target(boxes, pose, yaw, rng_seed, target, tol):
    """Sweep until the 0.3 m voxel set reaches the target (extra sweeps are elevation-jittered, like
    an accumulated scan), then thin uniformly to the target."""
    raw = []
    for sweep in range(6):
        rng = np.random.default_rng(rng_seed + 17 * sweep)
        raw.append(_scan(boxes, pose, yaw, 2048, rng, elev_jitter_deg=0.0 if sweep == 0 else 0.21 * sweep))
        pts = voxel_centroids(np.concatenate(raw), 0.3)
        if pts.shape[0] >= target - tol:
            break
    if pts.shape[0] > target + tol:
        keep = np.random.default_rng(rng_seed + 1).permutation(pts.shape[0])[:target]
        pts = pts[np.sort(keep)]
    return pts


def make_pair(pair_id, target_points=16000, tolerance=500):
    """Returns (ref_points f32[N,3], src_points f32[M,3], transform f64[4,4] mapping src -> ref)."""
    rng = np.random.default_rng(1000 + int(pair_id))
    boxes = _make_scene(rng)
    fwd, lat, yaw = rng.uniform(5, 15), rng.uniform(-0.5, 0.5), np.deg2rad(rng.uniform(-15, 15))
    ref = _scan_to_target(boxes, (0.0, 0.0), 0.0, 5000 + pair_id, target_points, tolerance)
    src = _scan_to_target(boxes, (fwd, lat), yaw, 9000 + pair_id, target_points, tolerance)
    c, s = np.cos(yaw), np.sin(yaw)
    T = np.eye(4)
    T[:3, :3] = [[c, -s, 0], [s, c, 0], [0, 0, 1]]
    T[:3, 3] = [fwd, lat, 0.0]
    # deterministic shuffle: real scans are not voxel-key ordered
    ref = ref[np.random.default_rng(1).permutation(ref.shape[0])]
    src = src[np.random.default_rng(2).permutation(src.shape[0])]
    return ref, src, T


def make_low_overlap_pair(pair_id, target_points=16000, tolerance=500, fov_loss_deg=70.0):
    """BASELINE.json configs[4] (Mulran-shaped): the second scan is taken >= 10 m away under an arbitrary
    yaw and loses a `fov_loss_deg` azimuth sector of its own field of view (Mulran's Ouster is occluded
    to the rear).  Irregular neighbourhoods at the sector's edges and far fewer valid patch pairs than the
    KITTI-shaped `make_pair`.  Returns (ref, src, transform src -> ref)."""
    rng = np.random.default_rng(7000 + int(pair_id))
    boxes = _make_scene(rng)
    fwd, lat = rng.uniform(10, 18), rng.uniform(-2.0, 2.0)
    yaw = np.deg2rad(rng.uniform(-180, 180))
    ref = _scan_to_target(boxes, (0.0, 0.0), 0.0, 15000 + pair_id, target_points, tolerance)
    src = _scan_to_target(boxes, (fwd, lat), yaw, 19000 + pair_id, target_points, tolerance)
    centre = rng.uniform(-np.pi, np.pi)
    az = np.arctan2(src[:, 1], src[:, 0])
    d = np.abs(np.angle(np.exp(1j * (az - centre))))
    src = src[d > np.deg2rad(fov_loss_deg) / 2]
    c, s_ = np.cos(yaw), np.sin(yaw)
    T = np.eye(4)
    T[:3, :3] = [[c, -s_, 0], [s_, c, 0], [0, 0, 1]]
    T[:3, 3] = [fwd, lat, 0.0]
    ref = ref[np.random.default_rng(1).permutation(ref.shape[0])]
    src = src[np.random.default_rng(2).permutation(src.shape[0])]
    return ref, src, T


def cached_pairs(n_pairs, cache_dir, fixture=None):
    """The bench workload's pairs 0 .. n-1 (make_pair): from `fixture` (an .npz with ref<i> / src<i> / T<i>, e.g.
    tests/golden/synthetic_pairs.npz: the generator's exact output for the first pairs) where present, else from
    `cache_dir`, else generated (~10 s of host ray casting each) and cached there.  -> [(ref, src, T)]."""
    import os
    os.makedirs(cache_dir, exist_ok=True)
    fx = np.load(fixture) if fixture and os.path.exists(fixture) else None
    pairs = []
    for pid in range(n_pairs):
        if fx is not None and f'ref{pid}' in fx.files:
            pairs.append((fx[f'ref{pid}'], fx[f'src{pid}'], fx[f'T{pid}']))
            continue
        f = os.path.join(cache_dir, f'pair_{pid}.npz')
        if os.path.exists(f):
            z = np.load(f)
            pairs.append((z['ref'], z['src'], z['T']))
        else:
            ref, src, T = make_pair(pid)
            np.savez(f, ref=ref, src=src, T=T)
            pairs.append((ref, src, T))
    return pairs
