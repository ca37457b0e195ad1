"""Comparisons that tolerate exactly what the reference's own fp32 rounding leaves undecided, and nothing else.

Three places of the path amplify 1e-7 feature differences into another discrete outcome; the reference differs from
ITSELF there between its 8- and 1-thread CPU runs (tests/golden/oracle_vs_reference.json, `self/*` entries of the
golden files):
  * the ORDER of the top-256 superpoint pairs among scores tied to ~1e-6 relative (superpoint_matching.py:56-61),
  * the ORDER of a patch's points when two points are equidistant from the node within the cancellation noise of
    |x|^2 - 2xy + |y|^2 at 60-80 m coordinates (pairwise_distance.py:4-31, pointcloud_partition.py:92-101),
  * WHICH local hypothesis wins when inlier counts tie within one (local_global_registration.py:204-221).
The helpers below map one run's outputs onto the other's through those permutations and fail on anything else.
"""
import numpy as np


def pair_permutation(got_pairs, want_pairs, want_scores, tie=1e-5):
    """got/want: lists of (ref node, src node); want_scores: the scores at want's positions.  Returns perm with
    want_pairs[perm[i]] == got_pairs[i]; asserts equal sets and that a pair moved only inside a group of scores equal
    to `tie` relative."""
    assert len(got_pairs) == len(want_pairs) and set(got_pairs) == set(want_pairs), len(set(got_pairs) ^ set(want_pairs))
    assert len(set(want_pairs)) == len(want_pairs)
    pos = {p: i for i, p in enumerate(want_pairs)}
    perm = np.array([pos[p] for p in got_pairs], dtype=np.int64)
    s = np.asarray(want_scores, np.float64)
    gap = float(np.abs(s[perm] - s).max() / s.max()) if len(s) else 0.0
    assert gap <= tie, f'a superpoint pair moved across a score gap of {gap} (> {tie})'
    return perm, gap


def patch_permutation(got_pts, got_mask, want_pts, want_mask):
    """One patch: [K,3] points + [K] masks.  Returns perm [K] with want_pts[perm[i]] == got_pts[i] for the valid rows
    (valid rows come first in both, pointcloud_partition.py:92-107); asserts the same SET of valid points."""
    got_mask, want_mask = np.asarray(got_mask, bool), np.asarray(want_mask, bool)
    n = int(got_mask.sum())
    assert n == int(want_mask.sum()) and got_mask[:n].all() and want_mask[:n].all()
    where = {tuple(p): i for i, p in enumerate(np.asarray(want_pts)[:n].tolist())}
    assert len(where) == n
    perm = np.arange(len(got_mask), dtype=np.int64)
    for i, p in enumerate(np.asarray(got_pts)[:n].tolist()):
        assert tuple(p) in where, 'patch holds a point the other run does not'
        perm[i] = where[tuple(p)]
    assert sorted(perm[:n].tolist()) == list(range(n))
    return perm


def corr_rows(rc, sc, cs=None):
    """{(ref xyz, src xyz): score} of a set of point correspondences."""
    rc, sc = np.asarray(rc, np.float64), np.asarray(sc, np.float64)
    cs = np.zeros(len(rc)) if cs is None else np.asarray(cs, np.float64)
    return {tuple(r) + tuple(s): float(c) for r, s, c in zip(rc.tolist(), sc.tolist(), cs.tolist())}


def rre_rte(T, G):
    """RRE through ||R_err - I||_F (acos of the trace turns one fp32 ulp into 0.02 deg), RTE in metres."""
    T, G = np.asarray(T, np.float64), np.asarray(G, np.float64)
    R = G[:3, :3].T @ T[:3, :3]
    ang = 2.0 * np.arcsin(min(1.0, np.linalg.norm(R - np.eye(3)) / (2.0 * np.sqrt(2.0))))
    return float(np.degrees(ang)), float(np.linalg.norm(T[:3, 3] - G[:3, 3]))


def lgr_alternatives(ofw, cfg, ref_knn, src_knn, ref_mask, src_mask, log_scores, within=1):
    """The poses the reference's LGR (restated in oracle.forward.lgr, bit-exact against the reference on every golden
    case) returns for these inputs from each local hypothesis whose inlier count is within `within` of the best.
    Returns (correspondence outputs of the un-forced run, [(hypothesis, T)], margin between best and runner-up)."""
    rc, sc, cs, T, info = ofw.lgr(ref_knn, src_knn, ref_mask, src_mask, log_scores, cfg)
    if 'inlier_counts' not in info:
        return (rc, sc, cs), [(-1, T.numpy())], None
    counts = info['inlier_counts'].numpy()
    top = np.sort(counts)[::-1]
    near = [int(i) for i in np.nonzero(counts >= counts.max() - within)[0]]
    alts = [(i, ofw.lgr(ref_knn, src_knn, ref_mask, src_mask, log_scores, cfg, force_best=i)[3].numpy()) for i in near]
    return (rc, sc, cs), alts, int(top[0] - top[1]) if len(top) > 1 else int(top[0])
