"""Test helper: the reference's 5-level pyramid recipe (geotransformer/utils/data.py:13-77) driven
through any implementation of the two native operators (numpy in / numpy out)."""
import hashlib

import numpy as np

LIMITS = [65, 63, 69, 70, 81]
VOXEL0, RADIUS0 = 0.3, 0.3 * 4.25


def build_levels(grid_subsampling, pts, lens, num_stages=5):
    P, L = [pts], [lens]
    voxel = VOXEL0
    for _ in range(1, num_stages):
        voxel *= 2  # doubled BEFORE first use: level 1 uses 0.6 (data.py:23-28)
        p, l = grid_subsampling(P[-1], L[-1], np.float32(voxel))
        P.append(p), L.append(l)
    return P, L


def search_calls(P, L):
    radius = RADIUS0
    for lvl in range(len(P)):
        yield f'self{lvl}', P[lvl], P[lvl], L[lvl], L[lvl], radius, LIMITS[lvl]
        if lvl < len(P) - 1:
            yield f'sub{lvl}', P[lvl + 1], P[lvl], L[lvl + 1], L[lvl], radius, LIMITS[lvl]
            yield f'up{lvl}', P[lvl], P[lvl + 1], L[lvl], L[lvl + 1], radius * 2, LIMITS[lvl + 1]
        radius *= 2


def sha(idx):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(idx.astype(np.int64)).tobytes()).digest(), dtype=np.uint8)
