"""Generates tests/golden/scans.npz and tests/golden/native_golden.npz.  Build container only.

scans.npz          the reference's bundled scans (assets/pc/00000{0,4,7}.npy, xyz columns) -- data
                   fixtures, used as golden INPUTS on the GPU box where /root/reference is absent.
native_golden.npz  outputs of the reference's own native code (oracle/_ref, i.e. its C++ compiled
                   where it lies) on those inputs through the reference pyramid recipe
                   (geotransformer/utils/data.py:13-77): subsampled clouds for every level, and for
                   every radius search its width, per-row counts and a strided sample of rows.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import native  # noqa: E402

REF = '/root/reference'
LIMITS = [65, 63, 69, 70, 81]


def main():
    ref = native.reference()
    assert ref is not None
    scans = {n: np.load(f'{REF}/assets/pc/{n}.npy')[:, :3].astype(np.float32) for n in ('000000', '000004', '000007')}
    np.savez_compressed(os.path.join(HERE, 'scans.npz'), **{'s' + k: v for k, v in scans.items()})
    out = {}
    for a, b in (('000000', '000004'), ('000000', '000007')):
        tag = f'{a}_{b}'
        pts = np.concatenate([scans[a], scans[b]])
        lens = np.array([len(scans[a]), len(scans[b])], dtype=np.int64)
        P, L = [pts], [lens]
        voxel = 0.3
        for lvl in range(1, 5):
            voxel *= 2
            p, l = ref.grid_subsampling(P[-1], L[-1], np.float32(voxel))
            P.append(p), L.append(l)
            out[f'{tag}/points{lvl}'] = p
            out[f'{tag}/lengths{lvl}'] = l
        radius = 0.3 * 4.25
        for lvl in range(5):
            calls = [('self', P[lvl], P[lvl], L[lvl], L[lvl], radius)]
            if lvl < 4:
                calls.append(('sub', P[lvl + 1], P[lvl], L[lvl + 1], L[lvl], radius))
                calls.append(('up', P[lvl], P[lvl + 1], L[lvl], L[lvl + 1], radius * 2))
            for name, q, s, ql, sl, r in calls:
                idx = ref.radius_neighbors(q, s, ql, sl, np.float32(r))
                idx = native.canonicalize_ties(q, s, idx)
                lim = LIMITS[lvl + 1] if name == 'up' else LIMITS[lvl]
                key = f'{tag}/{name}{lvl}'
                out[key + '/width'] = np.int64(idx.shape[1])
                out[key + '/counts'] = (idx < s.shape[0]).sum(1).astype(np.int32)
                out[key + '/rows'] = np.arange(0, idx.shape[0], 97, dtype=np.int64)
                out[key + '/sample'] = idx[::97, :lim].astype(np.int32)
                out[key + '/sha'] = np.frombuffer(
                    hashlib.sha256(np.ascontiguousarray(idx[:, :lim]).tobytes()).digest(), dtype=np.uint8)
            radius *= 2
    np.savez_compressed(os.path.join(HERE, 'native_golden.npz'), **out)
    print('wrote', len(out), 'arrays')


if __name__ == '__main__':
    main()
