"""Deterministic down-sampling of large tensors for golden fixtures (shared by the generator and
the tests, so both sides pick the same rows/columns)."""
import numpy as np


def sample(a, max_rows=96, max_cols=96):
    a = np.asarray(a)
    if a.ndim == 0:
        return a
    rs = max(1, -(-a.shape[0] // max_rows))
    a = a[::rs]
    if a.ndim >= 2 and a.shape[-1] > max_cols:
        cs = -(-a.shape[-1] // max_cols)
        a = a[..., ::cs]
    return np.ascontiguousarray(a)


def compact_scores(ms, row_mask, col_mask):
    """Valid (non-masked) part of Sinkhorn outputs: per patch the (nr+1, nc+1) block, flattened."""
    vals = []
    for b in range(ms.shape[0]):
        r = np.concatenate([np.nonzero(row_mask[b])[0], [ms.shape[1] - 1]])
        c = np.concatenate([np.nonzero(col_mask[b])[0], [ms.shape[2] - 1]])
        vals.append(ms[b][np.ix_(r, c)].reshape(-1))
    return np.concatenate(vals).astype(np.float32)


def expand_scores(vals, row_mask, col_mask, fill=-1e12):
    b, m, n = row_mask.shape[0], row_mask.shape[1] + 1, col_mask.shape[1] + 1
    out = np.full((b, m, n), np.float32(fill), dtype=np.float32)
    off = 0
    for i in range(b):
        r = np.concatenate([np.nonzero(row_mask[i])[0], [m - 1]])
        c = np.concatenate([np.nonzero(col_mask[i])[0], [n - 1]])
        k = len(r) * len(c)
        out[i][np.ix_(r, c)] = vals[off:off + k].reshape(len(r), len(c))
        off += k
    return out
