"""TEST INFRASTRUCTURE ONLY -- ctypes front-end of the oracle's native checkers.

`restatement` = oracle/liboracle_native.so (our CPU restatement, oracle_native.cpp)
`reference`   = oracle/_ref/libref_ext.so  (the reference's own C++ compiled by oracle/Makefile)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Both libraries expose the semantics of `rdmnet.ext` (reference pybind.cpp:6-17) on numpy arrays.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_c_f32p = ctypes.POINTER(ctypes.c_float)
_c_i64p = ctypes.POINTER(ctypes.c_int64)
_c_i32p = ctypes.POINTER(ctypes.c_int32)


def build(quiet=True):
    """Compile the checker libraries (building the checker is not using it)."""
    subprocess.run(['make', '-C', _HERE, 'all'], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


def _load(path):
    if not os.path.exists(path):
        return None
    return ctypes.CDLL(path)


class _Native:
    def __init__(self, lib, prefix):
        self.lib = lib
        self.prefix = prefix
        g = getattr(lib, prefix + '_grid_subsampling')
        g.restype = ctypes.c_int64
        g.argtypes = [_c_f32p, ctypes.c_int64, _c_i64p, ctypes.c_int64, ctypes.c_float,
                      ctypes.POINTER(_c_f32p), _c_i64p]
        r = getattr(lib, prefix + '_radius_neighbors')
        r.restype = ctypes.c_int64
        base = [_c_f32p, ctypes.c_int64, _c_f32p, ctypes.c_int64, _c_i64p, _c_i64p, ctypes.c_int64,
                ctypes.c_float, ctypes.POINTER(_c_i64p)]
        r.argtypes = base + ([_c_i32p] if prefix == 'oracle' else [])
        f = getattr(lib, prefix + '_free')
        f.restype = None
        f.argtypes = [ctypes.c_void_p]
        self._g, self._r, self._f = g, r, f

    def grid_subsampling(self, points, lengths, voxel):
        points = np.ascontiguousarray(points, dtype=np.float32)
        lengths = np.ascontiguousarray(lengths, dtype=np.int64)
        out = _c_f32p()
        out_len = np.zeros(lengths.shape[0], dtype=np.int64)
        m = self._g(points.ctypes.data_as(_c_f32p), points.shape[0], lengths.ctypes.data_as(_c_i64p),
                    lengths.shape[0], ctypes.c_float(voxel), ctypes.byref(out),
                    out_len.ctypes.data_as(_c_i64p))
        res = np.ctypeslib.as_array(out, shape=(m, 3)).copy() if m > 0 else np.zeros((0, 3), np.float32)
        self._f(out)
        return res, out_len

    def radius_neighbors(self, q, s, q_lengths, s_lengths, radius):
        q = np.ascontiguousarray(q, dtype=np.float32)
        s = np.ascontiguousarray(s, dtype=np.float32)
        ql = np.ascontiguousarray(q_lengths, dtype=np.int64)
        sl = np.ascontiguousarray(s_lengths, dtype=np.int64)
        out = _c_i64p()
        args = [q.ctypes.data_as(_c_f32p), q.shape[0], s.ctypes.data_as(_c_f32p), s.shape[0],
                ql.ctypes.data_as(_c_i64p), sl.ctypes.data_as(_c_i64p), ql.shape[0],
                ctypes.c_float(radius), ctypes.byref(out)]
        if self.prefix == 'oracle':
            args.append(None)
        w = self._r(*args)
        res = (np.ctypeslib.as_array(out, shape=(q.shape[0], w)).copy()
               if w > 0 and q.shape[0] > 0 else np.zeros((q.shape[0], 0), np.int64))
        self._f(out)
        return res


_cache = {}


def restatement():
    if 'o' not in _cache:
        path = os.path.join(_HERE, 'liboracle_native.so')
        if not os.path.exists(path):
            build()
        lib = _load(path)
        lib.oracle_neighbor_d2.restype = None
        lib.oracle_neighbor_d2.argtypes = [_c_f32p, ctypes.c_int64, _c_f32p, ctypes.c_int64, _c_i64p,
                                           ctypes.c_int64, _c_f32p]
        _cache['o'] = _Native(lib, 'oracle')
    return _cache['o']


def reference():
    """The reference's own native code, or None when oracle/_ref was never built."""
    if 'r' not in _cache:
        lib = _load(os.path.join(_HERE, '_ref', 'libref_ext.so'))
        _cache['r'] = _Native(lib, 'ref') if lib is not None else None
    return _cache['r']


def neighbor_d2(q, s, idx):
    """fp32 squared distances of listed neighbours (nanoflann metric); pad slots -> +inf."""
    o = restatement()
    q = np.ascontiguousarray(q, dtype=np.float32)
    s = np.ascontiguousarray(s, dtype=np.float32)
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    out = np.empty(idx.shape, dtype=np.float32)
    o.lib.oracle_neighbor_d2(q.ctypes.data_as(_c_f32p), q.shape[0], s.ctypes.data_as(_c_f32p),
                             s.shape[0], idx.ctypes.data_as(_c_i64p), idx.shape[1],
                             out.ctypes.data_as(_c_f32p))
    return out


def canonicalize_ties(q, s, idx):
    """Re-order each row so that equal-distance runs are ascending by index.

    nanoflann's std::sort leaves exact ties in arbitrary order (nanoflann.hpp:1280-1289);
    (d2, index) is the canonical order every implementation here is compared in.
    """
    d2 = neighbor_d2(q, s, idx)
    order = np.lexsort((idx, d2), axis=1)
    return np.take_along_axis(idx, order, axis=1)
