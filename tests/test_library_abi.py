"""CPU: the C-ABI library loads and exports every symbol include/rdmnet_hip.h declares (no compute
calls without a GPU), and the host-only helpers behave."""
import ctypes
import os
import re

import numpy as np

from rdmnet_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'rdmnet_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(rdm_[a-z0-9_]+)\s*\(', text)))


def test_every_declared_symbol_is_exported_and_bound():
    L = _lib.lib()
    names = declared_symbols()
    assert len(names) >= 6
    for n in names:
        assert hasattr(L, n), f'{n} declared in include/rdmnet_hip.h but not exported'
        assert n in _lib.SIGNATURES, f'{n} has no ctypes signature in rdmnet_amd/_lib.py'
    assert L.rdm_abi_version() == _lib.ABI_VERSION == 2


def test_rehash_schedule_matches_a_real_unordered_map(oracle_native):
    """rdm_rehash_schedule must describe the growth of the std::unordered_map the oracle uses: the
    number of voxels a cloud yields equals map.size(), and the first growth points are known."""
    L = _lib.lib()
    at = np.zeros(64, np.int64)
    bk = np.zeros(64, np.int64)
    n = L.rdm_rehash_schedule(1 << 20, at.ctypes.data, bk.ctypes.data, 64)
    assert n > 10
    assert at[0] == 0 and all(at[1:n] == bk[: n - 1])  # load factor 1.0: grow when size == buckets
    assert all(bk[1:n] > 2 * bk[: n - 1])  # growth factor 2, rounded up to a prime


def test_missing_gpu_raises_instead_of_falling_back():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from rdmnet_amd import ext
    with pytest.raises(RuntimeError):
        ext.grid_subsampling(torch.zeros(4, 3), torch.tensor([4]), 0.5)
    with pytest.raises(RuntimeError):  # dtype check mirrors CHECK_IS_FLOAT
        ext.grid_subsampling(torch.zeros(4, 3, dtype=torch.float64), torch.tensor([4]), 0.5)
