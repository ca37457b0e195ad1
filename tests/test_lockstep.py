"""CPU: the lock-step scheduler (rdmnet_amd/csrc/lockstep.cpp) on scripted stand-in launches -- no GPU, no kernel.  Contexts run
scripts of recorded launches of two stand-in kernels (1 = A, 2 = B), waits (3), layer boundaries (4) and deferred events (5); the
library logs what the scheduler does: (1 | 2, n) = that kernel issued ONCE for n contexts, (3, i) = the i-th wait of the group,
(4, k) = context k's event recorded.  The GPU half (every pair the bits of its own run) is tests/test_engine_gpu.py."""
import ctypes

import numpy as np

from rdmnet_amd import _lib


def run(scripts):
    L = _lib.lib()
    n, ln = len(scripts), max(len(s) for s in scripts)
    arr = np.zeros((n, ln), dtype=np.int32)
    for k, s in enumerate(scripts):
        arr[k, :len(s)] = s
    log = np.zeros(512, dtype=np.int32)
    rcs = np.zeros(8, dtype=np.int32)
    got = L.rdm_lockstep_selftest(n, arr.ctypes.data, ln, log.ctypes.data, log.size, rcs.ctypes.data)
    assert 0 <= got <= log.size // 2
    assert list(rcs[:n]) == [100 + k for k in range(n)]  # every context ran to its end
    return [(int(log[2 * i]), int(log[2 * i + 1])) for i in range(got)]


def test_identical_sequences_go_out_as_one_launch_per_kernel_and_one_wait_per_read_back():
    assert run([[1, 2, 1, 3, 2]] * 4) == [(1, 4), (2, 4), (1, 4), (3, 1), (2, 4)]
    assert run([[1, 1, 3]]) == [(1, 1), (1, 1), (3, 1)]  # a group of one
    assert run([[1, 3, 3, 2]] * 8) == [(1, 8), (3, 1), (3, 2), (2, 8)]


def test_different_kernels_in_one_round_are_grouped_per_kernel_wherever_the_contexts_stand():
    # contexts 0 and 2 record A while 1 and 3 record B: two launches, each carrying its two records
    assert run([[1, 2], [2, 1], [1, 2], [2, 1]]) == [(1, 2), (2, 2), (2, 2), (1, 2)]


def test_a_context_that_launches_an_extra_kernel_stays_behind_until_a_layer_boundary():
    # context 0 has one launch more (a split-K reduce, say): without a boundary it is one launch behind for the rest of the run ...
    behind = run([[2, 1, 1, 1], [1, 1, 1], [1, 1, 1]])
    assert behind == [(2, 1), (1, 2), (1, 3), (1, 3), (1, 1)]
    # ... with boundaries (4) the others wait for it at the end of the layer and the next layers are grouped completely
    aligned = run([[2, 1, 4, 1, 4, 1], [1, 4, 1, 4, 1], [1, 4, 1, 4, 1]])
    assert aligned == [(2, 1), (1, 2), (1, 1), (1, 3), (1, 3)]
    # contexts that reach boundaries a different number of times are released together all the same: nobody hangs
    assert run([[4, 4, 1], [1]]) == [(1, 1), (1, 1)]


def test_waits_park_a_context_until_every_live_context_waits_and_ended_contexts_do_not_block():
    # context 1 ends early; context 0 waits twice: the group waits when nobody can run any more
    assert run([[1, 3, 1, 3], [1]]) == [(1, 2), (3, 1), (1, 1), (3, 2)]
    # context 0 reaches its wait two launches before context 1: ONE wait, after both have issued everything before it
    assert run([[3, 1], [1, 1, 3, 1]]) == [(1, 1), (1, 1), (3, 1), (1, 2)]


def test_deferred_events_are_recorded_right_before_their_contexts_next_launch_or_before_a_wait():
    # the profile events of a layer: ev, kernel, ev, kernel -- on the first context only (the others report shapes)
    assert run([[5, 1, 5, 2, 5, 3], [1, 2, 3]]) == [(4, 0), (1, 2), (4, 0), (2, 2), (4, 0), (3, 1)]
    # an event at the very end of a context is recorded when the group winds up (before the last wait, if there is one)
    assert run([[1, 5], [1, 3]]) == [(1, 2), (4, 0), (3, 1)]
