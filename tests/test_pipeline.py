"""CPU: the scheduler of rdmnet_amd.pipeline.PairPipeline (ordered results, lazy shared job queue, bounded run-ahead,
error propagation) on injected stand-in engines -- no compute, no GPU.  The GPU half (N pairs in flight == serial run,
bit for bit) is tests/test_pipeline_gpu.py."""
import threading
import time

import pytest

from rdmnet_amd import pipeline


class FakeEngine:
    def __init__(self, k):
        self.k, self.jobs = k, []

    def set_wait(self, us):
        self.wait = us

    def set_pairs_in_flight(self, n):
        self.n = n

    def keep_taps(self, on):
        self.taps = on


def make(n, **kw):
    return pipeline.PairPipeline(None, None, pairs_in_flight=n, engines=[FakeEngine(k) for k in range(n)], stagger_ms=0.2, **kw)


def test_results_come_back_in_job_order_whatever_the_completion_order():
    p = make(4)
    assert all(e.n == 4 for e in p.engines)

    def fn(eng, job):
        time.sleep(0.001 * ((job * 7) % 5))  # later jobs often finish first
        eng.jobs.append(job)
        return job * job
    assert p.map(range(40), fn) == [j * j for j in range(40)]
    assert sorted(j for e in p.engines for j in e.jobs) == list(range(40))
    assert sum(1 for e in p.engines if e.jobs) >= 2  # the queue is shared: more than one worker drew from it


def test_jobs_are_drawn_lazily_and_run_ahead_is_bounded():
    p = make(2)
    drawn = []

    def source():
        for i in range(50):
            drawn.append(i)
            yield i
    seen = []
    for out in p.imap(source(), lambda eng, job: job, window=4):
        seen.append(out)
        time.sleep(0.002)
        assert len(drawn) - len(seen) <= 4 + 2  # window results + one job in each worker's hands
    assert seen == list(range(50))


def test_a_failing_job_surfaces_in_the_consumer_and_stops_the_workers():
    p = make(3)

    def fn(eng, job):
        if job == 7:
            raise ValueError('job 7 failed')
        time.sleep(0.001)
        return job
    with pytest.raises(ValueError, match='job 7'):
        p.map(range(1000), fn)
    assert threading.active_count() < 8  # the workers were joined


def test_single_pair_in_flight_runs_on_the_callers_stream_serially():
    p = make(1)
    assert p.streams == [None]
    assert p.map(range(5), lambda eng, job: (eng.k, job)) == [(0, j) for j in range(5)]


def test_pin_rank_slices_the_affinity_mask():
    import os
    before = sorted(os.sched_getaffinity(0))
    if len(before) < 2:
        pytest.skip('one CPU')
    try:
        assert pipeline.pin_rank(0, 1) is None  # single rank: untouched
        mine = pipeline.pin_rank(1, 2)
        per = len(before) // 2
        assert mine == before[per:2 * per] and sorted(os.sched_getaffinity(0)) == mine
    finally:
        os.sched_setaffinity(0, before)
    assert pipeline.choose_wait_us(4, local_world=10 ** 6) == 50  # no cores to spin on
