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


def test_workers_collate_several_drawn_jobs_at_once_and_shrink_the_batches_towards_the_end():
    """Round 5: with `tensors_of` a worker draws up to `collate_batch` jobs, hands their tensors to Engine.collate_batch ONCE and
    then runs fn on each; results still come back in job order; for a sized job list the last rounds are spread over the
    workers (20 jobs, 4 workers, batches of up to 8: 5 each, not 8 + 8 + 4 + 0); without `tensors_of`, with a batch of 1, or
    when stage tensors are kept, jobs are drawn one by one as before."""
    class Batching(FakeEngine):
        def __init__(self, k):
            super().__init__(k)
            self.batches = []

        def collate_batch(self, pairs):
            self.batches.append(list(pairs))

    def run(n_jobs, **kw):
        p = pipeline.PairPipeline(None, None, pairs_in_flight=4, engines=[Batching(k) for k in range(4)], stagger_ms=0.0, **kw)

        def fn(eng, job):
            eng.jobs.append(job)
            time.sleep(0.001)
            return job + 100
        out = p.map(range(n_jobs), fn, tensors_of=lambda job: ('ref%d' % job, 'src%d' % job))
        assert out == [j + 100 for j in range(n_jobs)]
        return p
    p = run(20, collate_batch=8)
    sizes = sorted(len(b) for e in p.engines for b in e.batches)
    assert sorted(j for e in p.engines for j in e.jobs) == list(range(20))  # every job ran once
    assert max(sizes) <= 5 and p.last_stats['collate_batches'] == len(sizes) and p.last_stats['jobs'] == 20
    for e in p.engines:  # a batch is collated with the tensors of exactly the jobs the worker then runs, in order
        flat = [t for b in e.batches for t in b]
        batched_jobs = [int(t[0][3:]) for t in flat]
        assert [j for j in e.jobs if j in set(batched_jobs)] == batched_jobs
    p = run(64, collate_batch=4)
    assert max(len(b) for e in p.engines for b in e.batches) == 4
    p = run(10, collate_batch=1)
    assert not any(e.batches for e in p.engines)
    p = run(10, collate_batch=4, keep_taps=True)
    assert not any(e.batches for e in p.engines)
    # no tensors_of: nothing to prepare
    p = pipeline.PairPipeline(None, None, pairs_in_flight=2, engines=[Batching(k) for k in range(2)], stagger_ms=0.0, collate_batch=4)
    assert p.map(range(6), lambda eng, job: job) == list(range(6)) and not any(e.batches for e in p.engines)


def test_workers_run_the_jobs_they_draw_as_one_lock_step_group():
    """Round 5: with `lockstep` = B a worker owns B engines, draws up to B jobs, calls prepare(engine, job, i, n) for each and
    hands them to run_lockstep ONCE (on exactly its engines, in job order); fn then sees job i on the group's i-th engine.  A single
    drawn job runs through fn alone; without tensors_of (or with stage tensors kept) jobs run one by one on the first engines."""
    calls = []

    class Grouped(FakeEngine):
        @staticmethod
        def run_lockstep(engines, pairs):
            calls.append(([e.k for e in engines], list(pairs)))

    def build(n, lb, **kw):
        return pipeline.PairPipeline(None, None, pairs_in_flight=n, engines=[Grouped(k) for k in range(n * lb)], stagger_ms=0.0, lockstep=lb, **kw)
    p = build(3, 4)
    assert [[e.k for e in g] for g in p.groups] == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10, 11]] and [e.k for e in p.engines] == [0, 4, 8]
    seen, prepared = [], []

    def fn(eng, job):
        seen.append((eng.k, job))
        time.sleep(0.001)
        return job * 2
    out = p.map(range(26), fn, tensors_of=lambda job: ('r%d' % job, 's%d' % job), prepare=lambda eng, job, i, n: prepared.append((eng.k, job, i, n)))
    assert out == [2 * j for j in range(26)] and sorted(j for _, j in seen) == list(range(26))
    assert p.last_stats['lockstep_groups'] == len(calls) and max(len(c[1]) for c in calls) == 4
    by_job = dict((j, k) for k, j in seen)
    for ks, pairs in calls:  # a group's jobs ran on its worker's engines, job i on engine i
        jobs = [int(r[1:]) for r, _ in pairs]
        assert ks[0] % 4 == 0 and ks == list(range(ks[0], ks[0] + 4)) and [by_job[j] for j in jobs] == ks[:len(jobs)]
    assert sorted(prepared) == sorted((by_job[j], j, by_job[j] % 4, n) for (_, j, _, n) in prepared) and len(prepared) == 26
    # the end of a sized job list: the remainder is split once, evenly (80 jobs on 4 x 4: twenty full groups, not 4 3 3 2 1 1 1 1)
    p = build(4, 4)
    assert p.map(range(80), fn, tensors_of=lambda job: (job, job)) == [2 * j for j in range(80)] and p.last_stats['group_sizes'] == {4: 20}
    p = build(3, 4)
    p.map(range(26), fn, tensors_of=lambda job: (job, job))
    assert p.last_stats['group_sizes'] == {4: 5, 3: 2}, p.last_stats['group_sizes']
    # no tensors_of: one by one
    calls.clear()
    assert build(2, 4).map(range(5), lambda eng, job: eng.k) and not calls
    # stage tensors kept (round 6): still lock-step groups -- the engines collate their own pairs inside the group
    p = build(2, 4, keep_taps=True)
    assert p.map(range(9), lambda eng, job: job, tensors_of=lambda job: (job, job)) == list(range(9)) and calls
    assert sum(sz * cnt for sz, cnt in p.last_stats['group_sizes'].items()) == 9
    # group_fn: the caller runs the group itself, on the worker's engines, and returns one result per job
    calls.clear()
    groups = []

    def group_fn(engines, jobs):
        groups.append(([e.k for e in engines], list(jobs)))
        return [j * 3 for j in jobs]
    p = build(2, 4)
    assert p.map(range(13), None, group_fn=group_fn) == [3 * j for j in range(13)] and not calls
    assert sorted(j for _, js in groups for j in js) == list(range(13)) and max(len(js) for _, js in groups) == 4
    assert all(ks == list(range(ks[0], ks[0] + len(js))) and ks[0] % 4 == 0 for ks, js in groups)
    with pytest.raises(RuntimeError):  # one result per job
        p.map(range(4), None, group_fn=lambda engines, jobs: [0])
    with pytest.raises(RuntimeError):  # 2 x 4 needs 8 engines when they are injected
        pipeline.PairPipeline(None, None, pairs_in_flight=2, engines=[Grouped(0)], lockstep=4)


def _fake_sysfs(root, gpu_nodes, node_cpulists):
    """A sysfs tree with one CPU agent and len(gpu_nodes) GPU agents in the KFD topology (GPU k on PCI bus 0x10 + k, NUMA node
    gpu_nodes[k]) and the nodes' cpulist files."""
    import os
    top = os.path.join(root, 'class', 'kfd', 'kfd', 'topology', 'nodes')
    os.makedirs(os.path.join(top, '0'))
    open(os.path.join(top, '0', 'properties'), 'w').write('cpu_cores_count 64\nsimd_count 0\nlocation_id 0\ndomain 0\n')
    for k, node in enumerate(gpu_nodes):
        os.makedirs(os.path.join(top, str(k + 1)))
        bus = 0x10 + k
        open(os.path.join(top, str(k + 1), 'properties'), 'w').write(f'cpu_cores_count 0\nsimd_count 1024\nlocation_id {bus << 8}\ndomain 0\n')
        d = os.path.join(root, 'bus', 'pci', 'devices', f'0000:{bus:02x}:00.0')
        os.makedirs(d)
        open(os.path.join(d, 'numa_node'), 'w').write(f'{node}\n')
    for node, text in node_cpulists.items():
        d = os.path.join(root, 'devices', 'system', 'node', f'node{node}')
        os.makedirs(d)
        open(os.path.join(d, 'cpulist'), 'w').write(text + '\n')


def test_gpu_numa_nodes_reads_the_kfd_topology(tmp_path):
    _fake_sysfs(str(tmp_path), [0, 0, 1, 1], {0: '0-3', 1: '4-7'})
    assert pipeline.gpu_numa_nodes(str(tmp_path), env={}) == [0, 0, 1, 1]
    assert pipeline.gpu_numa_nodes(str(tmp_path), env={'HIP_VISIBLE_DEVICES': '2,0'}) == [1, 0]
    assert pipeline.gpu_numa_nodes(str(tmp_path), env={'ROCR_VISIBLE_DEVICES': '1,2,3', 'HIP_VISIBLE_DEVICES': '1'}) == [1]
    assert pipeline.gpu_numa_nodes(str(tmp_path), env={'HIP_VISIBLE_DEVICES': 'GPU-abcdef'}) is None  # UUIDs: not resolvable from sysfs
    assert pipeline.gpu_numa_nodes(str(tmp_path / 'missing'), env={}) is None
    assert pipeline._parse_cpulist('0-2,8,10-11') == [0, 1, 2, 8, 10, 11]


def test_pin_rank_keeps_a_rank_on_its_gpus_numa_node(tmp_path, monkeypatch):
    """VERDICT r4 (next 6): the rank's CPUs are a share of its GPU's NUMA node, not the r-th slice of the host; where the
    topology is unknown the r-th slice remains; a mask somebody else chose is left alone; and the spin-or-poll
    decision after pinning divides the rank's own mask no further."""
    import os
    before = sorted(os.sched_getaffinity(0))
    if len(before) < 4 or len(before) != os.cpu_count():
        pytest.skip('needs >= 4 CPUs and the whole host in the mask')
    half = len(before) // 2
    lo, hi = before[:half], before[half:2 * half]
    fmt = lambda c: ','.join(str(x) for x in c)
    # GPUs 0 and 1 hang off node 1 (the UPPER half of the CPUs), GPUs 2 and 3 off node 0: the opposite of the r-th slice
    _fake_sysfs(str(tmp_path), [1, 1, 0, 0], {0: fmt(lo), 1: fmt(hi)})
    monkeypatch.delenv('HIP_VISIBLE_DEVICES', raising=False)
    monkeypatch.delenv('ROCR_VISIBLE_DEVICES', raising=False)
    monkeypatch.delenv('CUDA_VISIBLE_DEVICES', raising=False)
    try:
        assert pipeline.pin_rank(0, 1, sysfs=str(tmp_path)) is None  # single rank: untouched
        mine = pipeline.pin_rank(1, 4, sysfs=str(tmp_path))           # second GPU of node 1: the second half of `hi`
        share = len(hi) // 2
        assert mine == hi[share:2 * share] and sorted(os.sched_getaffinity(0)) == mine
        # the mask is this rank's own now: the budget is not divided by the local world a second time
        assert pipeline.rank_cpu_budget(4) == min(float(len(mine)), pipeline._quota_cpus() / 4)
        assert pipeline.choose_wait_us(1, local_world=4) == (0 if pipeline.rank_cpu_budget(4) >= 2 else 50)
        # ... and a second call (a launcher that binds, then a library that pins) leaves the chosen mask alone
        assert pipeline.pin_rank(1, 4, sysfs=str(tmp_path)) is None and sorted(os.sched_getaffinity(0)) == mine
        os.sched_setaffinity(0, before)
        mine = pipeline.pin_rank(2, 4, sysfs=str(tmp_path))           # first GPU of node 0
        assert mine == lo[:len(lo) // 2]
        os.sched_setaffinity(0, before)
        # no topology: the r-th contiguous slice of the mask
        mine = pipeline.pin_rank(1, 2, sysfs=str(tmp_path / 'missing'))
        assert mine == before[half:2 * half]
    finally:
        os.sched_setaffinity(0, before)
        pipeline._pinned = False
    assert pipeline.choose_wait_us(4, local_world=10 ** 6) == 50  # no cores to spin on
    assert pipeline.rank_cpu_budget(2) == min(len(before) / 2.0, pipeline._quota_cpus() / 2)  # shared mask: divided


def test_a_held_pair_is_recognised_by_identity_and_version_only():
    """ADVICE r5 (round 6): what a lock-step group remembers of a pair's tensors -- the tensors themselves and their version counters --
    answers `Engine.run` only for the same memory, extent AND version: an in-place refill, another tensor of the same shape, or a
    tensor that merely took over the address do not match.  (Pure host logic: CPU tensors.)"""
    import torch
    from rdmnet_amd.engine import Engine
    r, s = torch.zeros(10, 3), torch.ones(12, 3)
    held = Engine._held(r, s)
    assert Engine._is_held(held, r, s)
    assert Engine._is_held(held, r[:], s.view(12, 3))          # views of the same memory and extent
    assert not Engine._is_held(held, r.clone(), s)             # same values, other memory
    assert not Engine._is_held(held, r[:5], s)                 # same address, other extent
    assert not Engine._is_held(held, s, r)
    r.add_(1.0)                                                # refilled in place: the version counter moved
    assert not Engine._is_held(held, r, s)
    assert Engine._is_held(Engine._held(r, s), r, s)
