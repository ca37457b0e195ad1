"""Several scan pairs in flight on one GPU: the product's scheduler.

The reference's test loop (geotransformer/engine/single_tester.py:86-134) pushes one pair at a time through the
model and synchronises after each; most kernels of one pair are far too small to fill 256 CUs, so a GPU driven
like that idles (docs/EXPERIMENTS.md 5: 240 pairs/s with one pair in flight against 500+ with four).  `PairPipeline` is
what replaces that loop here -- used by `rdmnet_amd.infer.Tester`, by `bench.py`'s timed region and by anything
else that has a stream of pairs:

  * N native engines (`rdm_engine`, one per in-flight pair) that share ONE copy of the prepared weights
    (`rdm_engine_share_params`, reference counted by the library),
  * N host threads, each with its own HIP stream, drawing their next job from ONE shared queue (a stream that
    falls behind takes fewer; all drain together at the end of the input),
  * staggered starts: worker k draws its first job k x `stagger_ms` after worker 0, so the in-flight pairs do
    not walk through the same stages in phase (four serial subsampling kernels, then four encoders contending),
  * optionally (`collate_batch` > 1, round 5) a worker draws several jobs at a time and collates them with ONE sequence of
    launches (`Engine.collate_batch`: the subsampling chains, grid builds and searches of a pair are launch-bound; the pairs of
    a batch share them), then runs the pairs' forwards one by one -- the same bits as the one-by-one schedule, +2-3 % pairs/s
    for 2-4 x the per-pair latency: off by default (never with stage tensors kept: those runs build the reference's full tables),
  * lock step (`lockstep` = B > 1, round 5, the default with two or more streams): a worker owns B engines, draws B jobs and runs
    them as ONE lock-step group on its stream (`Engine.run_lockstep`): the B runs advance together on the worker's thread, the
    same kernel of the B pairs goes out as one grouped launch (B x the workgroups per launch, a quarter of the launches and host
    waits), their collates as one launch sequence.  Every pair keeps the bits of its own run; the B pairs finish together,
  * `rdm_engine_set_pairs_in_flight(N)` (GEMM residency hint from three pairs up; with N = 1 the engine runs in its latency
    mode instead: the wide, independent parts of the pair on a side stream, `Engine.set_overlap`),
  * waits at the engines' size read-backs by spinning (`hipStreamSynchronize`) when the process owns two host
    cores per in-flight pair, by polling with 50 us sleeps otherwise (`cpu_budget`),
  * one hardware queue per stream: the HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES (default 4) hardware
    queues, and four worker streams + the default stream on four queues serialise two streams (345 instead of
    460 pairs/s); `rdmnet_amd/__init__.py` sets GPU_MAX_HW_QUEUES=8 unless the caller chose a value -- it must
    be in the environment before the first HIP call of the process (`hw_queues()` reports what is in effect and
    warns when the default came too late); the launchers (`bench.py`, `python -m rdmnet_amd.infer`) put it into
    the ranks' environment explicitly,
  * one process per GPU: `pin_rank` keeps a rank on the CPUs of its GPU's NUMA node.

Results come back in INPUT order whatever the completion order, and every engine is deterministic, so N pairs in
flight give the bits of a serial run (tests/test_pipeline_gpu.py).
"""
import os
import threading
import time

import numpy as np
import torch

from .engine import Engine

DEFAULT_PAIRS_IN_FLIGHT = 4   # the 5th in-flight pair shares a hardware pipe with another one (docs/EXPERIMENTS.md 5b)
DEFAULT_STAGGER_MS = 1.5
# Pairs a worker collates with one sequence of launches (Engine.collate_batch) before it runs their forwards one by one.  Measured
# (round 5, profiles/r05_collate_batch_ab.txt): 4 per batch +3 % pairs/s on a 20-pair run (505 against 490) and +1.5 % on a long one, 8 per
# batch +2.7 % -- for twice / four times the per-pair latency (13.9 / 32 ms against 7.3), and the KPConv kernels of the timed
# region then share the GPU with more of the other pairs' wide kernels (their event-bracketed durations grow by a fifth).  Default:
# every pair collates itself, the schedule of rounds 1-4; a throughput-only caller sets 4-8.
DEFAULT_COLLATE_BATCH = 1
# Pairs a worker runs as ONE lock-step group on its stream (Engine.run_lockstep, round 5): B engines per worker, their runs advance
# together on the worker's thread and identical kernels of the B pairs go out as one grouped launch (their collates as one launch
# sequence).  Every pair keeps the bits of its own run.  Measured (tools/lockstep_lab.py, docs/EXPERIMENTS.md 5g): 4 workers x 4
# pairs 578 pairs/s against 533 for 4 x 1 -- the B pairs of a group finish together, so a pair's latency is the group's
# (26 ms against 7.4 ms; 8 per group: no more pairs/s, twice the latency).  The default of a pipeline that builds its own engines
# and keeps two or more streams busy (one stream = the caller asks for latency: one pair per call, the engine's latency mode);
# lockstep=1 is the schedule of rounds 1-4.
DEFAULT_LOCKSTEP = 4


def _quota_cpus():
    """The cgroup CPU quota alone (cpu.max / cfs_quota_us; inf when there is none)."""
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        return float('inf') if quota == 'max' else float(quota) / float(period)
    except (OSError, ValueError):
        try:
            q = float(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            per = float(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            return q / per if q > 0 else float('inf')
        except (OSError, ValueError):
            return float('inf')


def cpu_budget():
    """Host CPUs this process may use: affinity mask capped by the cgroup quota."""
    return min(float(len(os.sched_getaffinity(0))), _quota_cpus())


def _parse_cpulist(text):
    """'0-15,32-47' -> [0..15, 32..47]"""
    cpus = []
    for part in text.strip().split(','):
        if not part:
            continue
        lo, _, hi = part.partition('-')
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_numa_nodes(sysfs='/sys', env=None):
    """NUMA node of every GPU this process can see, in HIP device order, from sysfs alone (no HIP call: the affinity must be
    set before the runtime starts its helper threads).  The KFD topology lists the agents in the order the ROCm runtime
    enumerates them (`<sysfs>/class/kfd/kfd/topology/nodes/<i>/properties`: GPUs are the nodes with simd_count > 0;
    `domain` and `location_id` = bus << 8 | devfn give the PCI address, whose `numa_node` file names the node);
    ROCR_VISIBLE_DEVICES then HIP_VISIBLE_DEVICES re-index that list when they are plain index lists.  Returns a list of
    node numbers (-1 = unknown), or None when the topology cannot be read or a visibility variable holds UUIDs."""
    env = os.environ if env is None else env
    root = os.path.join(sysfs, 'class', 'kfd', 'kfd', 'topology', 'nodes')
    try:
        ids = sorted(int(d) for d in os.listdir(root) if d.isdigit())
    except OSError:
        return None
    gpus = []
    for i in ids:
        try:
            props = dict(line.split()[:2] for line in open(os.path.join(root, str(i), 'properties')) if len(line.split()) >= 2)
        except OSError:
            continue  # (nodes of GPUs outside this container's device cgroup are unreadable)
        if int(props.get('simd_count', '0')) <= 0:
            continue
        loc, dom = int(props.get('location_id', '0')), int(props.get('domain', '0'))
        addr = f'{dom:04x}:{(loc >> 8) & 0xff:02x}:{(loc >> 3) & 0x1f:02x}.{loc & 7:x}'
        try:
            node = int(open(os.path.join(sysfs, 'bus', 'pci', 'devices', addr, 'numa_node')).read())
        except (OSError, ValueError):
            node = -1
        gpus.append(node)
    for var in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        val = env.get(var)
        if val is None or val.strip() == '':
            continue
        toks = [t.strip() for t in val.split(',') if t.strip()]
        if not all(t.isdigit() for t in toks):
            return None
        gpus = [gpus[int(t)] for t in toks if int(t) < len(gpus)]
    return gpus or None


_pinned = False  # pin_rank narrowed this process's affinity mask to its own share


def pin_rank(local_rank, local_world, device_index=None, sysfs='/sys', force=False):
    """One process per GPU on a shared host: the rank keeps a share of the CPUs OF ITS GPU'S NUMA NODE (round 4
    handed rank r the r-th contiguous slice of the host, whatever socket the rank's GPU hangs off) -- the ranks whose GPUs
    sit on the same node split that node's CPUs in rank order; its engines' worker threads, the stager's readers and the
    runtime's helper threads then stay next to the GPU they feed (with spinning waits that is 4 busy threads per rank).
    `device_index`: the rank's HIP device (default: local_rank).  Where the topology cannot be read (`gpu_numa_nodes`), or
    the node's CPUs are not in this process's mask, the rank falls back to the r-th contiguous slice of the mask.

    Pins only when the mask is still the whole host's: a launcher that already bound each rank to its own CPU
    set (numactl, torchrun with binding, k8s cpusets) is left alone -- slicing a rank's own mask by local_world again would
    leave it 1/local_world of its share; `force=True` slices whatever mask there is (all ranks must then share it).
    No-op for a single rank, with fewer CPUs than ranks, or where the platform has no affinity call.  Returns the CPU list it
    set (or None).  Call it at the top of the rank's main(), before the first HIP call."""
    global _pinned
    if local_world <= 1 or not hasattr(os, 'sched_setaffinity'):
        return None
    cpus = sorted(os.sched_getaffinity(0))
    if not force and len(cpus) < (os.cpu_count() or len(cpus)):
        return None  # somebody chose this mask already
    per = len(cpus) // local_world
    if per < 1:
        return None
    mine = cpus[local_rank * per:(local_rank + 1) * per]
    nodes = gpu_numa_nodes(sysfs)
    dev = local_rank if device_index is None else device_index
    if nodes is not None and len(nodes) >= local_world and 0 <= dev < len(nodes) and nodes[dev] >= 0:
        try:
            node_cpus = [c for c in _parse_cpulist(open(os.path.join(sysfs, 'devices', 'system', 'node', f'node{nodes[dev]}',
                                                                     'cpulist')).read()) if c in set(cpus)]
        except (OSError, ValueError):
            node_cpus = []
        # the ranks of this host that share the node (rank r drives device r unless the caller says otherwise: then only
        # this rank's own device is known, and it is placed by its rank among the devices of the node)
        peers = [r for r in range(local_world) if r < len(nodes) and nodes[r] == nodes[dev]]
        me = peers.index(dev) if dev in peers else 0
        share = len(node_cpus) // max(len(peers), 1)
        if share >= 1:
            mine = node_cpus[me * share:(me + 1) * share]
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return None
    _pinned = True
    return mine


def rank_cpu_budget(local_world=1):
    """Host CPUs THIS RANK may use.  The cgroup quota is shared by the ranks of the host: always divided.  The affinity mask
    is divided only while the ranks still share it -- after pin_rank (or a launcher's own binding: a mask smaller than the
    host) it is this rank's share already (dividing it again made every multi-rank run poll instead of spin)."""
    local_world = max(int(local_world), 1)
    aff = float(len(os.sched_getaffinity(0))) if hasattr(os, 'sched_getaffinity') else float(os.cpu_count() or 1)
    own_mask = _pinned or aff < float(os.cpu_count() or aff)
    share = aff if own_mask else aff / local_world
    return min(share, _quota_cpus() / local_world)


def hw_queues():
    """GPU_MAX_HW_QUEUES as this process's HIP runtime sees it (None = the runtime's default of 4: a fifth stream then
    shares a queue, module docstring).  The variable only counts if it was in the environment at the first HIP call."""
    v = os.environ.get('GPU_MAX_HW_QUEUES')
    return int(v) if v and v.isdigit() else None


def choose_wait_us(pairs_in_flight, local_world=1):
    """0 = spin at the read-backs (needs ~2 host cores per in-flight pair of this rank), else poll with 50 us sleeps."""
    return 0 if rank_cpu_budget(local_world) >= 2 * pairs_in_flight else 50


class PairResult:
    """What a pair leaves on the host: pose, correspondences (numpy copies of the engine's pinned buffer) and counters."""
    __slots__ = ('transform', 'ref_corr_points', 'src_corr_points', 'corr_scores', 'n_correspondences', 'n_ref_nodes',
                 'n_src_nodes', 'n_node_correspondences', 'level_sizes', 'ms')

    def __init__(self, eng, res, ms):
        self.transform = eng.transform()
        self.ref_corr_points, self.src_corr_points, self.corr_scores = eng.host_corr()
        self.n_correspondences = int(res.n_correspondences)
        self.n_ref_nodes, self.n_src_nodes = int(res.n_ref_nodes), int(res.n_src_nodes)
        self.n_node_correspondences = int(res.n_node_correspondences)
        self.level_sizes = [int(x) for x in res.level_sizes]
        self.ms = ms


class PairPipeline:
    """`pairs_in_flight` engines / host threads / HIP streams on one device; see the module docstring.

    map(jobs, fn) / imap(jobs, fn): fn(engine, job) runs on a worker thread inside `torch.cuda.stream(<its stream>)`
    and must take what it needs from the engine before returning (the next job reuses the engine's arena).  Jobs are
    drawn lazily from the iterable under a lock, IN the worker's stream context -- a `dataset.PairStager` works as the
    job source (its hand-off event is waited for by the drawing worker's stream)."""

    def __init__(self, cfg, state, device=None, pairs_in_flight=DEFAULT_PAIRS_IN_FLIGHT, wait_us=None,
                 stagger_ms=DEFAULT_STAGGER_MS, keep_taps=False, local_world=None, engines=None, streams=None,
                 collate_batch=DEFAULT_COLLATE_BATCH, lockstep=None, arena_bytes=None):
        self._gpu = torch.cuda.is_available()
        if not self._gpu and engines is None:
            raise RuntimeError('rdmnet_amd.pipeline needs a GPU (no CPU fallback)')
        # (without a GPU only the scheduler itself can run, on engines the caller injected: tests/test_pipeline.py)
        self.device = (torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)) if self._gpu else None
        self.n = max(1, int(pairs_in_flight))
        self.collate_batch = max(1, int(collate_batch))
        if lockstep is None:  # (injected engines: the caller's set is what runs)
            lockstep = DEFAULT_LOCKSTEP if (engines is None and self.n >= 2) else 1
        self.lockstep = max(1, min(8, int(lockstep)))  # (8 = what one grouped launch carries, lockstep.h)
        self.keep_taps = bool(keep_taps)
        self.stagger_s = max(0.0, float(stagger_ms)) * 1e-3
        local_world = int(os.environ.get('LOCAL_WORLD_SIZE', '1')) if local_world is None else local_world
        self.wait_us = choose_wait_us(self.n, local_world) if wait_us is None or wait_us < 0 else int(wait_us)
        if streams is not None:
            self.streams = list(streams)
        else:  # (also for a single pair in flight: the engine's latency mode does not work on the null stream)
            self.streams = [torch.cuda.Stream(device=self.device) for _ in range(self.n)] if self._gpu else [None] * self.n
        # Engines.  Worker k owns groups[k]: `lockstep` engines sharing ONE copy of the weights.  A pipeline that builds its own engines
        # creates ONE per worker here and the other lockstep - 1 of every group on the first draw that runs in lock step
        # (`ensure_groups`; ADVICE r5: callers that never pass `tensors_of` / `group_fn` -- one job per engine call -- do not pay for
        # 4 x the arenas).  HBM: an engine's activation arena defaults to 3 GiB (growable: a pair that exhausts it is re-run on a
        # doubled one), i.e. 48 GiB of the 288 at the default 4 x 4 once the groups exist, whether or not a pair ever needs it
        # (measured, tools/dbg/arena_probe.py: a 2 x 16 k-point pair bumps through 0.85-1.2 GiB).  `arena_bytes` re-allocates every
        # engine's arena at that size instead (still growable): what several ranks on one device, or a GPU with less memory, pass.
        self.arena_bytes = None if arena_bytes is None else int(arena_bytes)
        self._cfg_state = (cfg, state)
        self._groups_lock = threading.Lock()
        given = list(engines) if engines is not None else []
        want = self.n * self.lockstep
        if given and len(given) >= want:  # injected: the caller's set is what runs
            self.groups = [given[k * self.lockstep:(k + 1) * self.lockstep] for k in range(self.n)]
            spare = given[want:]
        else:
            if not self._gpu:
                raise RuntimeError(f'rdmnet_amd.pipeline: {want} injected engines needed ({self.n} workers x {self.lockstep} in lock step)')
            with torch.cuda.device(self.device):
                while len(given) < self.n:
                    given.append(Engine(cfg, state, device=self.device, share_with=given[0] if given else None))
            self.groups = [[given[k]] for k in range(self.n)]
            spare = given[self.n:]  # (injected engines beyond one per worker join the groups first, ensure_groups)
        self._spare = spare
        self.engines = [grp[0] for grp in self.groups] + spare  # (engines[k]: worker k's engine, as without lock step)
        for eng in [e for grp in self.groups for e in grp] + spare:
            self._configure(eng)
        self.cfg = cfg
        self.last_stats = None

    def _configure(self, eng):
        if self.arena_bytes is not None and hasattr(eng, 'reserve'):
            eng.reserve(self.arena_bytes)
        eng.set_wait(self.wait_us)
        eng.set_pairs_in_flight(self.n * self.lockstep)
        eng.keep_taps(self.keep_taps)

    def ensure_groups(self, k=None):
        """Completes worker k's lock-step group (every group with k = None) to `lockstep` engines; no-op once they exist.
        Called by the workers on their first lock-step draw, and by callers that address `groups` before running."""
        for g in (range(self.n) if k is None else [k]):
            if len(self.groups[g]) >= self.lockstep:
                continue
            with self._groups_lock:
                cfg, state = self._cfg_state
                with torch.cuda.device(self.device):
                    while len(self.groups[g]) < self.lockstep:
                        if self._spare:
                            eng = self._spare.pop(0)
                        else:
                            eng = Engine(cfg, state, device=self.device, share_with=self.groups[0][0])
                            self._configure(eng)
                        self.groups[g].append(eng)
        return self.groups

    # ------------------------------------------------------------------ generic scheduler
    def imap(self, jobs, fn, stagger=True, window=None, tensors_of=None, prepare=None, group_fn=None):
        """Yields fn(engine, job) for every job, in job order.  `window` bounds how far completion may run ahead of
        the consumer (default 4 x pairs_in_flight results held).

        tensors_of(job) -> (ref, src): the CUDA tensors fn will hand to `engine.run` for this job.  When given (and the
        pipeline's `collate_batch` > 1, no stage tensors kept), a worker draws several jobs at a time, collates them with ONE
        sequence of launches (`Engine.collate_batch`: the collate is the launch-bound part of a pair) and then runs fn on each --
        `engine.run` recognises the prepared pair and runs its forward alone; the results are the bits of the one-by-one
        schedule.  At the end of a sized job list the remainder is split evenly over the workers so that they still finish together.

        With `lockstep` > 1 (and tensors_of given) the drawn jobs run as ONE lock-step group on the worker's engines
        (`Engine.run_lockstep`) before fn is called on each (engine, job) -- `engine.run` then returns the result that is
        already in place (engines that keep their stage tensors collate their own pair inside the group and hold its tensors
        afterwards: round 6).  prepare(engine, job, i, n), if given, is called for job i of the n drawn before the group runs
        (per-run engine settings, e.g. the layer profile).

        group_fn(engines, jobs) -> [result per job], if given (instead of tensors_of; fn is then unused and may be None): the worker
        hands the jobs it drew (up to `lockstep`) and as many of its engines to ONE call -- for callers that run the group
        themselves (e.g. `Engine.forward_lockstep` / `model([data_dict, ...])` on data_dicts they collated)."""
        n = self.n
        window = max(n, 4 * n if window is None else int(window))
        lockstep = self.lockstep > 1 and (tensors_of is not None or group_fn is not None)
        bmax = self.lockstep if lockstep else (self.collate_batch if (tensors_of is not None and not self.keep_taps) else 1)
        window = max(window, 2 * n * bmax)
        total = len(jobs) if hasattr(jobs, '__len__') else None
        it = enumerate(iter(jobs))
        draw_lock = threading.Lock()
        cv = threading.Condition()
        done = {}
        state = {'next_out': 0, 'drawn': 0, 'exhausted': False, 'error': None, 'stop': False, 'alive': n, 'tail': None}
        # where the workers' time went (seconds, summed over workers): waiting for / drawing the next job (the job source's
        # cost: reading and staging a pair), running fn, waiting for the consumer's window -- `last_stats` after the run
        # `latency_ms`: per job (by slot), from the moment its worker had drawn it -- with its batch -- to its result: the batch's
        # collate and the forwards of the pairs before it in the batch included
        stats = self.last_stats = {'draw_s': 0.0, 'work_s': 0.0, 'window_s': 0.0, 'jobs': 0, 'wall_s': 0.0, 'workers': n,
                                   'collate_batches': 0, 'lockstep_groups': 0, 'group_sizes': {}, 'latency_ms': {}}
        t_begin = time.perf_counter()

        def draw():
            """-> up to bmax (slot, job) pairs, or None at the end of the input."""
            with draw_lock:
                if state['exhausted'] or state['stop']:
                    return None
                want = bmax
                if total is not None and bmax > 1:
                    # The last round: once fewer than n full groups are left, the remainder is split ONCE into n nearly equal
                    # draws (16 left, 4 workers: 4 4 4 4; 10 left: 3 3 2 2), so that the workers still finish together.  (Round 5
                    # re-evaluated "what is left / n" at every draw: 16 left went out as 4 3 3 2 1 1 1 1 -- eight groups, half of
                    # them single pairs -- VERDICT r5, weak 5.  Measured on the 80-pair run: full last groups 621-625 pairs/s, the same
                    # shares as two half-size draws each 606-617, round 5's taper 605-619: docs/EXPERIMENTS.md 5h.)
                    left = total - state['drawn']
                    if state['tail'] is None and left <= n * bmax:
                        base, extra = divmod(left, n)
                        state['tail'] = [base + (1 if k < extra else 0) for k in range(n) if base + (1 if k < extra else 0) > 0]
                    if state['tail'] is not None:
                        want = state['tail'].pop(0) if state['tail'] else 1
                got = []
                while len(got) < want:
                    try:
                        slot, job = next(it)
                    except StopIteration:
                        state['exhausted'] = True
                        break
                    state['drawn'] = slot + 1
                    got.append((slot, job))
                return got or None

        def worker(k):
            stream = self.streams[k]
            import contextlib
            # this thread's CPU-side torch ops (small copies, metrics): no 128-thread OpenMP teams.  (torch keeps the intra-op
            # width per calling thread on OpenMP builds; the consumer restores the process's value after the run)
            torch.set_num_threads(1)
            try:
                with (torch.cuda.device(self.device) if self._gpu else contextlib.nullcontext()):
                    ctx = torch.cuda.stream(stream) if stream is not None else None
                    if ctx is not None:
                        ctx.__enter__()
                    try:
                        if stagger and k > 0 and self.stagger_s > 0:
                            time.sleep(k * self.stagger_s)
                        while True:
                            t0 = time.perf_counter()
                            with cv:  # do not run further ahead of the consumer than `window` results
                                while state['drawn'] - state['next_out'] >= window and not state['stop']:
                                    cv.wait(timeout=0.05)
                            t1 = time.perf_counter()
                            got = draw()
                            t2 = time.perf_counter()
                            if got is None:
                                return
                            if lockstep and len(self.groups[k]) < self.lockstep:
                                self.ensure_groups(k)
                            if lockstep and prepare is not None:
                                for i, (_, job) in enumerate(got):
                                    prepare(self.groups[k][i], job, i, len(got))
                            outs = None
                            if group_fn is not None:  # the caller runs the group itself
                                outs = group_fn(self.groups[k][:len(got)] if lockstep else [self.engines[k]], [job for _, job in got])
                                if len(outs) != len(got):
                                    raise RuntimeError(f'group_fn returned {len(outs)} results for {len(got)} jobs')
                                with cv:
                                    stats['lockstep_groups'] += 1 if len(got) > 1 else 0
                                    stats['group_sizes'][len(got)] = stats['group_sizes'].get(len(got), 0) + 1
                            elif lockstep and len(got) > 1:  # the pairs as one lock-step group; fn's engine.run finds each result in place
                                type(self.groups[k][0]).run_lockstep(self.groups[k], [tensors_of(job) for _, job in got])
                                with cv:
                                    stats['lockstep_groups'] += 1
                                    stats['group_sizes'][len(got)] = stats['group_sizes'].get(len(got), 0) + 1
                            elif len(got) > 1:  # one collate for all of them; fn's engine.run then finds each pair prepared
                                self.engines[k].collate_batch([tensors_of(job) for _, job in got])
                                with cv:
                                    stats['collate_batches'] += 1
                            elif lockstep:
                                with cv:
                                    stats['group_sizes'][1] = stats['group_sizes'].get(1, 0) + 1
                            try:
                                for i, (slot, job) in enumerate(got):
                                    out = outs[i] if outs is not None else fn(self.groups[k][i] if lockstep else self.engines[k], job)
                                    t_done = time.perf_counter()
                                    with cv:
                                        done[slot] = out
                                        stats['jobs'] += 1
                                        stats['latency_ms'][slot] = (t_done - t2) * 1e3
                                        cv.notify_all()
                            finally:  # a result nobody picked up (fn raised, or did not call engine.run) must not answer a later run
                                for eng in (self.groups[k] if k < len(self.groups) else []):
                                    if hasattr(eng, 'clear_pending'):
                                        eng.clear_pending()
                            t3 = time.perf_counter()
                            with cv:
                                stats['window_s'] += t1 - t0
                                stats['draw_s'] += t2 - t1
                                stats['work_s'] += t3 - t2
                    finally:
                        if ctx is not None:
                            ctx.__exit__(None, None, None)
            except BaseException as exc:  # surfaced by the consumer (a worker thread must not fail silently)
                with cv:
                    if state['error'] is None:
                        state['error'] = exc
                    state['stop'] = True
            finally:
                with cv:
                    state['alive'] -= 1
                    cv.notify_all()

        threads = [threading.Thread(target=worker, args=(k,), daemon=True) for k in range(n)]
        torch_threads = torch.get_num_threads()
        for t in threads:
            t.start()
        try:
            while True:
                with cv:
                    while True:
                        if state['next_out'] in done:  # (results completed in order are delivered before an error is raised:
                            break                      # the consumer's output then ends at the first missing pair)
                        if state['error'] is not None:
                            raise state['error']
                        if state['alive'] == 0:  # every worker has left: nothing more will arrive
                            return
                        cv.wait(timeout=0.05)
                    out = done.pop(state['next_out'])
                    state['next_out'] += 1
                    cv.notify_all()
                yield out
        finally:
            with cv:
                state['stop'] = True
                cv.notify_all()
            for t in threads:
                t.join()
            torch.set_num_threads(torch_threads)
            stats['wall_s'] = time.perf_counter() - t_begin

    def map(self, jobs, fn, stagger=True, tensors_of=None, prepare=None, group_fn=None):
        return list(self.imap(jobs, fn, stagger=stagger, window=1 << 30, tensors_of=tensors_of, prepare=prepare, group_fn=group_fn))

    # ------------------------------------------------------------------ the common case
    def run_pairs(self, pairs, stagger=True):
        """pairs: iterable of (ref_points, src_points) float32 CUDA tensors [n,3] on this device (or of
        (item, ref, src) triples as a PairStager yields them).  Returns the PairResults in input order."""
        def one(eng, job):
            ref, src = job[-2], job[-1]
            t0 = time.perf_counter()
            res = eng.run(ref.contiguous(), src.contiguous())
            return PairResult(eng, res, (time.perf_counter() - t0) * 1e3)
        return self.map(pairs, one, stagger=stagger, tensors_of=lambda job: (job[-2].contiguous(), job[-1].contiguous()))

    def close(self):
        """Drops the engines (their arenas; the shared weights go with the last one)."""
        self.engines = []
        self.groups = []
        self._spare = []
