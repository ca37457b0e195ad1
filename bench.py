#!/usr/bin/env python
"""Benchmark of the RDMNet dense-matching inference path on MI355X.

A step = one scan pair through the WHOLE path: GPU collate (4 grid subsamplings + 13 radius
searches; 12 in the one-call engine, which does not build the up-sampling table of level 0 that
nothing reads -- the drop-in API pass builds all 13) + RDMNet.forward (KPConv encoder/decoder, 3DRoFormer x2, vote, NMS, grouping, coarse
matching, Sinkhorn, LGR) -> 4x4 pose on the device.  Inputs (synthetic KITTI-shaped pairs,
~16 k points per scan) are resident in HBM before the timed region; weights are the seeded
synthetic state dict (the reference ships no trained weights).  Metric: scan pairs per second,
whole job.  One process per GPU; pairs are sharded rank-strided with no data-path collective,
one all_gather of result records at the end (RCCL).

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import collections
import ctypes
import json
import os
import sys
import time

os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')  # (rdmnet_amd/__init__.py sets the same default: one hardware queue per in-flight pair)

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md)
MFMA_F32_PEAK_TF = 157.3  # dense fp32 MFMA peak (v_mfma_f32_32x32x2_f32, MI355X_MICROARCH.md)


def make_pairs(n_pairs, cache_dir):
    """Seeded synthetic pairs (rdmnet_amd.synthetic.make_pair).  Pairs 0 and 1 are also stored as a
    fixture (tests/golden/synthetic_pairs.npz, the generator's exact output) to skip ~10 s of host
    ray casting per pair; further ids are generated and cached under gpurun_out/."""
    from rdmnet_amd import synthetic
    return synthetic.cached_pairs(n_pairs, cache_dir, os.path.join(ROOT, 'tests', 'golden', 'synthetic_pairs.npz'))


def _cpu_baseline_worker(cache_dir, n_pairs, threads, budget_s):
    """Runs in a child process (so a pathological host cannot stall the bench): the CPU port (oracle restatement) of
    the same path on a bounded sample of the bench's pairs -- SURVEY.md §8d / BASELINE.md §3: all host cores, median
    of up to 5 pairs after one warm-up pair, collate and forward timed separately; then one pair on a single thread."""
    from oracle import forward as ofw
    from oracle import native
    from rdmnet_amd import config, weights
    cfg = config.make_cfg()
    W = ofw.to_torch(weights.synthetic_state_dict(cfg, seed=0))
    impl = native.reference() or native.restatement()
    pairs = make_pairs(min(n_pairs, 6), cache_dir)

    def one(ref, src):
        t0 = time.perf_counter()
        data = ofw.pyramid(np.concatenate([ref, src]), np.array([len(ref), len(src)], np.int64), cfg, impl=impl)
        t1 = time.perf_counter()
        ofw.forward(W, cfg, data, impl=impl)
        return t1 - t0, time.perf_counter() - t1

    torch.set_num_threads(threads)
    t_start = time.perf_counter()
    one(*pairs[0][:2])  # warm-up (thread pools, allocator)
    pre, fwd = [], []
    for k in range(5):
        a, b = one(*pairs[(k + 1) % len(pairs)][:2])
        pre.append(a)
        fwd.append(b)
        if time.perf_counter() - t_start > budget_s:
            break
    med = float(np.median(np.asarray(pre) + np.asarray(fwd)))
    single = None
    if time.perf_counter() - t_start < 1.5 * budget_s:
        torch.set_num_threads(1)
        a, b = one(*pairs[1 % len(pairs)][:2])
        single = {'value': 1.0 / (a + b), 'unit': 'pairs/s', 'cores': 1, 'collate_s': a, 'forward_s': b, 'sample': '1 pair, 1 run'}
    kind_native = 'reference C++ (oracle/_ref)' if native.reference() is not None else 'restatement C++'
    print('CPU_BASELINE ' + json.dumps({
        'value': 1.0 / med, 'unit': 'pairs/s', 'cores': threads, 'kind': 'port',
        'sample': f'median of {len(pre)} pair(s) of the bench workload after 1 warm-up pair; collate = {kind_native}, '
                  f'single-thread kd-tree/hash map (median {float(np.median(pre)):.2f} s/pair); forward = oracle/forward.py, '
                  f'torch fp32 on {threads} threads (median {float(np.median(fwd)):.2f} s/pair)',
        'single_thread': single}))


def cpu_baseline(cache_dir, n_pairs, timeout_s=150):
    import subprocess
    threads = max(1, min(os.cpu_count() or 1, 16))  # more threads make torch's small CPU ops slower, not faster
    code = (f'import sys; sys.path.insert(0, {ROOT!r}); import bench; '
            f'bench._cpu_baseline_worker({cache_dir!r}, {n_pairs}, {threads}, 20.0)')
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES='')
    try:
        p = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=timeout_s, env=env)
        for line in p.stdout.splitlines():
            if line.startswith('CPU_BASELINE '):
                return json.loads(line[len('CPU_BASELINE '):])
        return {'value': None, 'unit': 'pairs/s', 'cores': threads, 'kind': 'port', 'sample': 'failed: ' + p.stderr[-300:]}
    except subprocess.TimeoutExpired:
        return {'value': None, 'unit': 'pairs/s', 'cores': threads, 'kind': 'port',
                'sample': f'did not finish within {timeout_s} s'}


def _no_nan(x):
    """Strict JSON: NaN/inf -> null."""
    if isinstance(x, dict):
        return {k: _no_nan(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_no_nan(v) for v in x]
    if isinstance(x, float) and not np.isfinite(x):
        return None
    return x


def pose_error(T_est, T_gt):
    R = T_gt[:3, :3].T @ T_est[:3, :3].astype(np.float64)
    ang = 2.0 * np.arcsin(min(1.0, np.linalg.norm(R - np.eye(3)) / (2.0 * np.sqrt(2.0))))
    return float(np.degrees(ang)), float(np.linalg.norm(T_gt[:3, 3] - T_est[:3, 3]))


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: one child process per GPU (the reference's own multi-GPU entry
    self-spawns too, experiments/test_batchoffline.py:255-264 with rank setup engine/base_tester.py:36-40,70-76).  Each
    child is this script with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, exactly what torch.distributed.run would
    set; rank 0's JSON line goes to the inherited stdout.  Returns the exit code (first failing rank's, else 0)."""
    import socket
    import subprocess
    with socket.socket() as sock:  # a free rendezvous port
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   GPU_MAX_HW_QUEUES=os.environ.get('GPU_MAX_HW_QUEUES', '8'),  # explicit in every rank's environment (one queue per in-flight pair)
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))  # dmabuf IPC: RCCL needs it on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in pending:  # a dead rank would leave the others waiting at the next collective
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=480)
    ap.add_argument('--warmup', type=int, default=16)
    ap.add_argument('--pairs', type=int, default=8, help='distinct synthetic pairs cycled through')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--streams', type=int, default=4, help='pairs in flight per GPU (host threads, one HIP stream each)')
    ap.add_argument('--path', choices=['engine', 'python'], default='engine',
                    help='engine: one native call per pair (rdm_engine_run); python: per-op mirror (rdmnet_amd.model)')
    ap.add_argument('--dist-backend', default='nccl', help='nccl (= RCCL) in production; gloo only to exercise the\n'
                    'multi-process logic on a single GPU (with RDM_BENCH_SHARE_DEVICE=1)')
    ap.add_argument('--force-dist', action='store_true',
                    help='with --gpus 1: initialise the process group anyway (world size 1, backend --dist-backend) and send the\n'
                         'record gather, the barriers and the timing reduction through it -- the RCCL path on a single GPU')
    ap.add_argument('--layer-events-every', type=int, default=8,
                    help='record the per-KPConv-layer HIP events (roofline) on every N-th pair of a stream: 42 event\n'
                         'records per pair cost ~13 %% of the throughput with 4 pairs in flight; 0 = never')
    ap.add_argument('--wait-us', type=int, default=-1,
                    help='how an engine waits at its size read-backs: 0 = hipStreamSynchronize (spins a core per pair in\n'
                         'flight), N > 0 = poll and sleep N us; -1 = choose from the CPU budget of this rank')
    ap.add_argument('--stagger-ms', type=float, default=1.5,
                    help='in-flight pair k starts k x this many ms after the first one (de-phases the pairs; 0 = all at once)')
    ap.add_argument('--ramp-seconds', type=float, default=5.0,
                    help='untimed pairs run for this long before the warm-up steps so that host and GPU clocks are at their\n'
                         'steady state (a fresh box is 15-20 %% slower for its first seconds); 0 = none')
    ap.add_argument('--full-steps', type=int, default=96,
                    help='pairs of the all-13-tables pass of the engine path after the timed region (0 = skip)')
    ap.add_argument('--host-steps', type=int, default=96,
                    help='pairs of the host-to-host pass that follows the timed region (0 = skip)')
    ap.add_argument('--api-collate', choices=['native', 'python'], default='native',
                    help='collate of the drop-in-API pass: native = rdm_engine_collate (one call), python = 17 launches from Python')
    ap.add_argument('--api-steps', type=int, default=192,
                    help='pairs of the drop-in-API pass (Python collate + model(data_dict)) after the timed region (0 = skip)')
    ap.add_argument('--real-slots', choices=['on', 'off'], default='on',
                    help='on: count the real (non-padding) neighbour slots of every distinct pair before the run (8 untimed\n'
                         'collates) for the real-slot variant of the roofline; off: profile runs that should contain bench pairs only')
    ap.add_argument('--pin', choices=['on', 'off'], default='on',
                    help='with several ranks on one host: pin rank r to the r-th contiguous slice of the CPUs this job may use')
    ap.add_argument('--cache', default=os.path.join(ROOT, 'gpurun_out', 'bench_pairs'))
    ap.add_argument('--collate-batch', type=int, default=None,
                    help='pairs a worker collates with ONE sequence of launches before it runs their forwards one by one\n'
                         '(rdm_engine_collate_batch; default: rdmnet_amd.pipeline.DEFAULT_COLLATE_BATCH; 1 = every pair collates itself,\n'
                         'the schedule of rounds 1-4).  Same results bit for bit.')
    ap.add_argument('--lockstep', type=int, default=None,
                    help='pairs a worker runs as ONE lock-step group on its stream (rdm_engine_run_lockstep: identical kernels of the\n'
                         'pairs of a group go out as one grouped launch, their collates as one launch sequence; default:\n'
                         'rdmnet_amd.pipeline.DEFAULT_LOCKSTEP; 1 = one pair per engine call, the schedule of rounds 1-4).  A step is then\n'
                         'one group: --steps K times this many pairs.  Same results bit for bit.')
    ap.add_argument('--arena-mb', type=int, default=0,
                    help='activation arena of every engine in MiB (growable; 0 = the library default of 3 GiB per engine, 48 GiB for the\n'
                         'default 4 streams x 4 engines): what several ranks sharing one device pass')
    ap.add_argument('--dry-run', action='store_true',
                    help='construct the communicator (RCCL for --dist-backend nccl), run the pre-flight -- the barrier, the ragged\n'
                         'record gather and the timing reduction of a real run on dummy records -- print one JSON line and exit\n'
                         'WITHOUT running a pair: everything a multi-GPU launch does besides the per-GPU work (needs no GPU with\n'
                         '--dist-backend gloo)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args.gpus))  # plain `python bench.py --gpus N`: this process becomes the launcher
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but the launcher set WORLD_SIZE={world}')
    # one contiguous slice of the host's CPUs per rank (self-spawned ranks and torch.distributed.run launches alike): with
    # spinning waits a rank keeps `--streams` host threads busy, which should not migrate over the sockets of a NUMA host
    local_world = int(os.environ.get('LOCAL_WORLD_SIZE', world))
    from rdmnet_amd import pipeline
    pinned_cpus = pipeline.pin_rank(local_rank, local_world) if args.pin == 'on' else None
    if os.environ.get('RDM_BENCH_SHARE_DEVICE') == '1':
        local_rank = 0  # test hook: all ranks on one GPU
    have_gpu = torch.cuda.is_available()
    if not have_gpu and not (args.dry_run and args.dist_backend != 'nccl'):
        raise SystemExit('bench.py needs a GPU (there is no CPU fallback; --dry-run --dist-backend gloo exercises the launcher alone)')
    if have_gpu:
        torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank) if have_gpu else torch.device('cpu')
    dist = None
    force = args.force_dist and world == 1
    if world > 1 or force:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if 'MASTER_PORT' not in os.environ:  # --force-dist without a launcher: a free rendezvous port
            import socket
            with socket.socket() as sock:
                sock.bind(('127.0.0.1', 0))
                os.environ['MASTER_PORT'] = str(sock.getsockname()[1])
        if args.dist_backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world)

    from rdmnet_amd import collate, config, engine, model, sharding, weights

    # Pre-flight of the multi-rank half: the same calls a run ends with -- barrier, ragged gather of result
    # records, max-reduction of the elapsed time + gather of the latencies -- on dummy records whose content is checked, BEFORE
    # any pair runs: a broken communicator fails here, in the first second, and `--dry-run` stops after it.
    comm_dev0 = dev if args.dist_backend == 'nccl' else torch.device('cpu')
    preflight = None
    if dist is not None:
        tp = time.perf_counter()
        dist.barrier()
        dummy = torch.full((rank % 3 + 1, 5), float(rank), dtype=torch.float32)  # ragged: 1-3 rows per rank
        got = sharding.gather_records(dummy.to(comm_dev0), world, dist, force)
        assert len(got) == world and all(g.shape == (r % 3 + 1, 5) and bool((g.cpu() == float(r)).all()) for r, g in enumerate(got)), \
            'pre-flight gather returned wrong records'
        tmax, lats = sharding.reduce_timing(0.001 * (rank + 1), [float(rank)] * (rank % 2 + 1), world, dist, comm_dev0, force)
        assert abs(tmax - 0.001 * world) < 1e-9 and lats == [float(r) for r in range(world) for _ in range(r % 2 + 1)], (tmax, lats)
        dist.barrier()
        preflight = {'ok': True, 'ms': (time.perf_counter() - tp) * 1e3, 'ops': 'barrier, all_gather (counts + ragged records), '
                     'all_reduce(MAX), all_gather (latencies), barrier'}
    if args.dry_run:
        if rank == 0:
            print(json.dumps({'dry_run': True, 'n_gpus': world, 'backend': args.dist_backend if dist is not None else None,
                              'library': (('RCCL ' + '.'.join(str(v) for v in torch.cuda.nccl.version())) if args.dist_backend == 'nccl' and dist is not None
                                          else ('gloo' if dist is not None else None)),
                              'preflight': preflight, 'gpu_max_hw_queues': pipeline.hw_queues(),
                              'host_cpus_per_rank': pipeline.rank_cpu_budget(local_world),
                              'host_cpus_pinned': len(pinned_cpus) if pinned_cpus else None,
                              'gpu_numa_nodes': pipeline.gpu_numa_nodes()}))
        if dist is not None:
            dist.destroy_process_group()
        return
    cfg = config.make_cfg()
    state = weights.synthetic_state_dict(cfg, seed=0)
    net = None
    if args.path == 'python':
        net = model.create_model(cfg).cuda(local_rank)
        net.load_state_dict(state)

    if rank == 0:
        pairs = make_pairs(args.pairs, args.cache)
    if dist is not None:
        dist.barrier()
    if rank != 0:
        pairs = make_pairs(args.pairs, args.cache)
    # inputs resident in HBM before the timed region
    dev_pairs = [(torch.from_numpy(r).to(dev), torch.from_numpy(s).to(dev)) for r, s, _ in pairs]
    n_points = float(np.mean([len(r) + len(s) for r, s, _ in pairs]))

    def pid_of(i):
        """The synthetic pair of this rank's local step i.  The pair STREAM is sharded rank-strided (global step rank + i * world,
        recorded with every result); which of the `--pairs` distinct clouds a step uses must not depend on `world` alone --
        (rank + i * world) % 8 hands an 8-rank run the same cloud at every step of a rank, i.e. lock-step groups of four identical
        pairs, the best case for grouping.  With one rank this is i % len, as in every earlier round."""
        return (i + 3 * rank) % len(dev_pairs)

    def step(i):  # per-op Python mirror
        r, s = dev_pairs[pid_of(i)]
        item = {'ref_points': r, 'src_points': s, 'ref_feats': torch.ones((r.shape[0], 1), device=dev),
                'src_feats': torch.ones((s.shape[0], 1), device=dev)}
        data = collate.registration_collate_fn_stack_mode([item], cfg.backbone.num_stages, cfg.backbone.init_voxel_size,
                                                          cfg.backbone.init_radius, cfg.neighbor_limits, device=dev)
        data['testing'] = True
        out = net(data, {})  # a taps dictionary selects the per-op mirror (without one forward() is the native call)
        return out['estimated_transform'].cpu().numpy(), out['corr_scores'].shape[0]

    def fence():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # (RDM_BENCH_HIGH_PRIORITY_STREAMS=k, developer experiment: the first k streams are created with high priority)
    n_high = int(os.environ.get('RDM_BENCH_HIGH_PRIORITY_STREAMS', '0'))
    custom_streams = None
    if n_high > 0 and args.streams > 1:
        custom_streams = [torch.cuda.Stream(device=dev, priority=-1 if k < n_high else 0) for k in range(args.streams)]
    # (RDM_BENCH_CU_MASK=interleave|blocks, developer experiment: every stream gets its own 1/streams of the CUs through
    # hipExtStreamCreateWithCUMask -- spatial partitioning instead of contention for the same CUs)
    cu_mode = os.environ.get('RDM_BENCH_CU_MASK')
    if cu_mode and args.streams > 1:
        import ctypes
        hip = ctypes.CDLL('libamdhip64.so')
        n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
        custom_streams = []
        for k in range(args.streams):
            bits = [0] * ((n_cu + 31) // 32)
            for cu in range(n_cu):
                mine = (cu % args.streams == k) if cu_mode == 'interleave' else (cu * args.streams // n_cu == k)
                if mine:
                    bits[cu // 32] |= 1 << (cu % 32)
            arr = (ctypes.c_uint32 * len(bits))(*bits)
            h = ctypes.c_void_p()
            rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), len(bits), arr)
            assert rc == 0, rc
            custom_streams.append(torch.cuda.ExternalStream(h.value, device=dev))

    # The scheduler is the PRODUCT's (rdmnet_amd.pipeline.PairPipeline, the one `python -m rdmnet_amd.infer` runs on): N engines
    # sharing one copy of the weights, N host threads / HIP streams drawing steps from one queue, staggered starts, the
    # pairs-in-flight hint, and spin-or-poll waits chosen from this rank's CPU budget.
    budget = pipeline.rank_cpu_budget(local_world)
    collate_batch = pipeline.DEFAULT_COLLATE_BATCH if args.collate_batch is None else max(1, args.collate_batch)
    # (the pipeline's own default: lock-step groups with two or more streams; one stream = latency: one pair per call)
    lockstep = (pipeline.DEFAULT_LOCKSTEP if args.streams >= 2 else 1) if args.lockstep is None else max(1, min(8, args.lockstep))
    if args.path != 'engine':
        collate_batch = lockstep = 1
    pps = lockstep  # pairs per step: one step = one pass of the hot path over one batch = one engine call (a lock-step group)
    n_timed, n_warm = args.steps * pps, args.warmup * pps
    pipe = pipeline.PairPipeline(cfg, state, device=dev, pairs_in_flight=args.streams, wait_us=args.wait_us,
                                 stagger_ms=args.stagger_ms, local_world=local_world, streams=custom_streams,
                                 collate_batch=collate_batch, lockstep=lockstep, arena_bytes=(args.arena_mb << 20) if args.arena_mb > 0 else None)
    pipe.ensure_groups()  # (the pipeline completes its lock-step groups on their first draw; the layer profile addresses them before)
    wait_us, engines, streams = pipe.wait_us, pipe.engines, pipe.streams
    worker_of = {id(e): k for k, grp in enumerate(pipe.groups) for e in grp}
    for eng in engines:
        eng.enable_profile(False)

    # Slots of every KPConv layer's neighbour table that hold a real neighbour (the rest is padding behind them), per
    # distinct pair: the `real_slots` variant of the roofline.  Untimed; layer order = Encoder.forward (backbone.py:72-107).
    real_slots = {}
    if args.real_slots == 'on':
        for pid, (r_, s_) in enumerate(dev_pairs):
            dd = engines[0].collate(r_, s_)
            nb = [int((dd['neighbors'][l] < dd['points'][l].shape[0]).sum()) for l in range(5)]
            sub = [int((dd['subsampling'][l] < dd['points'][l].shape[0]).sum()) for l in range(4)]
            real_slots[pid] = [nb[0], nb[0], sub[0], nb[1], nb[1], sub[1], nb[2], nb[2], sub[2], nb[3], nb[3], sub[3], nb[4], nb[4]]
        del dd

    iso_lat = []  # per-pair latencies of the one-pair-in-flight pass after the timed region
    iso_serial = []  # the same with the engine's latency mode off

    grouped = lockstep > 1 and args.path == 'engine'  # the pipeline runs the drawn pairs as a lock-step group BEFORE one_step sees them

    def layer_records(eng, pid, role):
        """The KPConv layer records of the engine's last run.  role 1: with event-bracketed durations; role 2: shapes only -- another
        pair of a lock-step group whose launches (and durations) are those of the group's first engine."""
        out = eng.kpconv_profile()
        for li, layer_rec in enumerate(out):
            layer_rec['pid'], layer_rec['layer'], layer_rec['launches'] = pid, li, 1 if role == 1 else 0
        return out

    def one_step(eng, slot, first, rec, lat_out, prof_out, events_every, n_workers, prepared=False):
        """One pair of the hot path on the worker thread / stream the pipeline hands it to."""
        i = first + slot
        ts = time.perf_counter()
        pid = pid_of(i)
        if net is None or args.path == 'engine':
            if prepared:  # (set_profile below chose before the group ran)
                role = eng._bench_prof if prof_out is not None else 0
            else:
                role = 1 if prof_out is not None and events_every > 0 and (slot // max(n_workers, 1)) % events_every == 0 else 0
                eng.enable_profile(role)
            res = eng.run(*dev_pairs[pid])  # returns with pose AND correspondences in host memory
            T, n_corr = eng.transform(), res.n_correspondences
            rc_h, sc_h, cs_h = eng.host_corr()  # "... to estimated_transform + correspondences on the host"
            assert rc_h.shape[0] == n_corr
            if role:
                prof_out.extend(layer_records(eng, pid, role))
        else:
            net.set_thread_profile(prof_out)
            try:
                T, n_corr = step(i)
            finally:
                net.set_thread_profile(None)
        if rec is not None:
            rre, rte = pose_error(T, pairs[pid][2])
            rec[slot] = torch.tensor([pid, rre, rte, n_corr, rank + i * world])
        if rec is not None or lat_out is iso_lat or lat_out is iso_serial:
            lat_out.append((time.perf_counter() - ts) * 1e3)

    def run_all(first, count, rec, lat_out, prof_lists, events_every=None):
        events_every = args.layer_events_every if events_every is None else events_every
        n_workers = len(streams)

        def set_profile(eng, slot, i, n):
            """Lock step: every events_every-th group of a stream records its layer events -- on its first engine; the others report
            their layers' shapes (the grouped launches serve all of them)."""
            on = prof_lists[worker_of[id(eng)]] is not None and events_every > 0
            if i == 0:  # (one engine of pps records events: the rate per PAIR stays what --layer-events-every says)
                # FULL groups only (n == pps): a launch's bytes are then those of pps pairs in a short run and a long one alike
                # (the end of a run draws smaller groups)
                eng._bench_sampled = on and n == pps and ((slot // pps) // max(n_workers, 1)) % max(1, events_every // pps) == 0
            sampled = pipe.groups[worker_of[id(eng)]][0]._bench_sampled
            eng._bench_prof = (1 if i == 0 else 2) if sampled else 0
            eng.enable_profile(eng._bench_prof)
        # (tensors_of: what one_step will hand to eng.run for this slot -- the pipeline's workers collate / run several drawn pairs at once)
        pipe.map(range(count), lambda eng, slot: one_step(eng, slot, first, rec, lat_out, prof_lists[worker_of[id(eng)]],
                                                          events_every, n_workers, prepared=grouped),
                 tensors_of=(lambda slot: dev_pairs[pid_of(first + slot)]) if args.path == 'engine' else None,
                 prepare=set_profile if grouped else None)
        if rec is not None and lat_out is not None and pipe.last_stats.get('latency_ms'):
            # a pair's latency counts from the moment its worker drew it: the batch's collate and the pairs before it included
            lat_out[:] = [pipe.last_stats['latency_ms'][k] for k in sorted(pipe.last_stats['latency_ms'])]

    # Clock ramp (untimed, before the W warm-up steps): the first GPU process on a freshly started box runs 15-20 % slower
    # for its first seconds -- idle host cores and GPU power states take that long to reach their steady clocks (measured:
    # 400 vs 475 pairs/s for a first run, 475 for every later one; 4.5 s of untimed pairs closes the gap).
    # The ramp also exercises the per-layer HIP events (every 2nd pair): in the first GPU process of a box the runtime's
    # pool of timing signals grows once, after ~500 event records, with a ~45 ms stall of all streams -- which the
    # default run would otherwise meet at its 97th-100th timed pair (-13 % on the 120-pair region that was the default then).
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < args.ramp_seconds:
        run_all(0, 16 * len(streams), None, [], [[] for _ in streams], events_every=2 if args.layer_events_every > 0 else 0)
    run_all(0, n_warm, None, [], [None] * len(streams))
    lat = []
    prof_lists = [[] for _ in streams]
    records = torch.zeros((n_timed, 5), dtype=torch.float32)  # [pair_id, rre_deg, rte_m, n_corr, global step index]
    fence()
    engine.Engine.lockstep_stats(reset=True)
    t0 = time.perf_counter()
    run_all(n_warm, n_timed, records, lat, prof_lists)
    group_sizes = dict(pipe.last_stats.get('group_sizes', {}))
    ls = engine.Engine.lockstep_stats()
    # the path's only collective: one gather of per-pair result records (RCCL)
    comm_dev = dev if args.dist_backend == 'nccl' else torch.device('cpu')  # gloo gathers CPU tensors
    gathered = sharding.gather_records(records.to(comm_dev), world, dist, force)
    fence()
    elapsed = time.perf_counter() - t0
    if os.environ.get('RDM_BENCH_DUMP_LAT'):  # developer: per-pair latencies in completion order
        json.dump(lat, open(os.environ['RDM_BENCH_DUMP_LAT'], 'w'))
    elapsed, lat = sharding.reduce_timing(elapsed, lat, world, dist, comm_dev, force)  # max over ranks; all ranks' latencies

    def timed_pass(n_steps, fn, **kw):
        """A second, shorter region after the timed one, same pairs in flight, fenced like it: -> pairs/s (whole job).
        kw: tensors_of / group_fn -- the pipeline then runs the drawn pairs as lock-step groups, the schedule of `value`."""
        fence()
        tp0 = time.perf_counter()
        pipe.map(range(n_steps), fn, **kw)
        fence()
        t_pass, _ = sharding.reduce_timing(time.perf_counter() - tp0, [], world, dist, comm_dev, force)
        return n_steps * world / t_pass

    def side(value_grouped, value_single, steps, note):
        """A side figure measured in the headline's schedule (lock-step groups, `value`) with the one-pair-per-call schedule of
        rounds 1-4 beside it."""
        out = {'value': value_grouped if grouped else value_single, 'unit': 'pairs/s', 'steps': steps,
               'schedule': (f'{len(streams)} streams x lock-step groups of {lockstep} (as `value`)' if grouped else f'{len(streams)} pairs in flight, one pair per call'),
               'note': note}
        if grouped:
            out['one_pair_per_call'] = {'value': value_single, 'unit': 'pairs/s', 'note': 'the same pass, one pair per engine call (the schedule of rounds 1-4)'}
        return out

    dev_tensors = lambda i: dev_pairs[pid_of(i)]

    # ---- the same engine path building ALL 13 search tables (a plain rdm_engine_run skips the up-sampling search of level 0,
    # which nothing reads, and keeps one column of the other three: `value` times 12 searches; docs/EXPERIMENTS.md 5d).  Side key,
    # never `value`: with the stage tensors kept the engine builds the reference's full tables (and every pair of a lock-step
    # group collates itself).
    full_tables = None
    if args.path == 'engine' and args.full_steps > 0:
        def full_step(eng, i):
            eng.enable_profile(False)
            res = eng.run(*dev_tensors(i))
            assert eng.host_corr()[0].shape[0] == res.n_correspondences
        all_engines = [e for grp in pipe.groups for e in grp]
        pipe.keep_taps = True
        for eng in all_engines:
            eng.keep_taps(True)
        try:
            pipe.map(range(4 * len(streams) * pps), full_step, tensors_of=dev_tensors if grouped else None)
            v_grouped = timed_pass(args.full_steps * (pps if grouped else 1), full_step, tensors_of=dev_tensors) if grouped else None
            pipe.map(range(4 * len(streams)), full_step)
            v_single = timed_pass(args.full_steps, full_step)
            full_tables = side(v_grouped, v_single, args.full_steps * (pps if grouped else 1),
                               'rdm_engine_run with every stage tensor kept: all 13 radius searches at the reference\'s '
                               'table widths (the 32 k-query up-sampling search of level 0 included), same pairs in flight')
        finally:
            pipe.keep_taps = False
            for eng in all_engines:
                eng.keep_taps(False)

    # ---- the schedule of rounds 1-4 beside the lock-step groups: the same engines, streams and pairs, one pair per engine call
    one_by_one = None
    if grouped and args.full_steps > 0:
        def single_step(eng, i):
            eng.enable_profile(False)
            res = eng.run(*dev_tensors(i))
            assert eng.host_corr()[0].shape[0] == res.n_correspondences
        pipe.map(range(4 * len(streams)), single_step)
        v_single = timed_pass(2 * args.full_steps, single_step)
        lat_single = sorted(pipe.last_stats.get('latency_ms', {}).values())
        one_by_one = {'value': v_single, 'unit': 'pairs/s', 'steps': 2 * args.full_steps,
                      'p50_ms_per_pair': float(np.median(lat_single)) if lat_single else None,
                      'note': f'rdm_engine_run, one pair per call, {len(streams)} pairs in flight (the schedule `value` was measured on up to '
                              'round 4; --lockstep 1 makes it the timed region)'}

    # ---- host-to-host rate (SURVEY §8d's definition of a pair: two clouds in HOST memory -> transform + correspondences
    # in HOST memory).  Never `value`: a second, shorter region after the timed one.  Each in-flight pair copies its scans
    # from pinned host memory on its worker's stream, runs the engine, and copies the correspondences back.
    host_to_host = None
    if args.path == 'engine' and args.host_steps > 0:
        pinned = [(torch.from_numpy(r).pin_memory(), torch.from_numpy(s_).pin_memory()) for r, s_, _ in pairs]
        staged = {}  # step -> the device copies its worker made (tensors_of runs on the worker's stream, right before the group)

        def h2d(i):
            pr, ps = pinned[pid_of(i)]
            staged[i] = (pr.to(dev, non_blocking=True), ps.to(dev, non_blocking=True))
            return staged[i]

        def h2h_step(eng, i):
            eng.enable_profile(False)
            r_, s_ = staged.pop(i, None) or h2d(i)
            staged.pop(i, None)
            res = eng.run(r_, s_)
            rc, sc, cs = eng.host_corr()  # written to pinned host memory by the run's last kernel; the pose too
            assert rc.shape[0] == res.n_correspondences and cs.shape[0] == res.n_correspondences

        v_grouped = None
        if grouped:
            pipe.map(range(2 * len(streams) * pps), h2h_step, tensors_of=h2d)
            v_grouped = timed_pass(args.host_steps * pps, h2h_step, tensors_of=h2d)
        v_single = timed_pass(args.host_steps, h2h_step)
        host_to_host = side(v_grouped, v_single, args.host_steps * (pps if grouped else 1),
                            'pinned host scans -> H2D -> engine -> D2H of correspondences (points + scores) and pose; '
                            'measured after the timed region, same pairs in flight')

    # ---- the drop-in operator API (north star: "keeps the existing model.forward operator API"): the same pairs through
    # rdmnet_amd.collate (the reference's collate signature) + model(data_dict) (rdm_engine_forward: one native call) -> the
    # reference's 31-key output_dict; in the headline's schedule a worker hands the data_dicts of the pairs it drew to ONE
    # model([data_dict, ...]) call (rdm_engine_forward_lockstep).  A second figure, never `value`.
    api = None
    if args.api_steps > 0:
        if net is None:
            net = model.create_model(cfg).cuda(local_rank)
            net.load_state_dict(state)
        net.pairs_in_flight = args.streams * lockstep

        def make_data(i):
            r, s_ = dev_tensors(i)
            item = {'ref_points': r, 'src_points': s_, 'ref_feats': torch.ones((r.shape[0], 1), device=dev),
                    'src_feats': torch.ones((s_.shape[0], 1), device=dev)}
            data = collate.registration_collate_fn_stack_mode([item], cfg.backbone.num_stages, cfg.backbone.init_voxel_size,
                                                              cfg.backbone.init_radius, cfg.neighbor_limits, device=dev,
                                                              engine=net.engine() if args.api_collate == 'native' else None)
            data['testing'] = True
            return data

        def api_step(eng, i):  # (the module keeps its own engines per stream; `eng` of the pipeline idles in this pass)
            out = net(make_data(i))
            out['estimated_transform'].cpu()

        def make_data_group(steps_):  # the drawn pairs' collates as one lock-step group on the module's engines of this stream
            if args.api_collate != 'native' or len(steps_) < 2:
                return [make_data(i) for i in steps_]
            items = []
            for i in steps_:
                r, s_ = dev_tensors(i)
                items.append({'ref_points': r, 'src_points': s_, 'ref_feats': torch.ones((r.shape[0], 1), device=dev),
                              'src_feats': torch.ones((s_.shape[0], 1), device=dev), 'testing': True})
            return collate.registration_collate_lockstep(items, cfg.backbone.num_stages, cfg.backbone.init_voxel_size, cfg.backbone.init_radius,
                                                         cfg.neighbor_limits, net.engine_group(len(items)), device=dev)

        def api_group(engs, steps_):  # the collates of the drawn pairs as one group, then their data_dicts through ONE model([...]) call
            outs = net(make_data_group(steps_))
            for out in outs:
                assert len(out) == 31
                out['estimated_transform'].cpu()
            return [None] * len(steps_)

        v_grouped = None
        if grouped:
            pipe.map(range(len(streams) * pps * 3), None, group_fn=api_group)  # warm-up: the module builds its per-stream engines here
            v_grouped = timed_pass(args.api_steps, None, group_fn=api_group)
        pipe.map(range(len(streams) * 6), api_step)
        v_single = timed_pass(args.api_steps, api_step)
        # forward only, one pair in flight: model(data_dict) (output_dict assembled) against the bare native call
        fwd_model, fwd_native = [], []
        eng_f = net._engine()
        for i in range(8):
            data = make_data(i)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            out = net(data)
            torch.cuda.synchronize(dev)
            t2 = time.perf_counter()
            eng_f.forward(data)
            torch.cuda.synchronize(dev)
            t3 = time.perf_counter()
            fwd_model.append((t2 - t1) * 1e3)
            fwd_native.append((t3 - t2) * 1e3)
        api = side(v_grouped, v_single, args.api_steps,
                   'rdmnet_amd.collate.registration_collate_fn_stack_mode (engine=model.engine(): one native call per pair) + '
                   'rdmnet_amd.model.RDMNet.__call__ (31-key output_dicts; a list of data_dicts = one lock-step group), same pairs in '
                   'flight; forward_only_ms: medians of 8 pairs, one in flight')
        api['forward_only_ms'] = {'model(data_dict)': float(np.median(fwd_model)), 'rdm_engine_forward': float(np.median(fwd_native))}
        api['collate'] = args.api_collate
        net.release_engines()

    # ---- roofline of the KPConv layers from HIP events recorded on the launch stream
    # Layer forms (rdm_kpconv_profile.fused): c_in = 1 / 32 / 64 run as ONE kernel (kpconv_fused* / kpconv_tile*: neighbourhood gather + weight
    # contraction, no [M, 15 C] tensor in HBM; 6 launches per pair), c_in >= 128 as kpconv_gather_kernel + gemm_kernel (8).
    # `gather_ms` of a record is the neighbourhood kernel alone in both forms.
    def kp_totals(prof):
        tot = {'t_total': 0.0, 'b_total': 0.0, 'n': 0}
        forms = {f: {'t': 0.0, 'b': 0.0, 'b_real': 0.0, 'b_moved': 0.0, 'flops': 0.0, 'n': 0} for f in ('fused', 'gather')}
        per_layer = {}
        for rec in prof:
            if 'events' in rec:  # per-op Python mirror: the events themselves
                e0, e1, e2 = rec.pop('events')
                rec['gather_ms'], rec['total_ms'] = e0.elapsed_time(e1), e0.elapsed_time(e2)
            tg, tt = rec['gather_ms'] * 1e-3, rec['total_ms'] * 1e-3
            tot['t_total'] += tt
            tot['b_total'] += rec['bytes']
            tot['n'] += rec.get('launches', 1)
            f = forms['fused' if rec.get('fused') else 'gather']
            f['t'] += tg
            f['b'] += rec['gather_bytes']
            f['b_real'] += rec.get('real_bytes', 0.0)
            f['b_moved'] += rec.get('moved_bytes', 0.0)
            f['flops'] += 2.0 * rec['m'] * (16 if rec['cin'] == 1 else 15 * rec['cin']) * rec['cout'] if rec.get('fused') else 0.0
            f['n'] += rec.get('launches', 1)  # (lock step: one launch serves the group; the other pairs' records add their bytes)
            key = rec.get('name') or f"kpconv M={rec['m']} H={rec['h']} C={rec['cin']}->{rec['cout']}"
            d = per_layer.setdefault(key, {'t': 0.0, 'tg': 0.0, 'bytes': rec['bytes'], 'n': 0, 'm': rec['m'], 'fused': rec.get('fused', 0),
                                           'h': rec['h'], 'cin': rec['cin'], 'cout': rec['cout'], 'real_bytes': rec.get('real_bytes')})
            d['t'] += tt
            d['tg'] += tg
            d['n'] += 1
        return tot, forms, per_layer

    def form_line(f):
        if f['n'] == 0 or f['t'] <= 0:
            return None
        gbs = f['b'] / f['t'] / 1e9
        out = {'achieved': gbs, 'frac': gbs / HBM_PEAK_GBS, 'launches': f['n'], 'us_per_launch': f['t'] / f['n'] * 1e6,
               'bytes_per_launch': f['b'] / f['n']}
        if f['b_real'] > 0:
            out['real_slots'] = {'achieved': f['b_real'] / f['t'] / 1e9, 'frac': f['b_real'] / f['t'] / 1e9 / HBM_PEAK_GBS,
                                 'bytes_per_launch': f['b_real'] / f['n'], 'fill': f['b_real'] / f['b']}
        if f.get('b_moved', 0) > 0:  # what the kernels request: slots holding a neighbour, int32 indices (<= 1 of the peak for every form)
            out['moved_bytes'] = {'achieved': f['b_moved'] / f['t'] / 1e9, 'frac': f['b_moved'] / f['t'] / 1e9 / HBM_PEAK_GBS,
                                  'bytes_per_launch': f['b_moved'] / f['n']}
        if f['flops'] > 0:  # the fused kernels also do the weight contraction: the bound of one launch is max(bytes / HBM, flops / MFMA)
            t_bound = max(f['b'] / (HBM_PEAK_GBS * 1e9), f['flops'] / (MFMA_F32_PEAK_TF * 1e12))
            out['mfma_tflops'] = f['flops'] / f['t'] / 1e12
            out['frac_of_max_hbm_mfma_bound'] = t_bound / f['t']
        return out

    def roofline_of(forms):
        both = {k: forms['fused'][k] + forms['gather'][k] for k in ('t', 'b', 'b_real', 'b_moved', 'flops', 'n')}
        both['flops'] = 0.0
        line = form_line(both) or {'achieved': 0.0, 'frac': 0.0, 'launches': 0, 'us_per_launch': 0.0, 'bytes_per_launch': 0.0}
        line['by_form'] = {'one_kernel_layers (c_in 1/32/64: kpconv_fused*, gather + weight contraction)': form_line(forms['fused']),
                           'gather_kernel_layers (c_in >= 128: kpconv_gather_kernel alone, as round 2 reported all 14)': form_line(forms['gather'])}
        return line

    def add_slot_bytes(records):
        """real_bytes: the contract's per-slot bytes (8-byte index + xyz + feature row) on the slots that hold a neighbour;
        moved_bytes: the same slots as the engine's kernels address them -- int32 tables (a plain run), rows cut at the last
        real neighbour -- i.e. what is requested from the memory system, whichever cache serves it."""
        for rec in records:
            if 'pid' in rec and real_slots.get(rec['pid']):
                n_real = real_slots[rec['pid']][rec['layer']]
                rec['real_bytes'] = n_real * (8 + 12 + 4 * rec['cin'])
                rec['moved_bytes'] = n_real * ((4 if args.path == 'engine' else 8) + 12 + 4 * rec['cin'])

    prof = [r for pl in prof_lists for r in pl]
    add_slot_bytes(prof)
    tot, forms, per_layer = kp_totals(prof)
    # With several pairs in flight the event-bracketed durations above include the time a KPConv kernel
    # shares the CUs with other pairs' kernels.  A short single-stream pass after the timed region gives the
    # same kernels' durations when they own the GPU (reported beside, never instead of, the timed-region figure).
    # Two passes: the roofline pass (HIP events around every KPConv layer of every pair, latency mode off so that the kernels
    # really have the GPU to themselves) and the latency pass (no events -- 42 event records cost a pair 0.2-0.3 ms -- and the
    # engine as a user with one pair in flight gets it: latency mode on, rdm_engine_set_overlap).
    iso_prof = []
    iso_single_prof = []  # single pairs alone (one launch = one pair): the figure of rounds 1-4, whatever the timed region's schedule
    if args.path == 'engine':
        engines[0].set_pairs_in_flight(1)  # (these passes ARE one pair in flight: no GEMM residency cap)
        with torch.cuda.stream(streams[0] if streams[0] is not None else torch.cuda.Stream()):  # (never the null stream)
            engines[0].set_overlap(0)
            if grouped:  # the timed region's launches are lock-step groups: 6 groups alone on the GPU, events on the first engine
                from rdmnet_amd.engine import Engine
                grp = pipe.groups[0]
                for g in range(6):
                    pids = [pid_of(n_warm + g * pps + i) for i in range(pps)]
                    for i, e in enumerate(grp):
                        e.enable_profile(1 if i == 0 else 2)
                    Engine.run_lockstep(grp, [dev_pairs[p] for p in pids])
                    for i, e in enumerate(grp):
                        iso_prof.extend(layer_records(e, pids[i], 1 if i == 0 else 2))
                for e in grp:
                    e.enable_profile(0)
                    e.clear_pending()  # (the groups' results are not picked up through run(): they must not answer the serial passes below)
            for k in range(min(8, n_timed)):
                one_step(engines[0], k, n_warm, None, [], iso_single_prof, 1, 1)
            if not grouped:
                iso_prof = iso_single_prof
            for k in range(min(16, n_timed)):  # serial, no events: the figure comparable with earlier rounds' (minus their events)
                one_step(engines[0], k, n_warm, None, iso_serial, None, 0, 1)
            engines[0].set_overlap(1)
            for k in range(min(4, n_timed)):  # (the first latency-mode run picks the side stream)
                one_step(engines[0], k, n_warm, None, [], None, 0, 1)
            for k in range(min(24, n_timed)):
                one_step(engines[0], k, n_warm, None, iso_lat, None, 0, 1)
        engines[0].set_pairs_in_flight(args.streams)
        fence()
    add_slot_bytes(iso_prof)
    if iso_single_prof is not iso_prof:
        add_slot_bytes(iso_single_prof)
    itot, iforms, per_layer_iso = kp_totals(iso_prof)
    _, sforms, _ = kp_totals([dict(r) for r in iso_single_prof]) if iso_single_prof is not iso_prof else (None, iforms, None)
    if iso_prof:
        per_layer = per_layer_iso

    def mfma_totals(records):
        """fp32 FLOPs and seconds of the KPConv weight contractions [M, 15 C] x [15 C, C'] that run on gemm_kernel: the
        non-strided two-kernel layers (the strided layers' second event interval also holds the shortcut max-pool; the
        one-kernel layers contract inside kpconv_fused*)."""
        fl = tt = 0.0
        for rec in records:
            if rec.get('pooled') or rec.get('fused'):
                continue
            fl += 2.0 * rec['m'] * 15 * rec['cin'] * rec['cout']
            tt += (rec['total_ms'] - rec['gather_ms']) * 1e-3
        return fl, tt

    n_layers = max(len(prof), 1)
    traffic, traffic_note = None, None
    # HBM bytes per dispatch from the committed rocprofv3 --pmc passes (same kernels, same kind of dispatch as `achieved`: since
    # round 5 `*_pmc_kpconv_gather.json` holds the lock-step grouped dispatches, `*_one_pair.json` the one-pair ones)
    for tag in ('r06', 'r05', 'r04', 'r03', 'r02', 'r01'):
        names = [f'{tag}_pmc_kpconv_gather.json'] if grouped else [f'{tag}_pmc_kpconv_gather_one_pair.json', f'{tag}_pmc_kpconv_gather.json']
        for nm in names:
            pmc_file = os.path.join(ROOT, 'profiles', nm)
            if not os.path.exists(pmc_file):
                continue
            pmc = json.load(open(pmc_file))
            if ('grouped' in pmc['kernel']) != grouped:
                continue
            traffic, traffic_note = pmc['traffic_bytes_per_dispatch'], pmc['kernel'] + '; ' + pmc['source']
            break
        if traffic is not None:
            break
    # `roofline`: the kernels the north star names -- the KPConv neighbourhood kernels, 14 launches per pair.  achieved =
    # SURVEY 8d gather bytes M*H*(8 + 12 + 4*C_in) per launch (padded slots counted: the contract figure) / the launch's
    # duration (HIP events on its stream); `real_slots` = the same with the slots that hold a neighbour only (the kernels
    # stop at a row's last real neighbour).  `achieved`/`frac` are the TIMED REGION's (several pairs share the GPU),
    # `group_alone` / `single_pair_alone` the same kernels with the GPU to themselves; `traffic` (PMC) covers the same kernels.
    # `timed_marginal`: what the KPConv neighbourhood kernels COST in the saturated schedule -- every launch of the class issued
    # twice (lab build, RDM_DUP; tools/exp_dup_lockstep.sh -> profiles/r0x_marginal_cost.json, committed), ms per pair added --
    # against the padded-slot bytes of a pair (this run's layer records): the timed-region figure that does not measure sharing.
    timed_marginal = None
    for tag in ('r06', 'r05'):
        mfile = os.path.join(ROOT, 'profiles', f'{tag}_marginal_cost.json')
        if not os.path.exists(mfile) or not grouped:
            continue
        mj = json.load(open(mfile))
        ref_forms = sforms if iso_single_prof else forms
        n_pairs_ref = max(len(iso_single_prof if iso_single_prof else prof) / 14.0, 1.0)
        by = {}
        for form, cls in (('fused', 'fused'), ('gather', 'gather')):
            ms = mj['classes'].get(cls)
            if ms and ms > 0 and ref_forms[form]['b'] > 0:
                bpp = ref_forms[form]['b'] / n_pairs_ref
                by[form] = {'ms_per_pair': ms, 'bytes_per_pair': bpp, 'achieved': bpp / (ms * 1e-3) / 1e9, 'frac': bpp / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        if by:
            ms_all = sum(v['ms_per_pair'] for v in by.values())
            b_all = sum(v['bytes_per_pair'] for v in by.values())
            timed_marginal = {'achieved': b_all / (ms_all * 1e-3) / 1e9, 'frac': b_all / (ms_all * 1e-3) / 1e9 / HBM_PEAK_GBS, 'ms_per_pair': ms_all,
                              'by_form': {'one_kernel_layers': by.get('fused'), 'gather_kernel_layers': by.get('gather')},
                              'source': mj.get('source'), 'note': 'marginal cost in the lock-step schedule (launches of the class issued twice, lab build): '
                                                                  'from the committed profile, not measured in this run'}
        break
    roofline = {'bound': 'hbm', 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', **roofline_of(forms),
                'definition': 'HEADLINE `achieved` / `frac` (fixed since r03): padded-slot bytes M*H*(8 + 12 + 4*C_in) of the 14 KPConv neighbourhood '
                              'launches of a pair / their summed durations (HIP events on the launch stream) in the timed region.  In the lock-step '
                              'schedule that duration is mostly time spent SHARING the GPU with the other streams\' grouped kernels, so the figure to '
                              'hold against the north star\'s 0.60 is `group_alone` (the same launches with the GPU to themselves: the kernel) with '
                              '`timed_marginal` beside it (what the class costs per pair in the saturated schedule: the schedule); `single_pair_alone` is '
                              'the round-1-4 quantity.  Every one of them also as real_slots (slots holding a neighbour) and moved_bytes (those slots with '
                              'the int32 indices the engine uses: what the kernels request -- <= 1 of the peak for every form; the padded-slot contract '
                              'figure of a gather-only layer alone can exceed 1), by_form (one-kernel vs gather-only layers), whole_layer (round-comparable '
                              'layer figure).  A launch of a lock-step schedule is a grouped launch of FULL groups (config.lockstep_pairs_per_stream pairs): '
                              'bytes and duration are the group\'s',
                'traffic': traffic, 'traffic_scope': traffic_note,
                'kernel': 'KPConv neighbourhood kernels, 14 launches/pair: kpconv_fused_c1_kernel + kpconv_tile_kernel<32|64> / '
                          'kpconv_fused_kernel<64> (6: gather + weight contraction in one launch) and kpconv_gather_kernel<*> (8)',
                'region': f'timed region, {len(streams) * lockstep} pair(s) in flight' + (f' ({len(streams)} streams x lock-step groups of {lockstep}: a launch serves '
                                                                                           f'the {lockstep} pairs of its group, bytes and duration are the group\'s)' if grouped else ''),
                'pairs_with_layer_events': len(prof) // 14,
                # the same kernels with the GPU to themselves, after the timed region, on one stream:
                'group_alone': ({**roofline_of(iforms), 'note': 'the timed region\'s launches alone: 6 lock-step groups (a launch serves the group\'s pairs) on one stream '
                                                                '(round 5 printed this as `one_pair_in_flight`)'} if iso_prof and grouped else None),
                'single_pair_alone': ({**roofline_of(sforms), 'note': 'one pair per launch, 8 pairs on one stream: the quantity rounds 1-4 reported as `one_pair_in_flight`'}
                                      if iso_single_prof else None),
                'timed_marginal': timed_marginal,
                # the whole KPConv layer (+ weight GEMM / GroupNorm passes + shortcut pool) against the same HBM peak, as round 1 reported it
                'whole_layer': {'definition': 'SURVEY 8d bytes of the WHOLE KPConv layer (gather + 4 M C_out output + the strided blocks\' pool) / the '
                                               'time from the layer\'s first launch to its last (events 0 -> 2): the same quantity in every round, '
                                               'whichever kernels the layer is made of (r01 0.34-0.36, r02 0.38, r03 0.39 of the HBM peak, one pair in flight)',
                                 'kernels': 'neighbourhood kernel + gemm_kernel (weights, two-kernel layers) + GroupNorm passes (one-kernel layers) + gather_max (strided layers)',
                                 'bytes_per_launch': tot['b_total'] / n_layers,
                                 'timed_region': {'achieved': tot['b_total'] / tot['t_total'] / 1e9 if tot['t_total'] > 0 else 0.0,
                                                  'frac': tot['b_total'] / tot['t_total'] / 1e9 / HBM_PEAK_GBS if tot['t_total'] > 0 else 0.0,
                                                  'us_per_launch': tot['t_total'] / n_layers * 1e6},
                                 ('group_alone' if grouped else 'single_pair_alone'): ({'achieved': itot['b_total'] / itot['t_total'] / 1e9,
                                                         'frac': itot['b_total'] / itot['t_total'] / 1e9 / HBM_PEAK_GBS,
                                                         'us_per_launch': itot['t_total'] / max(len(iso_prof), 1) * 1e6} if itot['t_total'] > 0 else None),
                                 'ms_per_pair_timed_region': tot['t_total'] / max(len(prof) / 14.0, 1.0) * 1e3}}
    # `roofline_mfma`: the kernel family that dominates the kernel time (gemm_kernel, fp32 MFMA): the KPConv weight
    # contractions, from the same per-layer events; peak = 157.3 TFLOP/s (v_mfma_f32_32x32x2_f32, MI355X_MICROARCH.md)
    mf, mt = mfma_totals(prof)
    imf, imt = mfma_totals(iso_prof)
    roofline_mfma = {'bound': 'mfma', 'kernel': 'gemm_kernel<*> on the KPConv weight contractions [M,15C]x[15C,C\'] (6 non-strided two-kernel layers/pair, c_in >= 128)',
                     'achieved': mf / mt / 1e12 if mt > 0 else 0.0, 'peak': MFMA_F32_PEAK_TF, 'unit': 'TFLOP/s',
                     'frac': mf / mt / 1e12 / MFMA_F32_PEAK_TF if mt > 0 else 0.0, 'region': f'timed region, {len(streams)} pair(s) in flight',
                     ('group_alone' if grouped else 'single_pair_alone'): ({'achieved': imf / imt / 1e12, 'frac': imf / imt / 1e12 / MFMA_F32_PEAK_TF} if imt > 0 else None)}
    if grouped and iso_single_prof:
        smf, smt = mfma_totals(iso_single_prof)
        roofline_mfma['single_pair_alone'] = {'achieved': smf / smt / 1e12, 'frac': smf / smt / 1e12 / MFMA_F32_PEAK_TF} if smt > 0 else None

    if rank == 0:
        result = {
            'metric': 'scan-pairs/sec (whole node)', 'value': n_timed * world / elapsed, 'unit': 'pairs/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'KITTI-shaped synthetic pair (~16k pts/scan), full pipeline (GPU collate + forward), '
                                   'fp32, seeded random-init weights', 'searches_per_pair': 12 if args.path == 'engine' else 13,
                       'scheduler': 'rdmnet_amd.pipeline.PairPipeline', 'points_per_pair': n_points,
                       'pairs_per_step': pps, 'pairs_per_gpu': n_timed, 'pairs_in_flight_per_gpu': args.streams * lockstep,
                       'streams_per_gpu': args.streams, 'lockstep_pairs_per_stream': lockstep, 'collate_batch': collate_batch,
                       'arena_mb_per_engine': args.arena_mb if args.arena_mb > 0 else 3072,
                       'lockstep_groups_by_size': {str(k): v for k, v in sorted(group_sizes.items())} if grouped else None,
                       'lockstep_records_per_launch': (ls['records'] / ls['launches'] if ls['launches'] else None) if grouped else None,
                       'host_path': args.path,
                       'host_cpus_per_rank': budget, 'host_cpus_pinned': len(pinned_cpus) if pinned_cpus else None,
                       'gpu_max_hw_queues': pipeline.hw_queues(), 'clock_ramp_s': args.ramp_seconds, 'wait': 'spin' if wait_us == 0 else f'poll+sleep {wait_us}us',
                       'parallelism': f'pairs sharded over {world} GPU(s)'},
            'p50_ms_per_pair': float(np.median(lat)),
            'one_pair_in_flight': ({'p50_ms_per_pair': float(np.median(iso_lat)), 'pairs': len(iso_lat),
                                    'serial_p50_ms_per_pair': float(np.median(iso_serial)) if iso_serial else None,
                                    'note': 'latency with the GPU to one pair: 24 pairs on one stream after the timed region, no HIP events, '
                                            'the engine in its latency mode (rdm_engine_set_overlap, the default with one pair in flight); '
                                            'serial_p50 = the same with that mode off (16 pairs)'}
                                   if iso_lat else None),
            'mean_ms_per_pair_by_quarter': [float(np.mean(q)) for q in np.array_split(np.asarray(lat), 4)] if len(lat) >= 4 else None,
            'registration': {**sharding.summarize(gathered), 'note': 'random-init weights: accuracy is not meaningful'},
            'records': {'gathered': int(sum(g.shape[0] for g in gathered)),
                        'distinct_steps': len({int(x) for g in gathered for x in g[:, 4].tolist()}),
                        'distinct_pairs': len({int(x) for g in gathered for x in g[:, 0].tolist()})},
            'collective': ({'backend': args.dist_backend, 'world': world, 'forced_single_rank': bool(force),
                            'library': ('RCCL ' + '.'.join(str(v) for v in torch.cuda.nccl.version())) if args.dist_backend == 'nccl' else 'gloo',
                            'ops': 'barrier x2 per region, all_gather (counts + records), all_reduce(MAX) of the elapsed time',
                            'preflight': preflight}
                           if dist is not None else None),
            'one_pair_per_call': one_by_one,
            'full_tables': full_tables,
            'host_to_host': host_to_host,
            'drop_in_api': api,
            'roofline': roofline,
            'roofline_mfma': roofline_mfma,
        }
        if not args.no_cpu_baseline and world == 1:
            result['cpu_baseline'] = cpu_baseline(args.cache, args.pairs)
        else:
            result['cpu_baseline'] = None
        os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
        with open(os.path.join(ROOT, 'gpurun_out', 'bench_layers.json'), 'w') as f:
            json.dump({k: {**v, 'us': v['t'] / v['n'] * 1e6, 'gather_us': v['tg'] / v['n'] * 1e6,
                           'GBps': v['bytes'] * v['n'] / v['t'] / 1e9} for k, v in per_layer.items() if v['t'] > 0}, f, indent=1)
        print(json.dumps(_no_nan(result)))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
