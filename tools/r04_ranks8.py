"""Eight ranks on ONE device (the launcher / sharding / gather path of an 8-GPU node, minus the other seven GPUs): pairs/s, host CPU
seconds of all ranks (children rusage), and that every step is accounted for once.   gpurun -- 'python tools/r04_ranks8.py'"""
import json, os, resource, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
print(f'host: {os.cpu_count()} CPUs visible, affinity {len(os.sched_getaffinity(0))}, cpu.max {open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else None}')
print('| run | rc | pairs/s (8 ranks x 4 pairs in flight on one GPU) | p50 ms | records gathered / distinct steps | wait mode | CPUs per rank (budget / pinned) | wall s | CPU s of all ranks (user + sys) | busy cores | note |')
print('|---|---|---|---|---|---|---|---|---|---|---|')
for name, backend, extra in (('nccl (RCCL), 8 ranks on one device', 'nccl', []), ('gloo, pinned slices, poll', 'gloo', []), ('gloo, not pinned', 'gloo', ['--pin', 'off']),
                             ('gloo, pinned, spinning waits', 'gloo', ['--wait-us', '0'])):
    env = dict(os.environ, RDM_BENCH_SHARE_DEVICE='1')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'MASTER_PORT'):
        env.pop(k, None)
    r0 = resource.getrusage(resource.RUSAGE_CHILDREN)
    t0 = time.time()
    try:
        p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '64', '--warmup', '8', '--ramp-seconds', '2', '--pairs', '4',
                            '--host-steps', '0', '--api-steps', '0', '--full-steps', '0', '--no-cpu-baseline', '--real-slots', 'off', '--dist-backend', backend] + extra,
                           capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
        rc, out, err = p.returncode, p.stdout, p.stderr
    except subprocess.TimeoutExpired as e:
        rc, out, err = -9, (e.stdout or b'').decode() if isinstance(e.stdout, bytes) else (e.stdout or ''), 'timeout'
    wall = time.time() - t0
    r1 = resource.getrusage(resource.RUSAGE_CHILDREN)
    cpu = (r1.ru_utime - r0.ru_utime) + (r1.ru_stime - r0.ru_stime)
    d = None
    for l in out.splitlines():
        if l.startswith('{'):
            d = json.loads(l)
    if d:
        print(f"| {name} | {rc} | {d['value']:.1f} | {d['p50_ms_per_pair']:.1f} | {d['records']['gathered']} / {d['records']['distinct_steps']} | {d['config']['wait']} | "
              f"{d['config']['host_cpus_per_rank']:.1f} / {d['config'].get('host_cpus_pinned')} | {wall:.1f} | {cpu:.1f} | {cpu / wall:.1f} | {d['collective']['library']} |")
    else:
        tail = [x for x in err.strip().splitlines() if x.strip()][-3:]
        print(f"| {name} | {rc} | - | - | - | - | - | {wall:.1f} | {cpu:.1f} | {cpu / wall:.1f} | " + ' / '.join(t[:160] for t in tail).replace('|', '/') + ' |')
    sys.stdout.flush()
