"""Eight ranks on ONE device in the lock-step default (the launcher / sharding / gather path of an 8-GPU node, minus the other seven
GPUs): every rank runs the product scheduler -- 4 host threads x lock-step groups of 4 pairs, 16 engines -- so the
one GPU carries 128 engines and 128 pairs in flight.  Reports pairs/s, the host CPU seconds of all ranks (children rusage), the peak
HBM in use (rocm-smi, sampled) and that every step is accounted for once.     gpurun -- 'python tools/ranks8_lockstep.py > gpurun_out/ranks8.md'"""
import json, os, resource, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def vram_used_gib():
    try:
        out = subprocess.run(['rocm-smi', '--showmeminfo', 'vram', '--json'], capture_output=True, text=True, timeout=10).stdout
        d = json.loads(out)
        return max(int(v.get('VRAM Total Used Memory (B)', 0)) for v in d.values() if isinstance(v, dict)) / 2 ** 30
    except Exception:
        return float('nan')


print(f'host: {os.cpu_count()} CPUs visible, affinity {len(os.sched_getaffinity(0))}, cpu.max '
      f'{open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else None}; HBM in use before: {vram_used_gib():.1f} GiB')
print('| run | rc | pairs/s (all ranks, one GPU) | p50 ms | records gathered / distinct steps / distinct clouds | lock-step groups by size, records per launch | wait mode | '
      'CPUs per rank (budget / pinned) | wall s | CPU s of all ranks | busy cores | peak HBM in use GiB | note |')
print('|---|---|---|---|---|---|---|---|---|---|---|---|---|')
runs = (('8 ranks, gloo, 4 x 4 lock step, 1280 MiB arenas', '8', 'gloo', ['--arena-mb', '1280']),
        ('8 ranks, gloo, 4 x 4 lock step, spinning waits', '8', 'gloo', ['--arena-mb', '1280', '--wait-us', '0']),
        ('8 ranks, gloo, 2 x 4 lock step (2 streams per rank)', '8', 'gloo', ['--arena-mb', '1280', '--streams', '2']),
        ('8 ranks, gloo, 4 x 1 (one pair per call, rounds 1-4)', '8', 'gloo', ['--arena-mb', '1280', '--lockstep', '1']),
        ('2 ranks, gloo, 4 x 4 lock step, default arenas', '2', 'gloo', []),
        ('8 ranks, nccl (RCCL) on one device', '8', 'nccl', ['--arena-mb', '1280']))
for name, world, backend, extra in runs:
    env = dict(os.environ, RDM_BENCH_SHARE_DEVICE='1')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'MASTER_PORT'):
        env.pop(k, None)
    peak, stop = [vram_used_gib()], threading.Event()

    def sample():
        while not stop.wait(0.5):
            peak.append(vram_used_gib())
    th = threading.Thread(target=sample, daemon=True)
    th.start()
    r0 = resource.getrusage(resource.RUSAGE_CHILDREN)
    t0 = time.time()
    try:
        p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', world, '--steps', '16', '--warmup', '2', '--ramp-seconds', '2', '--pairs', '8',
                            '--host-steps', '0', '--api-steps', '0', '--full-steps', '0', '--no-cpu-baseline', '--real-slots', 'off', '--layer-events-every', '0',
                            '--dist-backend', backend] + extra, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
        rc, out, err = p.returncode, p.stdout, p.stderr
    except subprocess.TimeoutExpired as e:
        rc, out, err = -9, (e.stdout or b'').decode() if isinstance(e.stdout, bytes) else (e.stdout or ''), 'timeout'
    wall = time.time() - t0
    stop.set()
    th.join()
    r1 = resource.getrusage(resource.RUSAGE_CHILDREN)
    cpu = (r1.ru_utime - r0.ru_utime) + (r1.ru_stime - r0.ru_stime)
    d = None
    for l in out.splitlines():
        if l.startswith('{'):
            d = json.loads(l)
    if d:
        c = d['config']
        print(f"| {name} | {rc} | {d['value']:.1f} | {d['p50_ms_per_pair']:.1f} | {d['records']['gathered']} / {d['records']['distinct_steps']} / {d['records']['distinct_pairs']} | "
              f"{c.get('lockstep_groups_by_size')}, {(c.get('lockstep_records_per_launch') or 0):.2f} | {c['wait']} | {c['host_cpus_per_rank']:.1f} / {c.get('host_cpus_pinned')} | {wall:.1f} | {cpu:.1f} | "
              f"{cpu / wall:.1f} | {max(x for x in peak if x == x):.1f} | {(d.get('collective') or {}).get('library')} |")
    else:
        lines = [x for x in err.strip().splitlines() if x.strip()]
        tail = [x for x in lines if 'Duplicate GPU' in x or 'Error' in x][:2] or lines[-3:]
        print(f"| {name} | {rc} | - | - | - | - | - | - | {wall:.1f} | {cpu:.1f} | {cpu / wall:.1f} | {max(x for x in peak if x == x):.1f} | " + ' / '.join(t[:160] for t in tail).replace('|', '/') + ' |')
    sys.stdout.flush()
