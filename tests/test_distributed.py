"""CPU, world size 2, gloo: the multi-GPU layout (rank-strided pair sharding + one record gather)."""
import os

import pytest

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rdmnet_amd import sharding


def _worker(rank, world, port, n_pairs, out_q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    mine = sharding.pairs_for_rank(n_pairs, rank, world)
    rec = torch.tensor([[p, 0.1 * p, 0.01 * p, 100 + p] for p in mine], dtype=torch.float32).reshape(-1, 4)
    allr = sharding.gather_records(rec, world, dist)
    # the bench's timing reduction: max of the ranks' elapsed times, all latencies (ragged) in rank order
    elapsed, lat = sharding.reduce_timing(1.0 + rank, [10.0 * rank + k for k in range(len(mine))], world, dist)
    if rank == 0:
        out_q.put(([r.tolist() for r in allr], elapsed, lat))
    dist.barrier()
    dist.destroy_process_group()


def test_rank_strided_sharding_and_gather_world2():
    n_pairs, world = 7, 2  # ragged: rank 0 gets 4 pairs, rank 1 gets 3 -- no padding by repetition
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, 29617, n_pairs, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, elapsed, lat = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [len(g) for g in got] == [4, 3]
    ids = sorted(int(r[0]) for g in got for r in g)
    assert ids == list(range(n_pairs))  # every pair exactly once
    summary = sharding.summarize([torch.tensor(g) for g in got])
    assert summary['pairs'] == n_pairs and summary['recall'] == 1.0
    assert elapsed == 2.0 and lat == [0.0, 1.0, 2.0, 3.0, 10.0, 11.0, 12.0]


def test_single_rank_is_identity():
    rec = torch.ones(3, 4)
    assert sharding.gather_records(rec, 1)[0] is rec
    assert sharding.pairs_for_rank(5, 0, 1) == [0, 1, 2, 3, 4]
    assert sharding.reduce_timing(0.5, [1.0, 2.0], 1) == (0.5, [1.0, 2.0])


def _forced_single_worker(port, out_q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=0, world_size=1)
    rec = torch.arange(10, dtype=torch.float32).reshape(5, 2)
    got = sharding.gather_records(rec, 1, dist, force=True)
    out_q.put(([g.tolist() for g in got], got[0] is rec, sharding.reduce_timing(0.25, [3.0, 4.0], 1, dist, force=True)))
    dist.destroy_process_group()


def test_forced_single_rank_goes_through_the_collectives():
    """bench.py --force-dist: a world of one still calls all_gather / all_reduce (on the GPU box: RCCL)."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_forced_single_worker, args=(29633, q))
    p.start()
    got, same_object, timing = q.get(timeout=120)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert got == [torch.arange(10, dtype=torch.float32).reshape(5, 2).tolist()] and not same_object  # a gathered copy
    assert timing == (0.25, [3.0, 4.0])


def test_bench_self_spawns_its_ranks(tmp_path):
    """`python bench.py --gpus N` without torch.distributed.run: the launcher half (bench.spawn_ranks) starts N copies
    of the script with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set and propagates a failing rank's exit code.  The
    children here are a stand-in script (no GPU in this test); the GPU test runs the real thing on one device."""
    import subprocess
    import sys
    import textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bench_src = open(os.path.join(root, 'bench.py')).read()
    start = bench_src.index('def spawn_ranks(')
    end = bench_src.index('\ndef main(')
    script = tmp_path / 'fake_bench.py'
    script.write_text('import os, sys, time\n' + bench_src[start:end] + textwrap.dedent('''
        if __name__ == '__main__':
            if 'WORLD_SIZE' not in os.environ:
                sys.exit(spawn_ranks(int(sys.argv[1])))
            r = os.environ['RANK']
            open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'rank' + r), 'w').write(' '.join(
                os.environ[k] for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')))
            if len(sys.argv) > 2 and sys.argv[2] == 'fail' and r == '1':
                sys.exit(7)
            if len(sys.argv) > 2 and sys.argv[2] == 'fail':
                time.sleep(60)  # must be terminated by the launcher, not waited for
    '''))
    assert subprocess.run([sys.executable, str(script), '3'], timeout=60).returncode == 0
    envs = [(tmp_path / f'rank{r}').read_text().split() for r in range(3)]
    assert [e[0] for e in envs] == ['0', '1', '2'] and [e[1] for e in envs] == ['0', '1', '2']
    assert all(e[2] == '3' and e[3] == '3' and e[4] == '127.0.0.1' for e in envs) and len({e[5] for e in envs}) == 1
    assert subprocess.run([sys.executable, str(script), '2', 'fail'], timeout=30).returncode == 7


def _ddp_wrap_worker(port, out_q):
    from torch.nn.parallel import DistributedDataParallel
    from rdmnet_amd import config, model
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=0, world_size=1)
    net = model.create_model(config.make_cfg())
    ddp = DistributedDataParallel(net)  # what geotransformer/engine/base_tester.py:113 does with the reference's module
    out_q.put((ddp.module is net, len(list(ddp.parameters())), len(net.state_dict()), len(ddp.state_dict())))
    dist.destroy_process_group()


def test_module_is_wrappable_by_distributed_data_parallel():
    """DDP's constructor raises for a module without a parameter that requires a gradient; the module
    registers its parameters as the reference's does (requires_grad=True; forward runs under no_grad)."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_ddp_wrap_worker, args=(29641, q))
    p.start()
    same, n_par, n_keys, n_keys_ddp = q.get(timeout=300)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert same and n_keys == n_keys_ddp == 497 and n_par == 497 - 14  # 14 kernel_points buffers


@pytest.mark.parametrize('launcher', ['self', 'torchrun', 'self8'])
def test_bench_dry_run_exercises_the_multi_rank_half_without_a_gpu(launcher):
    """VERDICT r4 (next 6): `bench.py --gpus N --dry-run` builds the process group, runs the pre-flight -- barrier, ragged record
    gather, timing reduction: the calls every multi-rank run starts AND ends with -- and prints one line without running a
    pair.  With gloo it needs no GPU, so the launcher half of an 8-GPU driver run (both ways the driver may start it: the
    script spawning its own ranks, and torch.distributed.run) is executed here; on the GPU box the same flag runs it over RCCL."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bench = os.path.join(root, 'bench.py')
    world = 8 if launcher == 'self8' else 2  # (round 6: the driver's `--gpus 8` command line itself, eight ranks)
    tail = [bench, '--gpus', str(world), '--dry-run', '--dist-backend', 'gloo']
    if launcher.startswith('self'):
        cmd = [sys.executable] + tail
    else:
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
               '--master-port', '29671'] + tail
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'LOCAL_WORLD_SIZE')}
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [json.loads(line) for line in p.stdout.splitlines() if line.startswith('{')]
    assert len(lines) == 1  # rank 0 alone prints
    d = lines[0]
    assert d['dry_run'] and d['n_gpus'] == world and d['backend'] == 'gloo' and d['preflight']['ok']
    assert d['gpu_max_hw_queues'] == int(os.environ.get('GPU_MAX_HW_QUEUES', '8'))  # explicit in the ranks' environment
