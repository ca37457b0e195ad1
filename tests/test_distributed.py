"""CPU, world size 2, gloo: the multi-GPU layout (rank-strided pair sharding + one record gather)."""
import os

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
    if rank == 0:
        out_q.put([r.tolist() for r in allr])
    dist.barrier()
    dist.destroy_process_group()


def test_rank_strided_sharding_and_gather_world2():
    n_pairs, world = 7, 2  # ragged: rank 0 gets 4 pairs, rank 1 gets 3 -- no padding by repetition
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, 29617, n_pairs, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [len(g) for g in got] == [4, 3]
    ids = sorted(int(r[0]) for g in got for r in g)
    assert ids == list(range(n_pairs))  # every pair exactly once
    summary = sharding.summarize([torch.tensor(g) for g in got])
    assert summary['pairs'] == n_pairs and summary['recall'] == 1.0


def test_single_rank_is_identity():
    rec = torch.ones(3, 4)
    assert sharding.gather_records(rec, 1)[0] is rec
    assert sharding.pairs_for_rank(5, 0, 1) == [0, 1, 2, 3, 4]
