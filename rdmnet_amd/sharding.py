"""Multi-GPU layout of the path: pairs are independent, so they are sharded rank-strided with no
data-path collective; one all_gather of fixed-size result records ends the run (SURVEY.md §8e;
replaces the reference's DistributedSampler + per-iteration scalar all_reduce,
geotransformer/utils/torch.py:16-21,58-60, engine/base_tester.py:123-128)."""
import torch


def pairs_for_rank(n_pairs, rank, world):
    """Pair ids of `rank`: r, r+W, ... (no padding by repetition, ragged tail allowed)."""
    return list(range(rank, n_pairs, world))


def gather_records(records, world, dist=None, force=False):
    """records: float32 [k, c] tensor of this rank (k may differ per rank).  Returns the list of all
    ranks' records on every rank (one all_gather of a padded block + the per-rank counts).
    force=True sends a world of ONE through the collective too (bench.py --force-dist: the RCCL path on one GPU)."""
    if dist is None or (world == 1 and not force):
        return [records]
    k = torch.tensor([records.shape[0]], dtype=torch.int64, device=records.device)
    counts = [torch.zeros_like(k) for _ in range(world)]
    dist.all_gather(counts, k)
    kmax = int(max(int(c) for c in counts))
    if kmax == 0:  # nothing to exchange (a collective on zero-size tensors is backend-dependent)
        return [records[:0].clone() for _ in range(world)]
    block = torch.zeros((kmax, records.shape[1]), dtype=records.dtype, device=records.device)
    block[:records.shape[0]] = records
    blocks = [torch.zeros_like(block) for _ in range(world)]
    dist.all_gather(blocks, block)
    return [b[:int(c)] for b, c in zip(blocks, counts)]


def reduce_timing(elapsed_s, latencies_ms, world, dist=None, device='cpu', force=False):
    """The timed region of a multi-rank run ends when the slowest rank ends: -> (max over ranks of elapsed_s, the
    per-pair latencies of all ranks concatenated in rank order).  Ranks may hold different numbers of latencies."""
    if dist is None or (world == 1 and not force):
        return float(elapsed_s), list(latencies_ms)
    tmax = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    lat = torch.tensor(list(latencies_ms), dtype=torch.float32, device=device).reshape(-1, 1)
    return float(tmax.item()), torch.cat(gather_records(lat, world, dist, force)).reshape(-1).cpu().tolist()


def summarize(all_records, rre_thresh=5.0, rte_thresh=2.0):
    """RR / mean RRE / mean RTE over successful pairs (experiments/eval.py:223-237 semantics).
    Record columns: [pair_id, rre_deg, rte_m, n_corr]."""
    rec = torch.cat(all_records).cpu()
    ok = (rec[:, 1] < rre_thresh) & (rec[:, 2] < rte_thresh)
    n_ok = int(ok.sum())
    return {'pairs': rec.shape[0], 'recall': n_ok / max(rec.shape[0], 1),
            'rre_deg': float(rec[ok, 1].mean()) if n_ok else float('nan'),
            'rte_m': float(rec[ok, 2].mean()) if n_ok else float('nan')}
