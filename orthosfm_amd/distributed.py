"""Multi-GPU form of the matching path: view pairs are independent units, so
they are dealt to the ranks up front (no data-path collective) and only the
per-pair match lists are gathered on rank 0 for track building -- the one
exchange step of the path.  One process per GPU, torch.distributed
("nccl" = RCCL over xGMI on ROCm; "gloo" on CPU for the tests).

The reference has no distributed runtime at all (its pair loop is an OpenMP
`parallel for schedule(dynamic)`, src/mve/sfm/bundler_matching.cc:74); the
static deal below replaces that loop across GPUs.
"""
from __future__ import annotations

import numpy as np


def shard_pairs(pairs, rank: int, world: int):
    """Round-robin deal of the pair list: pair i goes to rank i % world.  Work
    per pair is N1*N2 and the triangular enumeration interleaves large and
    small view ids, so the deal is balanced without communication."""
    return list(pairs[rank::world])


def owner_of(pair_index: int, world: int) -> int:
    return pair_index % world


def gather_match_lists(local_counts, local_corr, num_pairs: int, rank: int, world: int, device="cpu"):
    """Gathers variable-length per-pair correspondence lists on rank 0.

    local_counts: int array, matches of each LOCAL pair (in shard order);
    local_corr:   (sum(local_counts), 2) int32, concatenated lists.
    Returns on rank 0 (counts[num_pairs], offsets[num_pairs + 1], corr[total, 2])
    in GLOBAL pair order, on other ranks None.

    Two collectives: an all_gather of the per-rank totals (to size the
    buffers) and one gather of the padded int32 payload.
    """
    import torch
    import torch.distributed as dist

    local_counts = np.ascontiguousarray(local_counts, dtype=np.int64)
    local_corr = np.ascontiguousarray(local_corr, dtype=np.int32).reshape(-1, 2)
    n_local = len(range(rank, num_pairs, world))
    assert local_counts.shape[0] == n_local and local_corr.shape[0] == int(local_counts.sum())
    if world == 1:
        offs = np.concatenate([[0], np.cumsum(local_counts)])
        return local_counts, offs, local_corr

    max_local = len(range(0, num_pairs, world))
    # header: per-pair counts padded to the largest shard, then the total
    head = torch.zeros(max_local + 1, dtype=torch.int64, device=device)
    head[:n_local] = torch.from_numpy(local_counts).to(device)
    head[max_local] = int(local_counts.sum())
    heads = [torch.zeros_like(head) for _ in range(world)]
    dist.all_gather(heads, head)
    totals = [int(h[max_local].item()) for h in heads]
    width = max(max(totals), 1)
    payload = torch.zeros(2 * width, dtype=torch.int32, device=device)
    if local_corr.size:
        payload[:local_corr.size] = torch.from_numpy(local_corr.reshape(-1)).to(device)
    bufs = [torch.zeros_like(payload) for _ in range(world)] if rank == 0 else None
    dist.gather(payload, bufs, dst=0)
    if rank != 0:
        return None
    counts = np.zeros(num_pairs, dtype=np.int64)
    per_rank_counts = []
    for r in range(world):
        n_r = len(range(r, num_pairs, world))
        c = heads[r][:n_r].cpu().numpy()
        per_rank_counts.append(c)
        counts[r::world] = c
    offsets = np.concatenate([[0], np.cumsum(counts)])
    corr = np.zeros((int(offsets[-1]), 2), dtype=np.int32)
    for r in range(world):
        data = bufs[r].cpu().numpy()[:2 * totals[r]].reshape(-1, 2)
        pos = 0
        for k, gi in enumerate(range(r, num_pairs, world)):
            n = int(per_rank_counts[r][k])
            corr[offsets[gi]:offsets[gi] + n] = data[pos:pos + n]
            pos += n
    return counts, offsets, corr


def max_over_ranks(value: float, world: int, device="cpu") -> float:
    if world == 1:
        return value
    import torch
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
