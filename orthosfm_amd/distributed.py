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


_pinned = {}


def pinned_array(name: str, rows: int, device="cpu"):
    """A (rows, 2) int32 numpy array backed by page-locked memory when the ranks
    drive GPUs (torch owns the allocation; the array is reused between calls):
    the matcher's device-to-host copy and the upload for the gather then run at
    full PCIe rate instead of through pageable staging."""
    import torch
    t = _pinned.get(name)
    if t is None or t.shape[0] < rows:
        pin = str(device) != "cpu" and torch.cuda.is_available()
        t = torch.empty((rows, 2), dtype=torch.int32, pin_memory=pin)
        _pinned[name] = t
    return t[:rows].numpy()


def gather_match_lists(local_counts, local_corr, num_pairs: int, rank: int, world: int, device="cpu"):
    """Gathers variable-length per-pair correspondence lists on rank 0.

    local_counts: int array, matches of each LOCAL pair (in shard order);
    local_corr:   (sum(local_counts), 2) int32, concatenated lists.
    Returns on rank 0 (counts[num_pairs], offsets[num_pairs + 1], corr[total, 2])
    in GLOBAL pair order, on other ranks None.

    Two collectives: an all_gather of the per-rank counts (to size the buffers)
    and one gather of the padded int32 payload.  On rank 0 the lists are put
    into global pair order ON THE DEVICE (one index_select over the gathered
    payload) and come to the host in one copy into a page-locked buffer.
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
    # header: per-pair counts padded to the largest shard
    head = torch.zeros(max_local, dtype=torch.int64)
    head[:n_local] = torch.from_numpy(local_counts)
    head = head.to(device)
    heads = [torch.zeros_like(head) for _ in range(world)]
    dist.all_gather(heads, head)
    heads = torch.stack(heads)                               # [world][max_local]
    totals = heads.sum(dim=1)
    width = max(int(totals.max().item()), 1)
    payload = torch.zeros((width, 2), dtype=torch.int32, device=device)
    if local_corr.size:
        payload[:local_corr.shape[0]].copy_(torch.from_numpy(local_corr), non_blocking=True)
    bufs = [torch.empty_like(payload) for _ in range(world)] if rank == 0 else None
    dist.gather(payload, bufs, dst=0)
    if rank != 0:
        return None
    return assemble_global_order(heads, bufs, num_pairs, world, device)


def assemble_global_order(heads, bufs, num_pairs: int, world: int, device="cpu"):
    """Rank 0's part after the gather: heads [world][max_local] per-pair counts of
    every rank, bufs[r] the (width, 2) payload of rank r.  Global pair
    gi = k * world + r is local pair k of rank r (shard_pairs)."""
    import torch
    width = bufs[0].shape[0]
    r_of = torch.arange(num_pairs, device=device) % world
    k_of = torch.arange(num_pairs, device=device) // world
    counts = heads[r_of, k_of]                               # [num_pairs], global order
    offsets = torch.zeros(num_pairs + 1, dtype=torch.int64, device=device)
    offsets[1:] = torch.cumsum(counts, 0)
    local_off = torch.cumsum(heads, 1) - heads               # start of local pair k inside rank r's payload
    src_start = r_of * width + local_off[r_of, k_of]         # into the rank-major concatenation
    total = int(offsets[-1].item())
    allbuf = torch.cat(bufs, 0)                              # [world * width][2]
    pair_of = torch.repeat_interleave(torch.arange(num_pairs, device=device), counts, output_size=total)
    src = src_start[pair_of] + (torch.arange(total, device=device) - offsets[:-1][pair_of])
    ordered = allbuf.index_select(0, src)
    out = pinned_array("gathered", max(total, 1), device)
    torch.from_numpy(out)[:total].copy_(ordered)             # one device-to-host copy
    return counts.cpu().numpy(), offsets.cpu().numpy(), out[:total]


def max_over_ranks(value: float, world: int, device="cpu") -> float:
    if world == 1:
        return value
    import torch
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
