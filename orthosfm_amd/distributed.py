"""Multi-GPU form of the matching path: view pairs are independent units, so
they are dealt to the ranks up front (no data-path collective) and only the
per-pair match lists are brought to rank 0 for track building -- the one
exchange step of the path: in place through a shared host segment
on one node (SharedMatchStore), as an RCCL gather across nodes
(gather_match_lists).  One process per GPU, torch.distributed
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


def deal_pairs(pairs, view_sizes, world: int):
    """Static deal of the pair list by work: the cost of pair (a, b) is N_a * N_b, known up
    front, so the pairs are handed out longest first to the rank with the
    least work so far (LPT; ties: the lower rank).  Returns one ascending array of pair
    indices per rank -- the same on every rank, no communication.  With equal view sizes
    this is exactly the round-robin deal of shard_pairs."""
    n = len(pairs)
    if world <= 1:
        return [np.arange(n, dtype=np.int64)]
    sizes = np.asarray(view_sizes, dtype=np.int64)
    pa = np.asarray(pairs, dtype=np.int64).reshape(-1, 2)
    w = sizes[pa[:, 0]] * sizes[pa[:, 1]]
    if n == 0 or np.all(w == w[0]):
        return [np.arange(r, n, world, dtype=np.int64) for r in range(world)]
    import heapq
    order = np.argsort(-w, kind="stable")
    heap = [(0, r) for r in range(world)]
    owner = np.empty(n, dtype=np.int64)
    for i in order:
        load, r = heapq.heappop(heap)
        owner[i] = r
        heapq.heappush(heap, (load + int(w[i]), r))
    return [np.nonzero(owner == r)[0].astype(np.int64) for r in range(world)]


def _shards_or_round_robin(shards, num_pairs, world):
    if shards is None:
        return [np.arange(r, num_pairs, world, dtype=np.int64) for r in range(world)]
    assert len(shards) == world and sum(len(s) for s in shards) == num_pairs
    return [np.asarray(s, dtype=np.int64) for s in shards]


def _owner_maps(shards, num_pairs):
    """r_of[gi], k_of[gi]: rank that owns global pair gi and its position in that shard."""
    r_of = np.empty(num_pairs, dtype=np.int64)
    k_of = np.empty(num_pairs, dtype=np.int64)
    for r, s in enumerate(shards):
        r_of[s] = r
        k_of[s] = np.arange(len(s))
    return r_of, k_of


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


def gather_match_lists(local_counts, local_corr, num_pairs: int, rank: int, world: int, device="cpu",
                       shards=None):
    """Gathers variable-length per-pair correspondence lists on rank 0.

    local_counts: int array, matches of each LOCAL pair (in shard order);
    local_corr:   (sum(local_counts), 2) int32, concatenated lists.
    Returns on rank 0 (counts[num_pairs], offsets[num_pairs + 1], corr[total, 2])
    in GLOBAL pair order, on other ranks None.  shards: the deal (deal_pairs), one index
    array per rank; None = round robin.

    Two collectives: an all_gather of the per-rank counts (to size the buffers)
    and one gather of the padded int32 payload.  On rank 0 the lists are put
    into global pair order ON THE DEVICE (one index_select over the gathered
    payload) and come to the host in one copy into a page-locked buffer.
    """
    import torch
    import torch.distributed as dist

    local_counts = np.ascontiguousarray(local_counts, dtype=np.int64)
    local_corr = np.ascontiguousarray(local_corr, dtype=np.int32).reshape(-1, 2)
    shards = _shards_or_round_robin(shards, num_pairs, world)
    n_local = len(shards[rank])
    assert local_counts.shape[0] == n_local and local_corr.shape[0] == int(local_counts.sum())
    if world == 1:
        offs = np.concatenate([[0], np.cumsum(local_counts)])
        return local_counts, offs, local_corr

    max_local = max(len(s) for s in shards)
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
    return assemble_global_order(heads, bufs, num_pairs, world, device, shards)


def assemble_global_order(heads, bufs, num_pairs: int, world: int, device="cpu", shards=None):
    """Rank 0's part after the gather: heads [world][max_local] per-pair counts of
    every rank, bufs[r] the (width, 2) payload of rank r.  Global pair gi is local pair
    k_of[gi] of rank r_of[gi] (the deal; round robin when shards is None)."""
    import torch
    width = bufs[0].shape[0]
    r_np, k_np = _owner_maps(_shards_or_round_robin(shards, num_pairs, world), num_pairs)
    r_of = torch.from_numpy(r_np).to(device)
    k_of = torch.from_numpy(k_np).to(device)
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


class SharedSegmentUnavailable(RuntimeError):
    """Raised on EVERY rank when rank 0 cannot create the segment; callers fall
    back to gather_match_lists."""


class SharedMatchStore:
    """The exchange step on ONE node without moving the lists twice.

    The consumer of the match lists is host code on rank 0 (RANSAC bookkeeping,
    track building), and every GPU of a node has its own PCIe link to the host.
    So each rank lets the matcher write its lists straight into its slice of
    one POSIX shared-memory segment (page-locked in the owning process), and
    rank 0 reads them where they lie: 8 device-to-host copies in parallel
    instead of a gather over xGMI followed by one 8x larger copy through rank
    0's link.  The only collective left is the all_gather of the per-pair
    counts (RCCL), which also orders the slices' contents before rank 0 reads.

    Across nodes use gather_match_lists (payload over RCCL) instead.
    """

    def __init__(self, rows_per_rank: int, rank: int, world: int, device="cpu"):
        import os
        import torch
        import torch.distributed as dist
        # whole pages per slice (512 rows of 8 bytes), so every rank page-locks only its own
        self.rank, self.world, self.rows = rank, world, (int(max(rows_per_rank, 1)) + 511) // 512 * 512
        self.device = device
        self._registered = None
        name = [None, self.rows]
        if rank == 0:
            need = self.world * self.rows * 8
            st = os.statvfs("/dev/shm")
            if st.f_bavail * st.f_frsize >= need + (64 << 20):     # a tmpfs write past its size is a SIGBUS
                name[0] = f"/dev/shm/osfm_matches_{os.getpid()}_{id(self) & 0xffffff:x}"
                with open(name[0], "wb") as f:
                    f.truncate(need)
        if world > 1:
            dist.broadcast_object_list(name, src=0)
        if name[0] is None:
            raise SharedSegmentUnavailable(f"/dev/shm cannot hold {self.world * self.rows * 8} bytes")
        if name[1] != self.rows:
            raise ValueError(f"SharedMatchStore: rank {rank} asks for {self.rows} rows per rank, rank 0 for {name[1]}")
        self.path = name[0]
        self.array = np.memmap(self.path, dtype=np.int32, mode="r+", shape=(self.world * self.rows, 2))
        if world > 1:
            dist.barrier()
        if rank == 0:
            os.unlink(self.path)          # the mappings keep the segment alive; nothing is left behind
        self.slice = self.array[rank * self.rows:(rank + 1) * self.rows]
        ok = 1
        if str(device) != "cpu" and torch.cuda.is_available():
            # page-lock the own slice so that the matcher's copy runs at PCIe rate
            rc = torch.cuda.cudart().cudaHostRegister(self.slice.ctypes.data, self.slice.nbytes, 0)
            if int(rc) != 0:
                ok = 0
            else:
                self._registered = self.slice.ctypes.data
        # a failure on one rank is everybody's: the ranks agree (one all_reduce) and ALL of
        # them raise SharedSegmentUnavailable, so no rank is left waiting in a collective
        if world > 1:
            flag = torch.tensor([ok], dtype=torch.int32, device=device if str(device) != "cpu" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = int(flag.item())
        if not ok:
            self.close()
            raise SharedSegmentUnavailable("hipHostRegister of a result slice failed on some rank")

    def collect(self, local_counts, num_pairs: int, shards=None):
        """After the rank's lists are in its slice (packed in shard order): returns on
        rank 0 (counts[num_pairs], starts[num_pairs], corr) with pair gi's list at
        corr[starts[gi] : starts[gi] + counts[gi]], global pair order; None elsewhere.
        shards: the deal (deal_pairs); None = round robin."""
        import torch
        import torch.distributed as dist
        local_counts = np.ascontiguousarray(local_counts, dtype=np.int64)
        shards = _shards_or_round_robin(shards, num_pairs, self.world)
        n_local = len(shards[self.rank])
        assert local_counts.shape[0] == n_local and int(local_counts.sum()) <= self.rows
        max_local = max(len(s) for s in shards)
        head = torch.zeros(max_local, dtype=torch.int64)
        head[:n_local] = torch.from_numpy(local_counts)
        if self.world > 1:
            head = head.to(self.device)
            heads = [torch.zeros_like(head) for _ in range(self.world)]
            dist.all_gather(heads, head)           # every rank's lists are complete once this returns
            heads = torch.stack(heads).cpu().numpy()
        else:
            heads = head.numpy()[None]
        if self.rank != 0:
            return None
        r_of, k_of = _owner_maps(shards, num_pairs)
        local_off = np.cumsum(heads, axis=1) - heads
        counts = heads[r_of, k_of]
        starts = r_of.astype(np.int64) * self.rows + local_off[r_of, k_of]
        return counts, starts, self.array

    def close(self):
        if self._registered is not None:
            import torch
            torch.cuda.cudart().cudaHostUnregister(self._registered)
            self._registered = None
        self.slice = None
        self.array = None


def max_over_ranks(value: float, world: int, device="cpu") -> float:
    if world == 1:
        return value
    import torch
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
