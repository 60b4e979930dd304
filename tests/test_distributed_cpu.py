"""world_size-2 (and 3) gloo tests of the N>1 path on CPU: the pair deal and
the gather of variable-length match lists on rank 0."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_matches(pair_index):
    """Deterministic stand-in for one pair's correspondence list."""
    r = np.random.default_rng(1000 + pair_index)
    n = int(r.integers(0, 40)) if pair_index % 5 else 0
    return np.stack([np.sort(r.choice(5000, n, replace=False)), r.integers(0, 5000, n)], axis=1).astype(np.int32)


def _worker(rank, world, port, num_pairs, out_dir):
    import torch.distributed as dist
    from orthosfm_amd import distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pairs = list(range(num_pairs))
        mine = D.shard_pairs(pairs, rank, world)
        assert mine == list(range(rank, num_pairs, world))
        lists = [_fake_matches(i) for i in mine]
        counts = np.array([l.shape[0] for l in lists], dtype=np.int64)
        corr = np.concatenate(lists + [np.zeros((0, 2), np.int32)], axis=0)
        res = D.gather_match_lists(counts, corr, num_pairs, rank, world)
        t = D.max_over_ranks(float(rank + 1), world)
        assert t == float(world)
        if rank == 0:
            c, offs, allc = res
            np.savez(os.path.join(out_dir, "gathered.npz"), c=c, offs=offs, corr=allc)
        else:
            assert res is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,num_pairs", [(2, 21), (3, 10), (2, 1)])
def test_pair_shard_and_gather_gloo(tmp_path, world, num_pairs):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(world, port, num_pairs, str(tmp_path)), nprocs=world, join=True)
    g = np.load(os.path.join(str(tmp_path), "gathered.npz"))
    exp = [_fake_matches(i) for i in range(num_pairs)]
    assert np.array_equal(g["c"], [e.shape[0] for e in exp])
    assert np.array_equal(g["corr"], np.concatenate(exp + [np.zeros((0, 2), np.int32)], axis=0))
    assert g["offs"][-1] == sum(e.shape[0] for e in exp)


def test_shard_is_a_partition():
    from orthosfm_amd import distributed as D
    pairs = [(a, b) for a in range(1, 30) for b in range(a)]
    for world in (1, 2, 4, 8):
        parts = [D.shard_pairs(pairs, r, world) for r in range(world)]
        assert sorted(sum(parts, [])) == sorted(pairs)
        sizes = [len(p) for p in parts]
        assert max(sizes) - min(sizes) <= 1
        for i in range(len(pairs)):
            assert pairs[i] in parts[D.owner_of(i, world)]


def _store_worker(rank, world, port, num_pairs, out_dir):
    import torch.distributed as dist
    from orthosfm_amd import distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = D.shard_pairs(list(range(num_pairs)), rank, world)
        store = D.SharedMatchStore(64 * len(range(0, num_pairs, world)), rank, world)   # same on every rank
        for step in range(2):             # the segment is reused between passes
            lists = [_fake_matches(i + 7 * step) for i in mine]
            counts = np.array([l.shape[0] for l in lists], dtype=np.int64)
            n = int(counts.sum())
            store.slice[:n] = np.concatenate(lists + [np.zeros((0, 2), np.int32)], axis=0)
            res = store.collect(counts, num_pairs)
            if rank == 0:
                c, starts, corr = res
                got = [np.array(corr[starts[i]:starts[i] + c[i]]) for i in range(num_pairs)]
                np.savez(os.path.join(out_dir, f"store{step}.npz"), c=c, starts=starts,
                         **{f"l{i}": g for i, g in enumerate(got)})
            else:
                assert res is None
            dist.barrier()                # nobody overwrites a slice rank 0 still reads
        assert not os.path.exists(store.path)      # unlinked once every rank had mapped it
        store.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,num_pairs", [(2, 21), (3, 10), (1, 4)])
def test_shared_match_store_gloo(tmp_path, world, num_pairs):
    """The single-node exchange: lists stay in the per-rank slices of one shared
    segment, rank 0 gets (counts, starts) in global pair order."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_store_worker, args=(world, port, num_pairs, str(tmp_path)), nprocs=world, join=True)
    for step in range(2):
        g = np.load(os.path.join(str(tmp_path), f"store{step}.npz"))
        for i in range(num_pairs):
            exp = _fake_matches(i + 7 * step)
            assert g["c"][i] == exp.shape[0]
            assert np.array_equal(g[f"l{i}"], exp)
        # slices do not overlap: rank r's lists lie inside [r * rows, (r + 1) * rows)
        rows = (64 * len(range(0, num_pairs, world)) + 511) // 512 * 512      # slices are whole pages
        r_of = np.arange(num_pairs) % world
        assert np.all(g["starts"] >= r_of * rows) and np.all(g["starts"] + g["c"] <= (r_of + 1) * rows)
