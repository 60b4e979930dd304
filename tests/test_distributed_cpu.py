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


def test_lpt_deal_balances_unequal_views():
    from orthosfm_amd import distributed as D
    sizes = np.array([20000, 500, 12000, 300, 18000, 7000, 100, 15000, 9000, 2500])
    pairs = [(a, b) for a in range(1, 10) for b in range(a)]
    for world in (2, 3, 8):
        shards = D.deal_pairs(pairs, sizes, world)
        assert sorted(np.concatenate(shards).tolist()) == list(range(len(pairs)))
        assert all(np.all(np.diff(s) > 0) for s in shards)
        w = np.array([sizes[a] * sizes[b] for a, b in pairs], dtype=np.int64)
        loads = np.array([w[s].sum() for s in shards])
        assert loads.max() <= loads.mean() + w.max()              # the LPT bound
        rr = np.array([w[r::world].sum() for r in range(world)])
        assert loads.max() <= rr.max()
    # equal sizes: exactly the round-robin deal
    eq = D.deal_pairs(pairs, np.full(10, 777), 4)
    assert all(np.array_equal(eq[r], np.arange(r, len(pairs), 4)) for r in range(4))


def _uneven_worker(rank, world, port, how, out_dir):
    import torch.distributed as dist
    import track_cases
    from orthosfm_amd import capi, distributed as D, tracks as T
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m = track_cases.random_matching(num_views=9, feats_per_view=150, num_scene_points=260, p_false=0.1, seed=17)
        num_pairs = m["pairs"].shape[0]
        lists = [m["corr"][m["pair_offsets"][i]:m["pair_offsets"][i + 1]] for i in range(num_pairs)]
        # the deal by work with unequal (pretended) view sizes: the ranks get DIFFERENT numbers of pairs
        weights = np.array([30000, 200, 25000, 400, 100, 18000, 900, 22000, 50])
        shards = D.deal_pairs([tuple(p) for p in m["pairs"]], weights, world)
        assert len({len(s) for s in shards}) > 1
        mine = shards[rank]
        counts = np.array([lists[i].shape[0] for i in mine], dtype=np.int64)
        packed = np.concatenate([lists[i] for i in mine] + [np.zeros((0, 2), np.int32)], axis=0).astype(np.int32)
        pairs = (capi.Pair * num_pairs)()
        for i, (a, b) in enumerate(m["pairs"]):
            pairs[i].view_1, pairs[i].view_2 = int(a), int(b)
        if how == "shm":
            store = D.SharedMatchStore(max(int(sum(lists[i].shape[0] for i in s)) for s in shards), rank, world)
            store.slice[:packed.shape[0]] = packed
            res = store.collect(counts, num_pairs, shards)
            if rank == 0:
                c, starts, corr = res
                out = T.compute_flat_ranges(m["view_sizes"], m["colors"], pairs, starts, c, np.asarray(corr))
            dist.barrier()
            store.close()
        else:
            res = D.gather_match_lists(counts, packed, num_pairs, rank, world, shards=shards)
            if rank == 0:
                c, offs, corr = res
                out = T.compute_flat(m["view_sizes"], m["colors"], pairs, offs, np.ascontiguousarray(corr))
        if rank == 0:
            single = T.compute_flat(m["view_sizes"], m["colors"], pairs, m["pair_offsets"], m["corr"])
            for a, b in zip(out[:4], single[:4]):
                assert np.array_equal(a, b)
            assert single[4].num_tracks > 50
            open(os.path.join(out_dir, f"ok_{how}"), "w").write(str(single[4].num_tracks))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("how", ["shm", "rccl"])
def test_uneven_shards_give_the_single_rank_tracks(tmp_path, how):
    """Two ranks with DIFFERENT shard sizes (the LPT deal over unequal views): the lists
    travel through the shared segment (osfm_tracks_compute_ranges reads them in place) or
    the gather, and rank 0 builds exactly the tracks a single rank builds."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_uneven_worker, args=(2, port, how, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(os.path.join(str(tmp_path), f"ok_{how}"))
