#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X backend (driver contract).

One "step" = one pass of the hot path over one synthetic image set:
  * matching: exhaustive pairwise matching (low-res gate, two-way SIFT match,
    cross-check, thresholds, ordered correspondence lists) of ALL V(V-1)/2
    pairs of a V-image set with ~20k SIFT features per image, descriptors
    resident in HBM (BASELINE.json configs[1]: 50 images on 1 GPU);
  * bundle adjustment (reported beside it): LM iterations/s of the global BA
    of configs[3] (200 quaternion cameras, 100k tracks) -- see `ba` in the
    JSON line.
`value` = image pairs matched per second over all ranks (pairs are sharded
across ranks with no data-path collective; the match lists reach rank 0 at the
end of every step, inside the timed region: on one node every rank's lists go
straight into its slice of a shared page-locked host segment and only the
per-pair counts are all-gathered over RCCL -- `--exchange rccl` gathers the
lists themselves over xGMI instead, as a multi-node run would).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))



def usable_cores():
    """Host cores this process may really use (affinity mask and cgroup CPU
    quota) -- the GPU boxes expose 256 hardware threads but grant a share."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("OSFM_BENCH_MAX_CORES", "64"))))


# the CPU oracle is OpenMP code: size its team before libgomp is loaded
os.environ.setdefault("OMP_NUM_THREADS", str(usable_cores()))

I8_MFMA_PEAK_TOPS = 5000.0      # dense int8 MFMA, MI355X_MICROARCH.md (2x the 2.5 PF bf16 figure)
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--views", type=int, default=0, help="0 = 50 at N=1, grown with N so pairs per GPU stay ~1225")
    ap.add_argument("--features", type=int, default=20000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ba", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the extra pass with RANSAC-F")
    ap.add_argument("--cpu-sample-pairs", type=int, default=0, help="0 = auto (about 15 s of CPU work)")
    ap.add_argument("--no-lowres-gate", action="store_true",
                    help="kernel experiments only: match every pair in full whatever the results are")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo for rehearsal)")
    ap.add_argument("--exchange", default="shm", choices=["shm", "rccl"],
                    help="how the match lists reach rank 0 for N > 1: shared host segment (one node) or RCCL gather")
    return ap.parse_args()


def exchanged_tracks(gathered, all_pairs, V, F, how):
    """Rank 0 after an N > 1 pass: track building over the lists of ALL ranks, read
    in place from the shared segment (shm) or from the gathered buffer (rccl)."""
    from orthosfm_amd import capi, tracks as T
    pairs = (capi.Pair * len(all_pairs))()
    for i, (a, b) in enumerate(all_pairs):
        pairs[i].view_1, pairs[i].view_2 = a, b
    sizes = np.full(V, F, dtype=np.int32)
    t0 = time.perf_counter()
    if how == "shm":
        counts, starts, corr = gathered
        res = T.compute_flat_ranges(sizes, None, pairs, starts, counts, corr)
    else:
        counts, offs, corr = gathered
        res = T.compute_flat(sizes, None, pairs, offs, np.ascontiguousarray(corr))
    ms = (time.perf_counter() - t0) * 1e3
    return {"how": "shared host segment, lists consumed in place" if how == "shm" else "RCCL gather to rank 0",
            "correspondences_all_ranks": int(np.sum(counts)), "tracks": int(res[4].num_tracks),
            "invalid_tracks": int(res[4].num_invalid_tracks), "track_building_ms": ms}


def cpu_baseline(iset, pairs, n_sample):
    """Times the CPU oracle (bit-exact restatement of the reference matcher,
    OpenMP over queries) on a bounded sample of the same workload."""
    import oracle_lib
    threads = oracle_lib.oracle().oracle_num_threads()
    empty = np.zeros((0, 64), np.int16)
    # calibrate on one quarter-size pair to pick the sample size
    a, b = pairs[0]
    t0 = time.perf_counter()
    oracle_lib.oracle_pairwise_match(iset.sift[a][:5000], empty, iset.sift[b][:5000], empty)
    t_quarter = time.perf_counter() - t0
    est_pair = t_quarter * 16.0
    if n_sample <= 0:
        n_sample = int(max(1, min(len(pairs), round(15.0 / max(est_pair, 1e-3)))))
    t0 = time.perf_counter()
    for (a, b) in pairs[:n_sample]:
        oracle_lib.oracle_pairwise_match(iset.sift[a], empty, iset.sift[b], empty)
    dt = time.perf_counter() - t0
    return {"value": n_sample / dt, "unit": "pairs/s", "cores": int(threads), "kind": "port",
            "sample": f"{n_sample} full-size pairs ({iset.sift[0].shape[0]} x {iset.sift[0].shape[0]} SIFT, "
                      f"two-way + cross-check) of the same image set, {dt:.1f} s"}


def pmc_traffic_per_launch(pairs_per_launch):
    """HBM bytes of one launch of the dominant kernel from the committed PMC
    passes (profiles/r01_match_traffic_pmc.json: FETCH_SIZE and WRITE_SIZE
    collected in separate rocprofv3 --pmc runs of this same command, FETCH_SIZE
    doubled as the gfx950 guide prescribes), scaled to the pairs per launch."""
    path = os.path.join(ROOT, "profiles", "r01_match_traffic_pmc.json")
    try:
        rec = json.load(open(path))
        return rec["hbm_bytes_per_pair"] * pairs_per_launch
    except Exception:
        return None


def ba_cpu_baseline(iterations=2):
    """The BA oracle (double-precision restatement of the reference's Ceres
    solve, parity unpinned) on the same config-4 scene, bounded to a few LM
    iterations."""
    import oracle_lib
    from orthosfm_amd import synth
    sc = synth.make_ba_scene(synth.MODEL_QUATERNION, 200, 100000, config_id=4)
    t0 = time.perf_counter()
    s = oracle_lib.oracle_ba_solve(sc, max_num_iterations=iterations)
    dt = time.perf_counter() - t0
    return {"value": s.num_iterations / dt, "unit": "LM iterations/s",
            "cores": int(oracle_lib.oracle().oracle_num_threads()), "kind": "port",
            "sample": f"{s.num_iterations} LM iterations of the same 200-camera / 100k-track problem, {dt:.1f} s "
                      "(residual/Jacobian evaluation and the dense Cholesky trailing updates OpenMP-parallel, Schur accumulation serial)"}


def cascade_bench(iset, V, pairs, capacity, device_index, with_cpu):
    """The same pairs through the cascade-hashing mode (sfm::CascadeHashing, the
    application's default, approximate matcher).  CPU baseline: the reference's
    own cascade_hashing.cc when oracle/_ref travelled (its pairwise_match on a
    sample of pairs, all host threads via its OpenMP init, matching single
    threaded per pair as in one iteration of bundler::Matching::compute's loop)."""
    from orthosfm_amd.matching import HipCascadeHashing
    m = HipCascadeHashing(V, device=device_index, copy_results=False)
    for v in range(V):
        m.set_view(v, iset.sift[v])
    t0 = time.perf_counter()
    m.cascade_hashes(0, 0)
    init_s = time.perf_counter() - t0
    m.compute(pairs, capacity=capacity)
    t0 = time.perf_counter()
    out = m.compute(pairs, capacity=capacity)
    dt = time.perf_counter() - t0
    st = m.stats()
    res = {"workload": f"{len(pairs)} pairs, cascade hashing (6 bucket groups x 256 buckets, 6..10 candidates)",
           "pairs_per_s": len(pairs) / dt, "ms_per_step": dt * 1e3, "hash_init_ms": init_s * 1e3,
           "kernel_ms": st.cashash_kernel_ms,
           "correspondences": int(sum(tv.num_matches for tv in out if tv.status == capi_mod().PAIR_MATCHED))}
    m.close()
    if with_cpu:
        import oracle_lib
        if oracle_lib.ref_cashash() is not None:
            nv = min(V, 6)
            empty = [np.zeros((0, 64), np.int16)] * nv
            t0 = time.perf_counter()
            ref = oracle_lib.RefCasHash(iset.sift[:nv], empty)
            t_init = time.perf_counter() - t0
            sample = [(a, b) for a in range(nv) for b in range(a)]
            t0 = time.perf_counter()
            for a, b in sample:
                ref.pairwise_match(a, b)
            dtc = time.perf_counter() - t0
            ref.close()
            res["cpu_baseline"] = {"value": len(sample) / dtc, "unit": "pairs/s", "cores": 1, "kind": "reference",
                                   "sample": f"{len(sample)} pairs of the first {nv} views, {dtc:.1f} s "
                                             f"(hash init of those views {t_init:.2f} s not included)"}
    return res


def capi_mod():
    from orthosfm_amd import capi
    return capi


def tracks_bench(out, V, F, with_cpu):
    """Tracks::compute (bundler_tracks.cc:49-145) over the match lists this run
    produced: host code in the reference and here; the CPU baseline is the
    reference's own file when oracle/_ref travelled, the oracle port otherwise."""
    from orthosfm_amd import tracks as T
    matched = [tv for tv in out if tv.matches.shape[0] > 0]
    sizes = np.full(V, F, dtype=np.int32)
    pairs, offsets, corr = T.flatten_matching(matched)
    T.compute_flat(sizes, None, pairs, offsets, corr)
    t0 = time.perf_counter()
    ids, toff, tfeat, tcol, summary = T.compute_flat(sizes, None, pairs, offsets, corr)
    dt = time.perf_counter() - t0
    res = {"workload": f"Tracks::compute over {len(matched)} pairs / {int(offsets[-1])} matches (host code)",
           "ms": dt * 1e3, "tracks": int(summary.num_tracks), "invalid_tracks": int(summary.num_invalid_tracks),
           "track_features": int(summary.num_features)}
    if with_cpu:
        import oracle_lib
        parr = np.array([(tv.view_1_id, tv.view_2_id) for tv in matched], dtype=np.int32).reshape(-1, 2)
        kind = "reference" if oracle_lib.ref_tracks() is not None else "port"
        fn = oracle_lib.ref_tracks_compute if kind == "reference" else oracle_lib.oracle_tracks
        t0 = time.perf_counter()
        ref = fn(sizes, None, parr, offsets, corr)
        dtc = time.perf_counter() - t0
        res["cpu_baseline"] = {"value": dtc * 1e3, "unit": "ms", "cores": 1, "kind": kind,
                               "sample": "the same match lists, whole job",
                               "identical_output": bool(np.array_equal(ref["track_features"], tfeat)
                                                        and np.array_equal(ref["track_ids"], ids))}
    return res


def outlier_filter_bench(device, with_cpu):
    """filterOutlierTracks on the 100k points of the global-BA config: the
    O(P^2) nearest-neighbour search is the kernel, the rest is O(P) host work."""
    from orthosfm_amd import filters, synth
    sc = synth.make_ba_scene(synth.MODEL_QUATERNION, 200, 100000, config_id=4)
    has = np.ones(sc.points.shape[0], dtype=bool)
    filters.outlier_track_flags(sc.points, has, device)
    t0 = time.perf_counter()
    keep, st = filters.outlier_track_flags(sc.points, has, device)
    dt = time.perf_counter() - t0
    out = {"workload": "filterOutlierTracks, 100000 points (1e10 distance evaluations, f64)",
           "ms": dt * 1e3, "kept": int(keep.sum()), "mean_nn": st.mean, "sigma": st.sigma}
    if with_cpu:
        import oracle_lib
        n = 20000
        t0 = time.perf_counter()
        oracle_lib.oracle_nn_distances(sc.points[:n])
        dtc = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": dtc * 1e3 * (sc.points.shape[0] / n) ** 2, "unit": "ms (extrapolated to 100000 points)",
                               "cores": int(oracle_lib.oracle().oracle_num_threads()), "kind": "port",
                               "sample": f"nearest-neighbour search over the first {n} points, {dtc:.2f} s, scaled by (100000/{n})^2"}
    return out


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    torch = None
    if world > 1:
        # torch first: it brings its own ROCm runtime, which libosfm_hip.so then shares
        import torch
        import torch.distributed as dist
    from orthosfm_amd import capi, synth
    from orthosfm_amd import distributed as D
    from orthosfm_amd.matching import HipExhaustiveMatching

    ndev = capi.device_count()
    if ndev < 1:
        raise RuntimeError("bench.py: no HIP device (the backend has no CPU fallback)")
    device_index = local_rank % ndev
    tdev = "cpu"
    if world > 1:
        if args.backend == "nccl":
            torch.cuda.set_device(device_index)
            tdev = torch.device("cuda", device_index)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=tdev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    # weak scaling: the image set grows with the rank count so that every
    # GPU keeps ~1225 pairs (50 views -> 1225; 71 -> 2485; 100 -> 4950; 141 -> 9870)
    V = args.views
    if V <= 0:
        V = 50
        while V * (V - 1) // 2 < 1225 * world:
            V += 1
    F = args.features
    iset = synth.make_image_set(V, F, config_id=2)
    all_pairs = [capi.pair_from_index(i) for i in range(V * (V - 1) // 2)]
    my_pairs = D.shard_pairs(all_pairs, rank, world)

    o1 = capi.default_match_options()
    if args.no_lowres_gate:
        o1.use_lowres_matching = 0
        o1.min_feature_matches = 0
    m = HipExhaustiveMatching(V, device=device_index, options=o1, copy_results=False)
    t0 = time.perf_counter()
    for v in range(V):
        m.set_view(v, iset.sift[v])
    upload_s = time.perf_counter() - t0
    capacity = F * max(len(my_pairs), 1)

    def barrier():
        if world > 1:
            dist.barrier()
            if args.backend == "nccl":
                torch.cuda.synchronize()

    store = None
    if world > 1 and args.exchange == "shm":
        # the matcher writes its lists into this rank's slice of the shared segment
        try:
            store = D.SharedMatchStore(F * len(range(0, len(all_pairs), world)), rank, world, tdev)   # same on every rank
            m.use_result_buffer(store.slice)
        except D.SharedSegmentUnavailable:
            args.exchange = "rccl"
    if world > 1 and store is None:
        # page-locked result buffer: the lists go device -> host -> device (gather) at full PCIe rate
        m.use_result_buffer(D.pinned_array("local", capacity, tdev))
    gathered = [None]

    def step():
        out = m.compute(my_pairs, capacity=capacity)
        st = m.stats()
        counts = np.array([tv.num_matches if tv.status == capi.PAIR_MATCHED else 0 for tv in out], dtype=np.int64)
        n_corr = int(counts.sum())
        # the only collective of the path: the match lists travel to rank 0
        # (pair order restored there) for RANSAC / track building
        if store is not None:
            gathered[0] = store.collect(counts, len(all_pairs))
        elif world > 1:
            gathered[0] = D.gather_match_lists(counts, m.last_flat, len(all_pairs), rank, world, device=tdev)
        return out, st, n_corr

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    kern_ms, kern_launches, macs, n_corr = 0.0, 0, 0, 0
    for _ in range(args.steps):
        out, st, n_corr = step()
        kern_ms += st.tile_kernel_ms
        kern_launches += st.tile_kernel_launches
        macs += st.mac_count
    barrier()
    dt = time.perf_counter() - t0
    dt = D.max_over_ranks(dt, world, device=tdev)
    total_pairs = len(all_pairs)

    # the same pass continued through geometric verification (RANSAC-F,
    # bundler_matching.cc:194-219), reported beside the headline number
    verified = None
    if not args.no_verify:
        o2 = capi.default_match_options()
        o2.geometric_verification = 1
        m2 = HipExhaustiveMatching(V, device=device_index, options=o2, copy_results=False)
        for v in range(V):
            m2.set_view(v, iset.sift[v])
            xy = (iset.pos[v] + 0.5 - np.array([iset.width / 2, iset.height / 2])) / max(iset.width, iset.height)
            m2.set_positions(v, xy.astype(np.float32))
        m2.compute(my_pairs, capacity=capacity)
        barrier()
        t1 = time.perf_counter()
        outv = m2.compute(my_pairs, capacity=capacity)
        barrier()
        dtv = D.max_over_ranks(time.perf_counter() - t1, world, device=tdev)
        verified = {"pairs_per_s": len(all_pairs) / dtv, "ms_per_step": dtv * 1e3,
                    "ransac": "1000 iterations, threshold 0.0015, >= 30 inliers",
                    "accepted_pairs_rank0": int(sum(1 for tv in outv if tv.status == capi.PAIR_MATCHED)),
                    "inliers_rank0": int(sum(tv.num_inliers for tv in outv if tv.status == capi.PAIR_MATCHED))}
        m2.close()

    cascade = None
    if rank == 0 and world == 1 and not args.no_ba and not args.no_verify:
        try:
            cascade = cascade_bench(iset, V, my_pairs, capacity, device_index, not args.no_cpu_baseline)
        except Exception as e:
            cascade = {"error": str(e)}

    # N > 1: rank 0 consumes the lists of all ranks where the exchange left them
    exchanged = None
    if rank == 0 and world > 1 and gathered[0] is not None:
        exchanged = exchanged_tracks(gathered[0], all_pairs, V, F, args.exchange)

    tracks = None
    if rank == 0 and world == 1 and not args.no_ba:
        try:
            tracks = tracks_bench(out, V, F, not args.no_cpu_baseline)
        except Exception as e:
            tracks = {"error": str(e)}

    ba = None
    if not args.no_ba and rank == 0:
        try:
            from orthosfm_amd import ba as ba_mod
            ba = ba_mod.bench_global_ba()
        except Exception as e:       # BA reporting must never hide the matching line
            ba = {"error": str(e)}

    if ba is not None and "error" not in ba and not args.no_cpu_baseline and rank == 0:
        ba["cpu_baseline"] = ba_cpu_baseline()

    if ba is not None and "error" not in ba and rank == 0:
        try:
            ba["outlier_filter"] = outlier_filter_bench(device_index, not args.no_cpu_baseline)
        except Exception as e:
            ba["outlier_filter"] = {"error": str(e)}

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = total_pairs * args.steps / dt
        flops_per_launch = 2.0 * macs / max(kern_launches, 1)
        avg_launch_s = kern_ms / max(kern_launches, 1) * 1e-3
        achieved = flops_per_launch / max(avg_launch_s, 1e-12) / 1e12
        statuses = [tv.status for tv in out]
        line = {
            "metric": "image-pairs matched/sec + BA iterations/sec (N images, M tracks) at 1/2/4/8 GPU",
            "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "i8",
            "data": "synthetic",
            "config": {"workload": f"{V} orthographic images, {F} SIFT features/img, exhaustive matching "
                                   f"({total_pairs} pairs, low-res gate + two-way + cross-check + lists)",
                       "views": V, "features_per_view": F, "pairs": total_pairs,
                       "sharding": f"pairs round-robin over {world} rank(s)"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": I8_MFMA_PEAK_TOPS,
                         "unit": "TFLOP/s", "frac": achieved / I8_MFMA_PEAK_TOPS,
                         "traffic": pmc_traffic_per_launch(len(my_pairs) * args.steps / max(kern_launches, 1)),
                         "kernel": "match_tile_kernel<8, false, true, true>", "launches_per_step": kern_launches / args.steps,
                         "avg_launch_ms": avg_launch_s * 1e3,
                         "hbm_algorithmic_GBs": (st.algorithmic_bytes / max(st.tile_kernel_launches, 1))
                                                / max(avg_launch_s, 1e-12) / 1e9},
            "matched_pairs_rank0": int(sum(1 for s in statuses if s == capi.PAIR_MATCHED)),
            "correspondences_rank0": int(n_corr),
            "upload_s": upload_s,
        }
        if tracks is not None:
            line["tracks"] = tracks
        if exchanged is not None:
            line["exchange"] = exchanged
        if cascade is not None:
            line["cascade_hashing"] = cascade
        if verified is not None:
            line["with_geometric_verification"] = verified
        if ba is not None:
            line["ba"] = ba
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(iset, all_pairs, args.cpu_sample_pairs)
        print(json.dumps(line))
    m.close()
    if store is not None:
        store.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
