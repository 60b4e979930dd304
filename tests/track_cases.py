"""Seeded matchings for the track-building tests (shared by the CPU tests and
the golden generator)."""
import numpy as np


def random_matching(num_views, feats_per_view, num_scene_points, p_seen=0.5, p_false=0.03, seed=0):
    """Every scene point is seen by a random subset of views as a fixed feature
    of that view; all view pairs (view_1 > view_2, the reference's order) match
    the points they share, plus a few false matches that create conflicts and
    track merges."""
    rng = np.random.default_rng(seed)
    feat_of = np.full((num_views, num_scene_points), -1, dtype=np.int64)
    for v in range(num_views):
        seen = np.nonzero(rng.random(num_scene_points) < p_seen)[0][:feats_per_view]
        feat_of[v, seen] = rng.permutation(feats_per_view)[:seen.size]
    pairs, offsets, corr = [], [0], []
    for a in range(1, num_views):
        for b in range(a):
            both = np.nonzero((feat_of[a] >= 0) & (feat_of[b] >= 0))[0]
            m = np.stack([feat_of[a, both], feat_of[b, both]], axis=1)
            nf = int(p_false * max(len(both), 1)) + (1 if p_false > 0 else 0)
            if nf:
                fm = np.stack([rng.integers(0, feats_per_view, nf), rng.integers(0, feats_per_view, nf)], axis=1)
                m = np.concatenate([m, fm])
            m = m[np.argsort(m[:, 0], kind="stable")]            # lists come ordered by the first id
            # drop some pairs entirely (rejected by the matcher's gates)
            if rng.random() < 0.15:
                m = m[:0]
            pairs.append((a, b))
            corr.append(m)
            offsets.append(offsets[-1] + m.shape[0])
    colors = rng.integers(0, 256, (num_views * feats_per_view, 3)).astype(np.uint8)
    return {"view_sizes": np.full(num_views, feats_per_view, dtype=np.int32), "colors": colors,
            "pairs": np.array(pairs, dtype=np.int32).reshape(-1, 2),
            "pair_offsets": np.array(offsets, dtype=np.int64),
            "corr": (np.concatenate(corr) if corr else np.zeros((0, 2))).astype(np.int32)}


CASES = {
    "small": dict(num_views=5, feats_per_view=40, num_scene_points=60, seed=1),
    "conflicts": dict(num_views=8, feats_per_view=120, num_scene_points=150, p_false=0.15, seed=2),
    "clean": dict(num_views=6, feats_per_view=200, num_scene_points=300, p_false=0.0, seed=3),
    "wide": dict(num_views=20, feats_per_view=300, num_scene_points=500, p_seen=0.3, seed=4),
}
