"""Regenerates tests/golden/match_*.npz from the REFERENCE's own matcher.

Run in the build container only (needs /root/reference):
    make -C oracle && python tests/golden/make_match_golden.py
It loads oracle/_ref/libref_match.so (the reference's matching.cc,
nearest_neighbor.cc, exhaustive_matching.cc compiled where they lie) and
stores inputs + the reference's outputs.  The .npz files are data only.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

import match_cases  # noqa: E402
import oracle_lib  # noqa: E402


def main():
    ref = oracle_lib.ref_matcher()
    assert ref is not None, "oracle/_ref/libref_match.so missing: run make -C oracle"
    out = {}
    for kind, cases in (("u16", match_cases.u16_cases()), ("s16", match_cases.s16_cases())):
        for name, (s1, s2, lowe) in cases.items():
            m12, m21 = ref.twoway(s1, s2, lowe)
            c12, c21 = ref.remove_inconsistent(m12, m21)
            cnt = ref.count_consistent(m12, m21)
            nn = np.array([ref.nn_find(s1[i], s2) for i in range(min(s1.shape[0], 8))],
                          dtype=np.int32).reshape(-1, 4) if s2.shape[0] else np.zeros((0, 4), np.int32)
            k = f"{kind}/{name}/"
            out[k + "s1"] = s1
            out[k + "s2"] = s2
            out[k + "lowe"] = np.float32(lowe)
            out[k + "m12"] = m12
            out[k + "m21"] = m21
            out[k + "c12"] = c12
            out[k + "c21"] = c21
            out[k + "count"] = np.int32(cnt)
            out[k + "nn"] = nn
    np.savez_compressed(os.path.join(HERE, "match_twoway.npz"), **out)

    views = match_cases.exhaustive_views()
    rm = oracle_lib.RefExhaustive(views)
    ex = {}
    for v, (s, u) in enumerate(views):
        ex[f"view{v}/sift"] = s
        ex[f"view{v}/surf"] = u
    V = len(views)
    for a in range(V):
        for b in range(V):
            if a == b:
                continue
            m12, m21 = rm.pairwise_match(a, b)
            ex[f"pair{a}_{b}/m12"] = m12
            ex[f"pair{a}_{b}/m21"] = m21
            ex[f"pair{a}_{b}/lowres40"] = np.int32(rm.pairwise_match_lowres(a, b, 40))
            ex[f"pair{a}_{b}/lowres500"] = np.int32(rm.pairwise_match_lowres(a, b, 500))
    np.savez_compressed(os.path.join(HERE, "match_exhaustive.npz"), **ex)
    print("wrote", len(out), "+", len(ex), "arrays")


if __name__ == "__main__":
    main()
