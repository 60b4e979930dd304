"""Generates tests/golden/tracks_*.npz: inputs of tests/track_cases.py and the
outputs of the REFERENCE's own sfm::bundler::Tracks::compute
(src/mve/sfm/bundler_tracks.cc, compiled into oracle/_ref/libref_tracks.so by
oracle/Makefile).  Run in the build container (needs /root/reference):
    python tests/golden/make_tracks_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib
import track_cases


def main():
    assert oracle_lib.ref_tracks() is not None, "build oracle/_ref first (make -C oracle)"
    for name, kw in track_cases.CASES.items():
        m = track_cases.random_matching(**kw)
        out = oracle_lib.ref_tracks_compute(m["view_sizes"], m["colors"], m["pairs"], m["pair_offsets"], m["corr"])
        np.savez_compressed(os.path.join(HERE, f"tracks_{name}.npz"), **m,
                            out_track_ids=out["track_ids"], out_track_offsets=out["track_offsets"],
                            out_track_features=out["track_features"], out_track_colors=out["track_colors"])
        print(name, "tracks", len(out["track_offsets"]) - 1, "features", out["track_features"].shape[0])


if __name__ == "__main__":
    main()
