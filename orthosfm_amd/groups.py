"""Host-side mirror of orthosfm::buildGroups (src/data_structures/group.cpp:13-88):
the order in which the incremental reconstruction adds views."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import capi


@dataclass
class ViewGroup:
    """orthosfm::ViewGroup (group.h:22-27)."""
    ids: list
    tracks: int


def flatten_tracks(tracks):
    """tracks: sequence of tracks, each a sequence of view ids (or of features with .viewID)."""
    offs = np.zeros(len(tracks) + 1, dtype=np.int64)
    views = []
    for i, t in enumerate(tracks):
        feats = getattr(t, "features", t)
        ids = [int(getattr(f, "viewID", f)) for f in feats]
        views.extend(ids)
        offs[i + 1] = offs[i] + len(ids)
    return offs, np.asarray(views, dtype=np.int32)


def build_groups(view_ids, tracks, group_size: int = 3, device: int = 0, verbose: bool = False):
    offs, views = flatten_tracks(tracks)
    return build_groups_flat(view_ids, offs, views, group_size, device, verbose)


def build_groups_flat(view_ids, offs, views, group_size: int = 3, device: int = 0, verbose: bool = False):
    """The same on tracks given as CSR: offs [num_tracks + 1], views [..] = Feature::viewID."""
    view_ids = np.ascontiguousarray(view_ids, dtype=np.int32)
    offs = np.ascontiguousarray(offs, dtype=np.int64)
    views = np.ascontiguousarray(views, dtype=np.int32)
    if views.size == 0:
        views = np.zeros(1, np.int32)
    cap = max(len(view_ids), 1)
    groups = np.zeros((cap, group_size), dtype=np.int32)
    gtracks = np.zeros(cap, dtype=np.int32)
    n = C.c_int32()
    capi.check(capi.lib.osfm_build_groups(device, len(view_ids), capi._ptr(view_ids, C.c_int32), len(offs) - 1,
                                          capi._ptr(offs, C.c_int64), capi._ptr(views, C.c_int32), group_size, cap,
                                          capi._ptr(groups, C.c_int32), capi._ptr(gtracks, C.c_int32), C.byref(n)))
    out = [ViewGroup([int(x) for x in groups[i]], int(gtracks[i])) for i in range(n.value)]
    if verbose:
        for i, g in enumerate(out):                               # printGroups (group.cpp:212-226)
            print(f"Group {i}: [{', '.join(str(x) for x in g.ids)}] --> {g.tracks} tracks")
        print(f"--> built {len(out)} groups.")
    return out
