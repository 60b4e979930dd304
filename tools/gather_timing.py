"""Times rank 0's post-gather work (device reorder + one D2H) at the size of an 8-rank run."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from orthosfm_amd import distributed as D

dev = torch.device("cuda:0")
world, per_rank_pairs, per_pair = 8, 1225, 6400
num_pairs = world * per_rank_pairs
heads = torch.full((world, per_rank_pairs), per_pair, dtype=torch.int64, device=dev)
width = per_rank_pairs * per_pair
bufs = [torch.randint(0, 20000, (width, 2), dtype=torch.int32, device=dev) for _ in range(world)]
for it in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    c, off, corr = D.assemble_global_order(heads, bufs, num_pairs, world, dev)
    torch.cuda.synchronize()
    print(f"assemble {1e3 * (time.perf_counter() - t0):.1f} ms for {corr.shape[0] / 1e6:.1f} M correspondences "
          f"({corr.nbytes / 1e6:.0f} MB to the host)")
# the per-rank side: pinned host buffer -> device
loc = D.pinned_array("local", width, dev)
t0 = time.perf_counter()
p = torch.zeros((width, 2), dtype=torch.int32, device=dev)
p.copy_(torch.from_numpy(loc), non_blocking=True)
torch.cuda.synchronize()
print(f"local upload {1e3 * (time.perf_counter() - t0):.1f} ms for {loc.nbytes / 1e6:.0f} MB")
