"""first_only search with / without nn_hint at the bench's level-0 -> level-1 shapes (4 stacked self-pairs)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from d3feat_amd import ops, tf_custom_ops as tfo
from d3feat_amd.utils.synthetic import room_fragment
dev = torch.device("cuda", 0)
subs = [tfo.grid_subsampling(torch.from_numpy(room_fragment(s, n_raw=300000, edge=1.68)).to(dev), 0.03) for s in range(4)]
pts = torch.cat([x for s in subs for x in (s, s)], 0)
lens = [int(s.shape[0]) for s in subs for _ in (0, 1)]
pool, pl, _, _ = ops.batch_grid_subsample(pts, lens, 0.06)
grid = ops.NeighborGrid(pool, pl, 0.30)
for hint in (0.0, 1.75 * 0.06, 0.0, 1.75 * 0.06):
    for _ in range(3):
        out, st = grid.search(pts, lens, 42, first_only=True, nn_hint=hint)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        out, st = grid.search(pts, lens, 42, first_only=True, nn_hint=hint)
    e.record(); torch.cuda.synchronize()
    print("hint %.3f: %.1f us   col0 checksum %d" % (hint, s.elapsed_time(e) / 20 * 1e3, int(out[:, 0].long().sum())))
