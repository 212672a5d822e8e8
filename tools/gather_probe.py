#!/usr/bin/env python3
"""Is a gather kernel bound by where its neighbour rows come from?  Times kpconv_fused32 / ind_max_pool / detect_head on the
level-0 stack of F synthetic fragments with (a) the real neighbour matrix, (b) the same matrix folded onto 256 distinct rows
(every gather an L1 / L2 hit on a tiny set, same instruction stream), (c) the real matrix with queries in memory order instead
of cell order.  GPU only; prints microseconds per launch."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(F=4):
    from d3feat_amd import ops
    from d3feat_amd import tf_custom_ops as tfo
    from d3feat_amd.kernels.kernel_points import create_kernel_points
    from d3feat_amd.utils.synthetic import room_fragment
    dev = torch.device("cuda", 0)
    subs = [tfo.grid_subsampling(torch.from_numpy(room_fragment(s, 300000, 1.68)).to(dev), 0.03) for s in range(F)]
    pts = torch.cat([x for s in subs for x in (s, s)], 0)
    lens = ops.as_lens([int(s.shape[0]) for s in subs for _ in (0, 1)], dev)
    grid = ops.NeighborGrid(pts, lens, 0.075)
    nb, _ = grid.search(pts, lens, 42)
    N = pts.shape[0]
    g = torch.Generator(device="cpu").manual_seed(0)
    f32 = torch.randn((N, 32), generator=g).to(dev)
    f128 = torch.randn((N, 128), generator=g).to(dev)
    W = (torch.randn((15, 32, 32), generator=g) * 0.2).to(dev)
    KP = create_kernel_points(0.045, 15, 1, 3, "center", rng=np.random.default_rng(1)).reshape(15, 3).astype(np.float32)
    folded = torch.where(nb < N, nb % 256, nb)                  # same validity pattern, 256 distinct rows
    sub1 = tfo.batch_grid_subsampling(pts, lens, 0.06)
    p1, l1 = sub1
    pool = grid.search(p1, l1, 42)[0]
    pool_f = torch.where(pool < N, pool % 256, pool)

    def timed(fn, reps=10):
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps * 1e3

    def with_order(t, order):
        t.order = order
        return t
    for name, q_order in (("cell order", grid.order), ("memory order", None)):
        P = with_order(pts.clone(), q_order) if q_order is not None else pts.clone()
        for label, idx in (("real", nb), ("folded-256", folded)):
            if q_order is not None:
                idx = with_order(idx.clone(), q_order)
            t = timed(lambda: ops.kpconv_fused32(P, pts, idx, f32, KP, W, 0.03))
            print("kpconv_fused32  %-12s %-11s %8.1f us" % (name, label, t))
            hl = torch.ones((lens.numel(),), dtype=torch.int32, device=dev)
            t = timed(lambda: ops.detect_head(f32, idx, lens, hl))
            print("detect_head     %-12s %-11s %8.1f us" % (name, label, t))
    o1 = ops.NeighborGrid(p1, l1, 0.15).order
    for label, idx in (("real", pool), ("folded-256", pool_f)):
        for name, order in (("cell order", o1), ("memory order", None)):
            i2 = idx.clone()
            if order is not None:
                i2.order = order
            t = timed(lambda: ops.ind_max_pool(f128, i2))
            print("ind_max_pool    %-12s %-11s %8.1f us" % (name, label, t))
    print("rows", N, "pool rows", p1.shape[0])


if __name__ == "__main__":
    main()
