#!/usr/bin/env python3
"""PROBE (round 4, e3): what would a cell-sorted INTERNAL row numbering buy the gather kernels?

Today every level keeps the reference's row order (the subsampler's libstdc++ iteration order: spatially random), the queries are
VISITED in cell order (q_order) but the rows they gather lie anywhere.  This probe renumbers one level-0 problem -- points, features,
index matrix (rows AND values) -- into the neighbour grid's cell-sorted order and times the same kernels on both numberings:
kpconv_fused32 (Cin = Cout = 32), kpconv_fused (64), ind_max_pool, detect_head, and the first-layer kpconv_c1.
    python tools/sorted_layout_probe.py [fragments per stack]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def main():
    from d3feat_amd import ops
    from d3feat_amd import tf_custom_ops as tfo
    from d3feat_amd.kernels.kernel_points import create_kernel_points
    from d3feat_amd.utils.synthetic import room_fragment
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    dev = torch.device("cuda", 0)
    subs = [tfo.grid_subsampling(torch.from_numpy(room_fragment(i, 300000, 1.68)).to(dev), 0.03) for i in range(F)]
    pts = torch.cat([x for s in subs for x in (s, s)], 0).contiguous()
    lens = ops.as_lens([int(s.shape[0]) for s in subs for _ in (0, 1)], dev)
    N = pts.shape[0]
    grid = ops.NeighborGrid(pts, lens, 0.075)
    pts.order = grid.order
    nb, _ = grid.search(pts, lens, 42)
    torch.cuda.synchronize()
    order = grid.order.long()                      # sorted position -> original row
    inv = torch.empty_like(order)
    inv[order] = torch.arange(N, device=dev)
    inv_ext = torch.cat([inv, torch.tensor([N], device=dev)])            # shadow index stays N
    pts_s = pts[order].contiguous()
    nb_s = inv_ext[nb.long()[order]].int().contiguous()                   # rows in sorted order, values in sorted numbering
    g = torch.Generator(device="cpu").manual_seed(0)
    KP = create_kernel_points(0.045, 15, 1, 3, "center", rng=np.random.default_rng(1)).reshape(15, 3).astype(np.float32)
    print("stack of %d fragments: %d rows, K = %d" % (F, N, nb.shape[1]))
    for cin in (32, 64):
        f = torch.randn((N, cin), generator=g).to(dev)
        W = (torch.randn((15, cin, cin), generator=g) * 0.05).to(dev)
        f_s = f[order].contiguous()
        fn = ops.kpconv_fused32 if cin == 32 else ops.kpconv_fused
        a = fn(pts, pts, nb, f, KP, W, 0.03)
        pts_s.order = None
        b = fn(pts_s, pts_s, nb_s, f_s, KP, W, 0.03)
        err = (b - a[order]).abs().max().item()
        ta = timed(lambda: fn(pts, pts, nb, f, KP, W, 0.03))
        tb = timed(lambda: fn(pts_s, pts_s, nb_s, f_s, KP, W, 0.03))
        print("kpconv Cin=%3d   reference numbering (visited in cell order) %7.1f us   cell-sorted numbering %7.1f us   (%.2fx, max diff %.1e)"
              % (cin, ta, tb, ta / tb, err))
    f1 = torch.ones((N, 1), device=dev)
    W1 = (torch.randn((15, 1, 64), generator=g) * 0.2).to(dev)
    ta = timed(lambda: ops.kpconv_fused_c1(pts, pts, nb, f1, KP, W1, 0.03))
    tb = timed(lambda: ops.kpconv_fused_c1(pts_s, pts_s, nb_s, f1, KP, W1, 0.03))
    print("kpconv Cin=  1   %7.1f us -> %7.1f us (%.2fx)" % (ta, tb, ta / tb))
    x = torch.randn((N, 32), generator=g).to(dev)
    x_s = x[order].contiguous()
    ta = timed(lambda: ops.detect_head(x, nb, lens, None, stack_group=2))
    tb = timed(lambda: ops.detect_head(x_s, nb_s, lens, None, stack_group=2))
    print("detect_head      %7.1f us -> %7.1f us (%.2fx)" % (ta, tb, ta / tb))
    x128 = torch.randn((N, 128), generator=g).to(dev)
    x128_s = x128[order].contiguous()
    nbp, nbp_s = nb[: N // 4].contiguous(), nb_s[: N // 4].contiguous()    # a pooling-sized query set
    ta = timed(lambda: ops.ind_max_pool(x128, nbp))
    tb = timed(lambda: ops.ind_max_pool(x128_s, nbp_s))
    print("ind_max_pool     %7.1f us -> %7.1f us (%.2fx)" % (ta, tb, ta / tb))


if __name__ == "__main__":
    main()
