#!/usr/bin/env python3
"""Every geometry launch of the pyramid, stand-alone, at the engine's F = 4 stacked shapes (4 self-pairs of ~29 k points):
grid builds, the conv / pool searches (full lists) and the upsampling searches (nearest only), the subsamplings.
    python tools/geom_bench.py [--frags 4] [--ab]      # --ab: D3F_NB_CELL=0 (lane-group kernel) next to the default, rows compared
Times are HIP-event averages over back-to-back launches of ONE op on an otherwise idle device (hot instruction cache): the
lower bound of what the op costs inside a replay."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from d3feat_amd import ops, tf_custom_ops as tfo          # noqa: E402
from d3feat_amd.utils.synthetic import room_fragment      # noqa: E402


def timeit(fn, iters=20, warm=3, graph=False):
    for _ in range(warm):
        fn()
    if graph:
        # GPU time without the host's ~35 us per eager call: the launches of one op captured back to back in a HIP graph
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            with torch.cuda.graph(g, stream=st):
                for _ in range(iters):
                    fn()
            g.replay()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            g.replay()
            e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / iters * 1e3
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frags", type=int, default=4)
    ap.add_argument("--ab", action="store_true")
    ap.add_argument("--q", type=str, default="", help="comma list of D3F_NBC_Q values to try on the full searches (Q:DBG pairs skip phases: -DD3F_NBC_MEASURE builds only)")
    ap.add_argument("--only", type=str, default="", help="substring filter on the op names (e.g. 'L0 conv')")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--internal", action="store_true", help="conv searches in the engine's form: query_grid = the grid, internal numbering")
    ap.add_argument("--limits", type=str, default="", help="comma list of the five matrix widths (default 37,35,36,38,38; bench.py's: 42,42,46,51,49)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    limits = [int(x) for x in args.limits.split(",")] if args.limits else [37, 35, 36, 38, 38]
    subs = [tfo.grid_subsampling(torch.from_numpy(room_fragment(s, n_raw=300000, edge=1.68)).to(dev), 0.03) for s in range(args.frags)]
    pts = torch.cat([x for s in subs for x in (s, s)], 0)
    lens = [int(s.shape[0]) for s in subs for _ in (0, 1)]
    levels = [(pts, ops.as_lens(lens, dev))]
    dl = 0.06
    for l in range(4):
        p, pl, _, _ = ops.batch_grid_subsample(levels[-1][0], levels[-1][1], dl)
        levels.append((p, pl))
        dl *= 2
    r = 0.03 * 2.5
    rows = []
    qgrid_prev = None
    for l, (p, pl) in enumerate(levels):
        grid = ops.NeighborGrid(p, pl, r)
        rows.append(("L%d build        %7d" % (l, p.shape[0]), lambda p=p, pl=pl, r=r: ops.NeighborGrid(p, pl, r)))
        st = torch.zeros((2,), dtype=torch.int32, device=dev)
        o1 = torch.empty((p.shape[0], limits[l]), dtype=torch.int32, device=dev)
        rows.append(("L%d conv search  %7d" % (l, p.shape[0]),
                     lambda grid=grid, p=p, pl=pl, l=l, o1=o1, st=st: grid.search(p, pl, limits[l], cap=192, out=o1, status=st, reset_status=False, want_kmax=False,
                                                                                  **(dict(query_grid=grid, internal=True) if args.internal else {}))))
        if l + 1 < len(levels):
            q, ql = levels[l + 1]
            o2 = torch.empty((q.shape[0], limits[l]), dtype=torch.int32, device=dev)
            rows.append(("L%d pool search  %7d" % (l, q.shape[0]),
                         lambda grid=grid, q=q, ql=ql, l=l, o2=o2, st=st: grid.search(q, ql, limits[l], cap=192, out=o2, status=st, reset_status=False, want_kmax=False)))
        if l > 0:
            q, ql = levels[l - 1]
            g2 = ops.NeighborGrid(p, pl, r)     # (radius of level l: 2 x the radius of level l - 1, the reference's up_i radius)
            o3 = torch.empty((q.shape[0], 1), dtype=torch.int32, device=dev)
            qg = qgrid_prev
            rows.append(("L%d up search    %7d" % (l, q.shape[0]),
                         lambda g2=g2, q=q, ql=ql, o3=o3, st=st, qg=qg, h=1.75 * 0.03 * 2 ** l: g2.search(q, ql, 1, cap=192, first_only=True, out=o3, status=st, reset_status=False,
                                                                                          want_kmax=False, nn_hint=h, query_grid=qg)))
            rows.append(("L%d up search-nq %6d" % (l, q.shape[0]),
                         lambda g2=g2, q=q, ql=ql, o3=o3, st=st, h=1.75 * 0.03 * 2 ** l: g2.search(q, ql, 1, cap=192, first_only=True, out=o3, status=st, reset_status=False,
                                                                                          want_kmax=False, nn_hint=h)))
        qgrid_prev = grid
        if l + 1 < len(levels):
            rows.append(("L%d subsample    %7d" % (l, p.shape[0]), lambda p=p, pl=pl, d=0.06 * 2 ** l: ops.batch_grid_subsample(p, pl, d)))
        r *= 2
    variants = [("default", {})]
    if args.ab:
        variants.append(("lane-group", {"D3F_NB_CELL": "0", "D3F_NB_NEAREST": "0"}))
    for qv in [x for x in args.q.split(",") if x]:
        if ":" in qv:
            variants.append(("Q%s d%s" % tuple(qv.split(":")), {"D3F_NBC_Q": qv.split(":")[0], "D3F_NBC_DBG": qv.split(":")[1]}))
        else:
            variants.append(("Q=" + qv, {"D3F_NBC_Q": qv}))
    print("%-28s" % "op  (rows)" + "".join("%12s" % v[0] for v in variants))
    sums = [0.0] * len(variants)
    for name, fn in rows:
        if args.only and args.only not in name:
            continue
        line = "%-28s" % name
        for vi, (vn, env) in enumerate(variants):
            for k in ("D3F_NB_CELL", "D3F_NBC_Q", "D3F_NBC_DBG", "D3F_NB_NEAREST"):
                os.environ.pop(k, None)
            os.environ.update(env)
            t = timeit(fn, iters=args.iters, graph=("search" in name))
            sums[vi] += t
            line += "%12.1f" % t
        print(line)
    for k in ("D3F_NB_CELL", "D3F_NBC_Q", "D3F_NBC_DBG", "D3F_NB_NEAREST"):
        os.environ.pop(k, None)
    print("%-28s" % "sum (us)" + "".join("%12.1f" % s for s in sums))
    # equality of the two forms on every full search of the pyramid
    if args.ab:
        r = 0.03 * 2.5
        for l, (p, pl) in enumerate(levels):
            grid = ops.NeighborGrid(p, pl, r)
            for tag, (q, ql) in (("conv", (p, pl)),) + ((("pool", levels[l + 1]),) if l + 1 < len(levels) else ()):
                os.environ.pop("D3F_NB_CELL", None)
                a, _ = grid.search(q, ql, limits[l], cap=192)
                os.environ["D3F_NB_CELL"] = "0"
                b, _ = grid.search(q, ql, limits[l], cap=192)
                os.environ.pop("D3F_NB_CELL", None)
                print("L%d %s: forms equal: %s   (rows %d)" % (l, tag, bool(torch.equal(a, b)), q.shape[0]))
            if l > 0:
                q, ql = levels[l - 1]
                g2 = ops.NeighborGrid(p, pl, r)
                qg = ops.NeighborGrid(q, ql, r / 2)
                h = 1.75 * 0.03 * 2 ** l
                os.environ["D3F_NB_NEAREST"] = "0"
                b, _ = g2.search(q, ql, 1, cap=192, first_only=True, want_kmax=False, nn_hint=h)
                os.environ.pop("D3F_NB_NEAREST", None)
                for tag2, kw in (("ordered", dict(query_grid=qg)), ("plain", {}), ("no hint", dict(query_grid=qg, hint0=True))):
                    hh = 0.0 if kw.pop("hint0", False) else h
                    a, _ = g2.search(q, ql, 1, cap=192, first_only=True, want_kmax=False, nn_hint=hh, **kw)
                    print("L%d up (%s): forms equal: %s   (rows %d)" % (l, tag2, bool(torch.equal(a, b)), q.shape[0]))
            r *= 2


if __name__ == "__main__":
    main()
