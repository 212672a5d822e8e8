"""TEST INFRASTRUCTURE ONLY (oracle).  Never imported by d3feat_amd/.

Fragment-level checker shared by tests/, __graft_entry__.smoke() and bench.py's `parity` / `cpu_baseline` legs:
one fragment through the CPU restatement exactly as the reference's tester does it (utils/tester.py:196-213):
    raw cloud -> grid subsample (stage 0) -> stacked with itself (datasets/ThreeDMatch.py:190-192)
              -> tf_descriptor_input (datasets/common.py:1301-1413) -> sess.run([out_features, out_scores])
and the comparison of a GPU result -- possibly one slice of a bigger stack [c_1; c_1; c_2; c_2; ...] as the
fragment engine builds it -- with that reference: points and indices bit-exact, descriptors / scores by
max-abs difference (BASELINE.json north_star: 1e-4, absolute).
"""
import time

import numpy as np

from . import network_np as onp

TOL = 1e-4


def geometry_ops(co=None, rl=None):
    """(batch_neighbors, batch_subsampling, grid_subsampling) from the reference's own C++ (rl) or the C restatement (co)."""
    if rl is not None:
        return (lambda q, s, ql, sl, r: rl.batch_nanoflann_neighbors(q, s, ql, sl, r),
                lambda p, l, dl: rl.batch_grid_subsampling(p, l, dl),
                lambda p, dl: rl.grid_subsampling(p, dl))
    return (lambda q, s, ql, sl, r: co.batch_neighbors(q, s, ql, sl, r),
            lambda p, l, dl: co.batch_grid_subsampling(p, l, dl),
            lambda p, dl: co.grid_subsampling(p, dl))


def fragment_reference(cfg, W, raw, limits, co=None, rl=None, forward=True, clouds=None):
    """One fragment on the CPU.  raw: f32[n,3] raw cloud (stage-0 subsampled here) -- or pass `clouds`, a list of already
    subsampled clouds to stack as they are (KITTI-style pairs of different frames).
    -> dict(sub, inp, desc, score, t_geometry, t_network)."""
    nbr, bsub, gsub = geometry_ops(co, rl)
    t0 = time.perf_counter()
    if clouds is None:
        s0 = gsub(raw, np.float32(cfg.first_subsampling_dl))
        clouds = [s0, s0]
    pts = np.concatenate(clouds)
    lens = np.asarray([len(c) for c in clouds], np.int32)
    inp = onp.descriptor_input(cfg, pts, np.ones((len(pts), 1), np.float32), lens, limits, nbr, bsub)
    t1 = time.perf_counter()
    desc = score = None
    if forward:
        desc, score = onp.forward(cfg, W, inp)
    t2 = time.perf_counter()
    return dict(sub=clouds[0], inp=inp, desc=desc, score=score, t_geometry=t1 - t0, t_network=t2 - t1)


def local_indices(mat, row0, nrows, sup0, nsup, pad_from):
    """Rows [row0, row0+nrows) of an index matrix of a bigger stack, re-based to the fragment's own stack:
    supports [sup0, sup0+nsup) -> [0, nsup); every value >= pad_from (the big stack's shadow index, or garbage beyond
    it) -> nsup, the fragment's own shadow index."""
    m = np.asarray(mat[row0:row0 + nrows]).astype(np.int64)
    out = np.where(m >= pad_from, nsup, m - sup0)
    return out.astype(np.int32)


def equal_up_to_ties(got, want, q, s):
    """Index matrices [n, K] equal, except that a row may differ by a permutation INSIDE a run of bit-equal fp32 squared distances.
    The reference's active search (nanoflann + std::sort, tf_custom_ops/tf_neighbors/neighbors/neighbors.cpp:211-332) leaves the
    order inside such a run unspecified; the restatement and the HIP kernel order by (d2, index) = the reference's own
    batch_ordered_neighbors (neighbors.cpp:125-208; SURVEY.md section 8c "parity definitions").  q / s: the query / support points
    (shadow index = len(s)).  -> (ok, number of rows that differ inside ties)."""
    got, want = np.asarray(got), np.asarray(want)
    if got.shape != want.shape:
        return False, 0
    rows = np.nonzero(np.any(got != want, axis=1))[0]
    if not len(rows):
        return True, 0
    sp = np.concatenate([np.asarray(s, np.float32), np.full((1, 3), 1e6, np.float32)])

    def d2(r, row):
        d = (sp[np.minimum(row, len(sp) - 1)] - q[r]).astype(np.float32)
        d = d * d
        return ((d[:, 0] + d[:, 1]) + d[:, 2]).view(np.uint32)      # nanoflann's metric: ((0 + dx^2) + dy^2) + dz^2 in fp32
    for r in rows:
        if sorted(got[r]) != sorted(want[r]) or not np.array_equal(d2(r, got[r]), d2(r, want[r])):
            return False, len(rows)
    return True, len(rows)


def compare_fragment(ref, pts, desc, score, nb0=None, row0=0, total=None):
    """ref: fragment_reference(...) result; pts/desc/score: numpy arrays of THIS fragment's rows; nb0: the level-0
    neighbour matrix of the whole stack the fragment was computed in (global indices), row0 the fragment's first row,
    total the stack's row count (its shadow index).  -> dict(points_equal, idx_equal, desc_max_abs, score_max_abs)."""
    inp = ref["inp"]
    want_p = inp["points"][0]
    res = dict(points_equal=bool(pts.shape == want_p.shape and
                                 np.array_equal(np.ascontiguousarray(pts).view(np.uint32), want_p.view(np.uint32))))
    if nb0 is not None:
        n = want_p.shape[0]
        w = inp["neighbors"][0]
        g = local_indices(nb0, row0, n, row0, n, total if total is not None else n)
        ok, ties = equal_up_to_ties(g[:, :w.shape[1]], w, want_p, want_p) if g.shape[0] == w.shape[0] else (False, 0)
        res["idx_equal"] = bool(ok and (g[:, w.shape[1]:] == n).all())
        res["idx_tie_rows"] = int(ties)       # rows that differ from the reference's nanoflann order only inside bit-equal-d2 runs
    if desc is not None and ref["desc"] is not None:
        res["desc_max_abs"] = float(np.abs(desc.astype(np.float64) - ref["desc"]).max()) if desc.shape == ref["desc"].shape \
            else float("inf")
        res["score_max_abs"] = float(np.abs(score.astype(np.float64) - ref["score"]).max()) if score.shape == ref["score"].shape \
            else float("inf")
    return res


def check_pyramid_slice(flat, ref, L, offsets, totals, fast=True):
    """Bit-exact comparison of every level of a GPU pyramid (`flat`: the tf_descriptor_input list, device tensors whose
    valid rows are a prefix) with one fragment's reference.  offsets[l] / totals[l]: the fragment's first row and the
    stack's row count at level l.  Raises AssertionError naming the first mismatch."""
    inp = ref["inp"]
    for l in range(L):
        want_p = inp["points"][l]
        n = want_p.shape[0]
        got_p = flat[l][offsets[l]:offsets[l] + n].cpu().numpy()
        assert np.array_equal(got_p.view(np.uint32), want_p.view(np.uint32)), ("points", l)
        for name, off in (("neighbors", L), ("pools", 2 * L), ("upsamples", 3 * L)):
            w = inp[name][l]
            if w.shape[0] == 0:
                continue
            # rows live at the level of the QUERIES, values index the level of the SUPPORTS
            ql = l + 1 if name == "pools" else l
            sl = l + 1 if name == "upsamples" else l
            ns = inp["points"][sl].shape[0]
            g = local_indices(flat[off + l].cpu().numpy(), offsets[ql], w.shape[0], offsets[sl], ns, totals[sl])
            if name == "upsamples" and fast:
                assert np.array_equal(g[:, 0], w[:, 0]), (name, l)
            else:
                assert np.array_equal(g[:, :w.shape[1]], w), (name, l)
                assert (g[:, w.shape[1]:] == ns).all(), (name, l, "padding")
