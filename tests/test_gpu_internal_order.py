"""The INTERNAL numbering of round 6: inside a replay every level is kept in the cell order of its own neighbour grid (index
matrices, point arrays, activations); the records leave in the reference's row order.  Checked here:
  * d3f_neighbor_grid_search_ordered with D3F_NB_INTERNAL: renumbering rows and entries back gives the matrices of the reference
    numbering bit for bit -- conv (queries = supports), pool (other queries, full rows) and up (nearest only), through all three
    search kernels (sizes on both sides of the dispatcher's thresholds);
  * FragmentEngine with and without it: the same records bit for bit, and reference_order_flat == the other engine's pyramid."""
import os

import numpy as np
import pytest
import torch

from conftest import surface_cloud

pytestmark = pytest.mark.gpu


def _back(mat, q_order, s_order, n_s):
    """internal matrix -> reference numbering: row j belongs to query q_order[j], entry v < n_s is support s_order[v]."""
    out = torch.empty_like(mat)
    ext = torch.cat([s_order.long(), torch.tensor([0], device=mat.device)])
    v = mat.long()
    valid = (v >= 0) & (v < n_s)
    vals = torch.where(valid, ext[torch.where(valid, v, torch.full_like(v, n_s))], v).to(torch.int32)
    out[q_order.long()] = vals
    return out


@pytest.mark.parametrize("n_raw", [9000, 60000, 400000])
def test_internal_matrices_renumber_back_to_the_reference_ones(device, n_raw):
    from d3feat_amd import ops
    s0 = torch.from_numpy(surface_cloud(3, n_raw=n_raw)).to(device)
    s1 = torch.from_numpy(surface_cloud(4, n_raw=n_raw // 2)).to(device)
    pts = torch.cat([s0, s1])
    lens = ops.as_lens([s0.shape[0], s1.shape[0]], device)
    sub, sl, _, _ = ops.batch_grid_subsample(pts, lens, 0.06)
    g0, g1 = ops.NeighborGrid(pts, lens, 0.075), ops.NeighborGrid(sub, sl, 0.15)
    n0, n1 = pts.shape[0], sub.shape[0]
    assert torch.equal(g0.xyz, pts[g0.order.long()]) and torch.equal(g0.inv[g0.order.long()].long(), torch.arange(n0, device=device))
    cases = (("conv", g0, pts, lens, g0, 34, False, 0.0), ("pool", g0, sub, sl, g1, 30, False, 0.0),
             ("up", g1, pts, lens, g0, 1, True, 1.75 * 0.06), ("up without hint", g1, pts, lens, g0, 1, True, 0.0))
    for tag, g, q, ql, qg, width, fo, hint in cases:
        ref, _ = g.search(q, ql, width, first_only=fo, want_kmax=False, nn_hint=hint)
        got, _ = g.search(q, ql, width, first_only=fo, want_kmax=False, nn_hint=hint, query_grid=qg, internal=True)
        back = _back(got, qg.order, g.order, g.Ns)
        assert torch.equal(back, ref), (tag, n_raw, int((back != ref).any(1).sum()))
        # the visiting order alone changes nothing
        same, _ = g.search(q, ql, width, first_only=fo, want_kmax=False, nn_hint=hint, query_grid=qg)
        assert torch.equal(same, ref), (tag, "ordered", n_raw)
    with pytest.raises(ValueError):
        g0.search(sub, sl, 30, internal=True)


def test_engine_on_the_internal_numbering_equals_the_reference_numbering(device):
    from d3feat_amd.engine import FragmentEngine
    from d3feat_amd.models.variables import build_variables
    from d3feat_amd.utils.config import threedmatch_config
    from d3feat_amd.utils.synthetic import room_fragment
    cfg = threedmatch_config()
    W = build_variables(cfg, seed=42, randomize_bn=True).values
    limits = np.asarray([37, 35, 36, 38, 38], np.int32)
    raws = [torch.from_numpy(room_fragment(s, n_raw=n, edge=1.0)).to(device) for s, n in ((31, 40000), (32, 30000))]
    kw = dict(raw_cap=50000, n0_cap=14000, slots=1, device=device, batch=2)
    a = FragmentEngine(cfg, W, limits, internal_order=True, **kw)
    b = FragmentEngine(cfg, W, limits, internal_order=False, **kw)
    assert a.internal and not b.internal
    a.submit(0, raws)
    b.submit(0, raws)
    ra, rb = a.fetch(0, packed=True), b.fetch(0, packed=True)
    assert a.fallbacks == 0 and b.fallbacks == 0
    for x, y in zip(ra, rb):
        assert x.shape == y.shape and torch.equal(x, y)                  # the records: bit for bit, in the reference's row order
    fa, fb = a.reference_order_flat(0), b.slots[0].flat
    L = cfg.num_layers
    for l in range(L):
        n = int(fb[l].n_dev.item())
        assert int(fa[l].shape[0]) == n and torch.equal(fa[l], fb[l][:n]), ("points", l)
    for k in range(L, 4 * L):
        if fb[k].shape[0] == 0:
            continue
        n = fa[k].shape[0]
        assert torch.equal(fa[k], fb[k][:n]), ("index matrix", k)
    # the separate outputs of the tuple form
    a.submit(0, raws)
    b.submit(0, raws)
    for (pa, da, sa), (pb, db, sb) in zip(a.fetch(0), b.fetch(0)):
        assert torch.equal(pa, pb) and torch.equal(da, db) and torch.equal(sa.reshape(-1), sb.reshape(-1))
