"""GPU parity of the preprocessing kernels against the oracle: bit-exact (integer / index work, and fp32
barycentres whose every rounding step is pinned)."""
import numpy as np
import pytest
import torch

from conftest import bits, small_cloud, surface_cloud

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("n,dl", [(1, 0.1), (5, 0.1), (13, 0.05), (14, 0.05), (100, 0.05), (3000, 0.02), (50000, 0.03),
                                  (200000, 0.011)])
def test_grid_subsampling_bit_exact(device, coracle, n, dl):
    from d3feat_amd import tf_custom_ops as tfo
    p = small_cloud(n, n)
    want = coracle.grid_subsampling(p, dl)
    got = tfo.grid_subsampling(_t(p, device), dl).cpu().numpy()
    assert got.shape == want.shape
    assert np.array_equal(bits(got), bits(want))


def test_batch_grid_subsampling_bit_exact(device, coracle):
    from d3feat_amd import tf_custom_ops as tfo
    a, b, c = small_cloud(1, 40000), small_cloud(2, 7000, (1, 1, 1)), small_cloud(3, 15)
    pts = np.concatenate([a, b, c])
    lens = [len(a), len(b), len(c)]
    want_p, want_b = coracle.batch_grid_subsampling(pts, lens, 0.04)
    got_p, got_b = tfo.batch_grid_subsampling(_t(pts, device), _t(np.asarray(lens, np.int32), device), 0.04)
    assert np.array_equal(got_b.cpu().numpy(), want_b)
    assert np.array_equal(bits(got_p.cpu().numpy()), bits(want_p))


def test_grid_subsampling_features_classes(device, coracle):
    from d3feat_amd import ops
    rng = np.random.default_rng(5)
    p = small_cloud(7, 20000)
    f = rng.standard_normal((20000, 4)).astype(np.float32)
    c = rng.integers(0, 7, (20000, 2)).astype(np.int32)
    wp, wf, wc = coracle.grid_subsampling(p, 0.05, f, c)
    gp, _, gf, gc = ops.batch_grid_subsample(_t(p, device), [20000], 0.05, _t(f, device), _t(c, device))
    assert np.array_equal(bits(gp.cpu().numpy()), bits(wp))
    assert np.array_equal(bits(gf.cpu().numpy()), bits(wf))
    assert np.array_equal(gc.cpu().numpy(), wc)


def test_grid_subsampling_status_flags(device):
    from d3feat_amd import _lib, ops
    p = small_cloud(1, 100)
    with pytest.raises(_lib.D3FeatLibraryError):
        ops.batch_grid_subsample(_t(p, device), [100, 0], 0.05)   # empty batch element


@pytest.mark.parametrize("seed", [0, 1])
def test_batch_neighbors_bit_exact(device, coracle, seed):
    from d3feat_amd import tf_custom_ops as tfo
    s0 = surface_cloud(seed)
    s1 = surface_cloud(seed + 10)[: len(s0) // 2]
    pts = np.concatenate([s0, s1])
    lens = np.asarray([len(s0), len(s1)], np.int32)
    r = np.float32(0.075)
    want = coracle.batch_neighbors(pts, pts, lens, lens, r)
    got = tfo.batch_ordered_neighbors(_t(pts, device), _t(pts, device), _t(lens, device), _t(lens, device), r).cpu().numpy()
    assert got.shape == want.shape
    assert np.array_equal(got, want)
    assert np.array_equal(got[:, 0], np.arange(len(pts)))   # a query that is a support finds itself first


def test_neighbors_pool_and_upsample_shapes(device, coracle):
    from d3feat_amd import tf_custom_ops as tfo
    s0 = surface_cloud(3)
    sub = coracle.grid_subsampling(s0, 0.06)
    l0, l1 = np.asarray([len(s0)], np.int32), np.asarray([len(sub)], np.int32)
    for q, s, ql, sl, r in ((sub, s0, l1, l0, 0.075), (s0, sub, l0, l1, 0.15)):
        want = coracle.batch_neighbors(q, s, ql, sl, np.float32(r))
        got = tfo.batch_ordered_neighbors(_t(q, device), _t(s, device), _t(ql, device), _t(sl, device), np.float32(r))
        assert np.array_equal(got.cpu().numpy(), want)


def test_neighbors_truncated_width_and_padding(device, coracle):
    from d3feat_amd import ops
    s0 = surface_cloud(4)
    lens = np.asarray([len(s0)], np.int32)
    want = coracle.batch_neighbors(s0, s0, lens, lens, np.float32(0.075))
    width = 20
    out, status = ops.batch_radius_neighbors(_t(s0, device), _t(s0, device), lens, lens, 0.075, width, ld=24)
    kmax = ops.check_status(status, "test")
    assert kmax == want.shape[1]
    assert np.array_equal(out[:, :width].cpu().numpy(), want[:, :width])
    # wider than Kmax: padded with the shadow index
    out2, _ = ops.batch_radius_neighbors(_t(s0, device), _t(s0, device), lens, lens, 0.075, kmax + 5)
    o2 = out2.cpu().numpy()
    assert np.array_equal(o2[:, :kmax], want) and np.all(o2[:, kmax:] == len(s0))


def test_ordered_neighbors_pad_minus_one(device, coracle):
    from d3feat_amd import tf_custom_ops as tfo
    s0 = surface_cloud(5)[:3000]
    lens = np.asarray([len(s0)], np.int32)
    want = coracle.batch_neighbors(s0, s0, lens, lens, np.float32(0.1))
    want = np.where(want == len(s0), -1, want)
    got = tfo.ordered_neighbors(_t(s0, device), _t(s0, device), 0.1).cpu().numpy()
    assert np.array_equal(got, want)


def test_neighbors_exact_ties_and_far_queries(device, coracle):
    """Lattice points give many bit-equal d2 (ties resolved by index) and queries far outside the support box."""
    from d3feat_amd import tf_custom_ops as tfo
    g = np.stack(np.meshgrid(np.arange(12), np.arange(12), np.arange(6), indexing="ij"), -1).reshape(-1, 3)
    s = (g * 0.05).astype(np.float32)
    q = np.concatenate([s[::3], np.asarray([[9, 9, 9], [-5, 0, 0], [0.3, 0.3, 0.31]], np.float32)])
    ql, sl = np.asarray([len(q)], np.int32), np.asarray([len(s)], np.int32)
    want = coracle.batch_neighbors(q, s, ql, sl, np.float32(0.11), grid=False)
    got = tfo.batch_ordered_neighbors(_t(q, device), _t(s, device), _t(ql, device), _t(sl, device), 0.11).cpu().numpy()
    assert np.array_equal(got, want)


def test_full_size_properties(device):
    """BASELINE config #2 size: structural properties that need no oracle run."""
    from d3feat_amd import ops, tf_custom_ops as tfo
    from d3feat_amd.utils.synthetic import room_fragment
    raw = room_fragment(0)
    sub = tfo.grid_subsampling(_t(raw, device), 0.03)
    n = sub.shape[0]
    assert 25000 < n < 36000
    # idempotence of the voxel set: every voxel holds exactly one barycentre -> subsampling again keeps the count
    again = tfo.grid_subsampling(sub, 0.03)
    assert again.shape[0] <= n and again.shape[0] > 0.9 * n
    pts = torch.cat([sub, sub])
    lens = np.asarray([n, n], np.int32)
    nb = tfo.batch_ordered_neighbors(pts, pts, _t(lens, device), _t(lens, device), 0.075).cpu().numpy()
    assert np.array_equal(nb[:, 0], np.arange(2 * n))
    # self-pair mirror: second half = first half + n (pad 2n stays 2n)
    a, b = nb[:n], nb[n:]
    assert np.array_equal(np.where(a == 2 * n, 2 * n, a + n), b)
    # rows sorted by distance
    p = pts.cpu().numpy()
    valid = nb < 2 * n
    d = np.where(valid, ((p[np.minimum(nb, 2 * n - 1)] - p[:, None]) ** 2).sum(-1), np.inf)
    assert np.all(np.diff(d, axis=1)[valid[:, 1:]] >= -1e-9)
    assert np.all(d[valid] < 0.075 ** 2 * 1.0001)


def test_capacity_mode_forms_of_the_subsampler_equal_the_oracle():
    """The capacity-mode subsampler (no hash table: stable radix sort, csrc/radix_sort.h; clouds of at most 16384 points: one
    workgroup per cloud out of LDS, csrc/gs_small.h) against the ORACLE and the synchronous (hash) call, bit for bit: ragged
    stacks, duplicates, one-point clouds, sizes around the sort tile, 100 clouds per stack, 4-pass keys, capacity tails, and the
    reported limits (key wider than 32 bits, cloud above its point capacity, more voxels than the LDS rounds hold)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    out = subprocess.run([sys.executable, os.path.join(here, "gs_sort_path_check.py")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "SORT-PATH-OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


@pytest.mark.parametrize("hint_scale", [0.05, 0.3, 0.7, 0.99])
def test_first_only_hint_never_changes_the_result(device, hint_scale):
    """d3f_neighbor_grid_search(first_only, nn_hint): the hint restricts the first attempt to the cells a ball of that radius
    touches; wherever the nearest support is farther than the hint (here: most queries, the hint is wrong on purpose) the full
    stencil is searched again, so column 0 equals the unhinted search and the full search's first column -- ties by index."""
    from d3feat_amd import ops
    rng = np.random.default_rng(11)
    s = (rng.random((30000, 3)) * np.asarray([2.0, 1.5, 0.4])).astype(np.float32)
    s[100:140] = s[100]                                   # exact ties: lowest index wins
    q = (rng.random((20000, 3)) * np.asarray([2.2, 1.6, 0.6]) - 0.1).astype(np.float32)
    q[:40] = s[100]
    r = np.float32(0.08)
    S, Q = torch.from_numpy(s).to(device), torch.from_numpy(q).to(device)
    sl, ql = [18000, 12000], [11000, 9000]
    grid = ops.NeighborGrid(S, sl, r)
    full, _ = grid.search(Q, ql, 96)
    plain, _ = grid.search(Q, ql, 4, first_only=True)
    hinted, st = grid.search(Q, ql, 4, first_only=True, nn_hint=float(hint_scale * r))
    torch.cuda.synchronize()
    assert st.tolist()[1] == 0
    assert torch.equal(plain[:, 0], full[:, 0])
    assert torch.equal(hinted, plain)
    assert int((plain[:, 0] == 30000).sum()) > 0          # queries without any support inside the radius are padded


def test_large_query_sets_take_the_16_lane_form_and_equal_the_oracle(device, coracle):
    """Launches of >= 100 k queries run the search kernel with 16 lanes per query (four queries per wavefront, the 64-key ordering
    network with four keys per lane; csrc/radius_neighbors.hip): four surface clouds of ~30 k points, (a) queries = supports (the
    convolution matrices), (b) another query set of the same size against them (the pooling matrices), both against the oracle
    row for row; (c) the nearest-only search (upsampling matrices) equals column 0 of the full search, with and without the
    distance hint; exact ties (duplicated supports) are ordered by index in every form."""
    from d3feat_amd import ops, tf_custom_ops as tfo
    from d3feat_amd.utils.synthetic import room_fragment
    clouds = [coracle.grid_subsampling(room_fragment(700 + i, n_raw=260000 + 15000 * i), 0.03) for i in range(4)]
    lens = np.asarray([len(c) for c in clouds], np.int32)
    s = np.concatenate(clouds).astype(np.float32)
    assert s.shape[0] >= 100000, s.shape
    s[5000:5040] = s[5000]                                 # exact ties
    r = np.float32(0.075)
    want = coracle.batch_neighbors(s, s, lens, lens, r)
    got = tfo.batch_ordered_neighbors(_t(s, device), _t(s, device), _t(lens, device), _t(lens, device), r).cpu().numpy()
    assert got.shape == want.shape and np.array_equal(got, want)
    rng = np.random.default_rng(5)
    q = (s + rng.normal(0.0, 0.02, s.shape)).astype(np.float32)
    want_q = coracle.batch_neighbors(q, s, lens, lens, r)
    got_q = tfo.batch_ordered_neighbors(_t(q, device), _t(s, device), _t(lens, device), _t(lens, device), r).cpu().numpy()
    assert got_q.shape == want_q.shape and np.array_equal(got_q, want_q)
    grid = ops.NeighborGrid(_t(s, device), lens.tolist(), r)
    full, _ = grid.search(_t(q, device), lens.tolist(), want_q.shape[1])
    plain, _ = grid.search(_t(q, device), lens.tolist(), 1, first_only=True)
    hinted, _ = grid.search(_t(q, device), lens.tolist(), 1, first_only=True, nn_hint=float(0.6 * r))
    torch.cuda.synchronize()
    assert np.array_equal(full.cpu().numpy(), want_q)
    assert torch.equal(plain[:, 0], full[:, 0]) and torch.equal(hinted, plain)
