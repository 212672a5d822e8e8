"""GPU parity of the network kernels against the torch-CPU restatement of the reference graph
(oracle/network_np.py).  Floating point: max |diff| <= 1e-4 (BASELINE.json north_star), on activations whose
scale is O(1); looser relative bound where the magnitude grows."""
import numpy as np
import pytest
import torch

from conftest import surface_cloud

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _close(got, want, tol=TOL, relative=False):
    """max |got - want| <= tol, ABSOLUTE (the north star's bar).  relative=True scales by max(1, |want|_inf): only for the
    raw contraction tests, whose random operands produce outputs of arbitrary magnitude (stated at the call site)."""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape
    scale = max(1.0, np.abs(want).max()) if relative else 1.0
    err = np.abs(got - want).max()
    assert err <= tol * scale, "max abs err %.3e (|want|max %.3g)" % (err, np.abs(want).max())


@pytest.mark.parametrize("M,K,N", [(1000, 64, 32), (777, 15, 64), (4097, 128, 256), (390, 7680, 512), (200, 1024, 2048),
                                   (64, 3072, 512), (5, 3, 7), (70001, 96, 128), (66000, 64, 32)])
def test_gemm_epilogues(device, M, K, N):
    from d3feat_amd import ops
    rng = np.random.default_rng(M + K + N)
    A = rng.standard_normal((M, K)).astype(np.float32)
    B = (rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32)
    rs = rng.random(M).astype(np.float32) + 0.5
    cs = rng.random(N).astype(np.float32) + 0.5
    ch = rng.standard_normal(N).astype(np.float32)
    res = rng.standard_normal((M, N)).astype(np.float32)
    ref = A.astype(np.float64) @ B.astype(np.float64)
    got = ops.gemm(_t(A, device), _t(B, device)).cpu().numpy()
    _close(got, ref, 2e-5, relative=True)      # raw contraction, random operands
    full = ref * rs[:, None] * cs + ch + res
    full = np.where(full > 0, full, 0.2 * full)
    got = ops.gemm(_t(A, device), _t(B, device), _t(rs, device), _t(cs, device), _t(ch, device), _t(res, device), True, 0.2)
    _close(got.cpu().numpy(), full, 2e-5, relative=True)


@pytest.mark.parametrize("M,K,N,real", [(200, 4096, 128, 0), (5000, 1000, 192, 0), (70000, 160, 64, 0), (33, 8, 4, 0),
                                        (9000, 512, 128, 6100), (300000, 256, 64, 171000), (1580, 7680, 512, 1200)])
def test_gemm_split_plans_and_capacity_rows(device, M, K, N, real):
    """Contraction shapes whose K range is split into slabs (few tiles, long K), not split (K <= 512), with a device-resident row
    count below the capacity (rows past it are never written): results vs fp64, bit-identical from call to call (the slabs are
    reduced in a fixed order), every epilogue.  (Written for the persistent stream-K schedule of experiment x10 -- parked as
    tools/ubench/gemm_streamk_pipelined.patch -- and kept: it covers the capacity-mode rows of the production kernel too.)"""
    from d3feat_amd import ops
    rng = np.random.default_rng(M + K + N)
    A = rng.standard_normal((M, K)).astype(np.float32)
    B = (rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32)
    rs = rng.random(M).astype(np.float32) + 0.5
    cs = rng.random(N).astype(np.float32) + 0.5
    ch = rng.standard_normal(N).astype(np.float32)
    res = rng.standard_normal((M, N)).astype(np.float32)
    n = real or M
    ref = A[:n].astype(np.float64) @ B.astype(np.float64)
    tA, tB = _t(A, device), _t(B, device)
    if real:
        tA.n_dev = torch.tensor([real], dtype=torch.int32, device=device)
        tA.n_hint = real
    sentinel = 12345.0
    outs = []
    for rep in range(3):
        out = torch.full((M, N), sentinel, dtype=torch.float32, device=device)
        ops.gemm(tA, tB, out=out)
        outs.append(out)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    _close(outs[0][:n].cpu().numpy(), ref, 2e-5, relative=True)
    assert bool((outs[0][n:] == sentinel).all())
    full = ref * rs[:n, None] * cs + ch + res[:n]
    full = np.where(full > 0, full, 0.2 * full)
    for kw in (dict(row_scale=_t(rs, device)), dict(residual=_t(res, device)), dict(row_scale=_t(rs, device), residual=_t(res, device))):
        want = ref * (rs[:n, None] if "row_scale" in kw else 1.0) * cs + ch + (res[:n] if "residual" in kw else 0.0)
        want = np.where(want > 0, want, 0.2 * want)
        got = ops.gemm(tA, tB, col_scale=_t(cs, device), col_shift=_t(ch, device), leaky=True, alpha=0.2, **kw)
        _close(got[:n].cpu().numpy(), want, 2e-5, relative=True)


def test_gemm_strided_views(device):
    from d3feat_amd import ops
    rng = np.random.default_rng(0)
    big = rng.standard_normal((300, 96)).astype(np.float32)
    B = rng.standard_normal((40, 24)).astype(np.float32)
    tb = _t(big, device)
    got = ops.gemm(tb[:, 8:48], _t(B, device)).cpu().numpy()      # lda = 96, base not 16B-multiple of row
    _close(got, big[:, 8:48].astype(np.float64) @ B, 2e-5, relative=True)


def _layer_case(seed, cin, cout, strided, layer=0):
    from d3feat_amd.utils.config import threedmatch_config
    cfg = threedmatch_config()
    s0 = surface_cloud(seed, n_raw=40000)
    rng = np.random.default_rng(seed)
    return cfg, s0, rng


@pytest.mark.parametrize("cin,cout", [(1, 64), (32, 32), (64, 64), (128, 128), (256, 256), (512, 512), (6, 10)])
@pytest.mark.parametrize("strided", [False, True])
def test_kpconv_vs_oracle(device, coracle, cin, cout, strided):
    from d3feat_amd.kernels import convolution_ops as conv_ops
    from d3feat_amd.kernels.kernel_points import create_kernel_points
    from oracle import network_np as onp
    if cin >= 256 and strided:
        pytest.skip("covered by the non-strided case (same kernels)")
    s0 = surface_cloud(cin + cout, n_raw=30000 if cin < 256 else 12000)
    rng = np.random.default_rng(cin * 7 + cout)
    lens = np.asarray([len(s0)], np.int32)
    if strided:
        q = coracle.grid_subsampling(s0, 0.06)
        nb = coracle.batch_neighbors(q, s0, np.asarray([len(q)], np.int32), lens, np.float32(0.075))
    else:
        q = s0
        nb = coracle.batch_neighbors(s0, s0, lens, lens, np.float32(0.075))
    nb = nb[:, :37]
    f = rng.standard_normal((len(s0), cin)).astype(np.float32)
    if cin == 1:
        f = np.ones_like(f)
    W = (rng.standard_normal((15, cin, cout)) * np.sqrt(2.0 / cout)).astype(np.float32)
    KP = create_kernel_points(1.5 * 0.03, 15, 1, 3, 'center', rng=np.random.default_rng(1)).reshape(15, 3).astype(np.float32)
    want = onp.KPConv_ops(q, s0, nb, f, KP, W, 0.03, 'linear', 'sum').numpy()
    got = conv_ops.KPConv_ops(_t(q, device), _t(s0, device), _t(nb.astype(np.int32), device), _t(f, device), KP,
                              _t(W, device), 0.03, 'linear', 'sum').cpu().numpy()
    _close(got, want)


@pytest.mark.parametrize("influence,mode", [("constant", "sum"), ("gaussian", "sum"), ("linear", "closest")])
def test_kpconv_modes(device, coracle, influence, mode):
    from d3feat_amd.kernels import convolution_ops as conv_ops
    from d3feat_amd.kernels.kernel_points import create_kernel_points
    from oracle import network_np as onp
    s0 = surface_cloud(9, n_raw=15000)
    rng = np.random.default_rng(3)
    lens = np.asarray([len(s0)], np.int32)
    nb = coracle.batch_neighbors(s0, s0, lens, lens, np.float32(0.075))[:, :30]
    f = rng.standard_normal((len(s0), 16)).astype(np.float32)
    W = (rng.standard_normal((15, 16, 8)) * 0.3).astype(np.float32)
    KP = create_kernel_points(0.045, 15, 1, 3, 'center', rng=np.random.default_rng(1)).reshape(15, 3).astype(np.float32)
    want = onp.KPConv_ops(s0, s0, nb, f, KP, W, 0.03, influence, mode).numpy()
    got = conv_ops.KPConv_ops(_t(s0, device), _t(s0, device), _t(nb.astype(np.int32), device), _t(f, device), KP,
                              _t(W, device), 0.03, influence, mode).cpu().numpy()
    _close(got, want)


def test_pools_vs_oracle(device, coracle):
    from d3feat_amd import ops
    from oracle import network_np as onp
    s0 = surface_cloud(11, n_raw=20000)
    sub = coracle.grid_subsampling(s0, 0.06)
    l0, l1 = np.asarray([len(s0)], np.int32), np.asarray([len(sub)], np.int32)
    pool_i = coracle.batch_neighbors(sub, s0, l1, l0, np.float32(0.05))[:, :20]    # small radius -> some all-shadow rows
    up_i = coracle.batch_neighbors(s0, sub, l0, l1, np.float32(0.07))[:, :20]
    rng = np.random.default_rng(0)
    x = rng.standard_normal((len(s0), 128)).astype(np.float32)
    y = rng.standard_normal((len(sub), 256)).astype(np.float32)
    want = onp.ind_max_pool(torch.from_numpy(x), pool_i).numpy()
    got = ops.ind_max_pool(_t(x, device), _t(pool_i.astype(np.int32), device)).cpu().numpy()
    assert np.array_equal(got, want)
    assert not (pool_i >= len(s0)).all(1).any()      # every row has a valid neighbour: the column minima are never computed
    # a radius below the voxel size leaves pooled points without any neighbour: those rows ARE the shadow row (column
    # minima, computed lazily); also the degenerate all-shadow index matrix
    pool_s = coracle.batch_neighbors(sub, s0, l1, l0, np.float32(0.02))[:, :7]
    assert (pool_s >= len(s0)).all(1).sum() > 50
    for inds in (pool_s, np.full((9, 3), len(s0), np.int32)):
        want = onp.ind_max_pool(torch.from_numpy(x), inds).numpy()
        got = ops.ind_max_pool(_t(x, device), _t(inds.astype(np.int32), device)).cpu().numpy()
        assert np.array_equal(got, want)
    want = torch.cat([onp.closest_pool(torch.from_numpy(y), up_i), torch.from_numpy(x)], 1).numpy()
    got = ops.closest_pool_cat(_t(y, device), _t(up_i.astype(np.int32), device), _t(x, device)).cpu().numpy()
    assert np.array_equal(got, want)


@pytest.mark.parametrize("lens", [(0.5, 0.5), (0.7, 0.3)])
def test_detection_head_vs_oracle(device, coracle, lens):
    from d3feat_amd import ops
    from d3feat_amd.models.D3Feat import detection_head
    from oracle import network_np as onp
    s0 = surface_cloud(13, n_raw=20000)
    n = len(s0)
    n0 = int(n * lens[0])
    L = np.asarray([n0, n - n0], np.int32)
    nb = coracle.batch_neighbors(s0, s0, L, L, np.float32(0.075))[:, :35]
    rng = np.random.default_rng(1)
    x = (rng.standard_normal((n, 32)) * 2).astype(np.float32)
    in_b = onp.stack_batch_inds(L)
    want_s = onp.detection_head(torch.from_numpy(x), nb, in_b, L).numpy()
    xt = torch.from_numpy(x)
    want_d = (xt * torch.rsqrt(torch.clamp((xt ** 2).sum(1, keepdim=True), min=1e-10))).numpy()
    inputs = dict(neighbors=[_t(nb.astype(np.int32), device)], stack_lengths=_t(L, device))
    desc, score = detection_head(_t(x, device), inputs)
    _close(desc.cpu().numpy(), want_d, 1e-5)
    _close(score.cpu().numpy(), want_s)


def test_detection_head_stack_groups_all_negative(device, coracle):
    """A batched engine stack holds several reference stacks (pairs).  The per-cloud maximum of models/D3Feat.py:84-85 includes
    the zero shadow row iff the cloud's in_batches row is padded (datasets/common.py:453-496) -- a rule about the cloud's OWN
    pair.  With all-negative last_unary outputs the zero row IS the maximum wherever it is included, so the grouping is
    visible: every pair of the 6-cloud stack must equal the oracle head run on that pair alone (equal lengths: both clouds
    padded; unequal: only the shorter one), whatever the lengths of its stack mates."""
    from d3feat_amd import ops
    from oracle import network_np as onp
    s0 = surface_cloud(17, n_raw=24000)
    n = len(s0)
    cuts = [0, n // 6, 2 * (n // 6), 2 * (n // 6) + n // 4, 2 * (n // 6) + n // 4 + n // 10, n - n // 5, n]
    L = np.diff(cuts).astype(np.int32)           # pairs: (equal, equal), (long, short), (x, y)
    assert L[0] == L[1] and L[2] != L[3]
    nb = coracle.batch_neighbors(s0, s0, L, L, np.float32(0.075))[:, :30]
    rng = np.random.default_rng(2)
    x = (-np.abs(rng.standard_normal((n, 32))) - 0.05).astype(np.float32)          # every entry negative
    desc, score = ops.detect_head(_t(x, device), _t(nb.astype(np.int32), device), _t(L, device), None, stack_group=2)
    score = score.cpu().numpy()
    for g in range(3):
        a, b = cuts[2 * g], cuts[2 * g + 2]
        Lp = L[2 * g:2 * g + 2]
        nbp = nb[a:b].astype(np.int64)
        nbp = np.where(nbp >= n, b - a, nbp - a)                                    # re-based to the pair
        want = onp.detection_head(torch.from_numpy(x[a:b]), nbp, onp.stack_batch_inds(Lp), Lp).numpy()
        # (a padded cloud's maximum is the zero row: y = x / 1e-6 -- scores of order 1e7, hence the relative bound here)
        _close(score[a:b], want, relative=True)
    # and the whole-stack rule (group 0) differs from it on this input: the grouping is not vacuous
    _, score0 = ops.detect_head(_t(x, device), _t(nb.astype(np.int32), device), _t(L, device), None, stack_group=0)
    assert np.abs(score0.cpu().numpy() - score).max() > 1.0


def _pair_inputs(coracle, cfg, cloud, limits):
    from oracle import network_np as onp
    pts = np.concatenate([cloud, cloud])
    lens = np.asarray([len(cloud)] * 2, np.int32)
    return onp.descriptor_input(cfg, pts, np.ones((len(pts), 1), np.float32), lens, limits,
                                lambda q, s, ql, sl, r: coracle.batch_neighbors(q, s, ql, sl, r),
                                lambda p, l, dl: coracle.batch_grid_subsampling(p, l, dl))


def test_pyramid_vs_oracle(device, coracle):
    """tf_descriptor_input on the GPU == the oracle pyramid, matrix by matrix, bit for bit (exact shapes);
    the fast (padded) variant equals it on the valid columns and holds the shadow index elsewhere."""
    from d3feat_amd.datasets.common import FragmentDataset
    from d3feat_amd.utils.config import threedmatch_config
    cfg = threedmatch_config()
    cloud = surface_cloud(21, n_raw=60000)
    limits = np.asarray([37, 35, 36, 38, 38], np.int32)
    want = _pair_inputs(coracle, cfg, cloud, limits)
    for fast in (False, True):
        ds = FragmentDataset([cloud], fast=fast)
        ds.neighborhood_limits = limits
        gen, _, _ = ds.get_batch_gen('test', cfg)
        flat = ds.get_tf_mapping(cfg)(*ds._to_device(next(iter(gen()))))
        L = cfg.num_layers
        for l in range(L):
            assert np.array_equal(flat[l].cpu().numpy().view(np.uint32), want['points'][l].view(np.uint32))
            for name, off in (('neighbors', L), ('pools', 2 * L), ('upsamples', 3 * L)):
                g, w = flat[off + l].cpu().numpy(), want[name][l]
                if w.shape[0] == 0:
                    assert g.shape[0] == 0
                    continue
                if not fast:
                    assert np.array_equal(g, w), (name, l)
                elif name == 'upsamples':
                    assert np.array_equal(g[:, 0], w[:, 0])
                else:
                    pad = want['points'][l].shape[0]     # shadow index = number of supports (the layer's points)
                    assert g.shape[1] == limits[l]
                    assert np.array_equal(g[:, :w.shape[1]], w), (name, l)
                    assert np.all(g[:, w.shape[1]:] == pad)
        if not fast:   # the fast path does not build the index matrices the network never reads
            assert np.array_equal(flat[4 * L + 2].cpu().numpy(), want['in_batches'])
            assert np.array_equal(flat[4 * L + 3].cpu().numpy(), want['out_batches'])
            assert np.allclose(flat[4 * L + 1].cpu().numpy(), want['batch_weights'])


def test_calibrate_neighbors_vs_oracle(device, coracle):
    from d3feat_amd.datasets.common import FragmentDataset
    from d3feat_amd.utils.config import threedmatch_config
    from oracle import network_np as onp
    cfg = threedmatch_config()
    clouds = [surface_cloud(31, n_raw=60000), surface_cloud(32, n_raw=50000)]
    ds = FragmentDataset(clouds)
    hist_n = onp.hist_size(cfg)
    ds.neighborhood_limits = np.full(cfg.num_layers, hist_n, np.int32)
    ds.calibrate_neighbors(cfg, samples_threshold=10 ** 9)
    hists = np.zeros((cfg.num_layers, hist_n), np.int64)
    for c in clouds:
        inp = _pair_inputs(coracle, cfg, c, np.full(cfg.num_layers, hist_n, np.int32))
        hists += onp.neighbor_histograms(inp['neighbors'], hist_n)
    assert np.array_equal(ds.neighborhood_limits, onp.limits_from_histograms(hists))


def test_full_forward_vs_oracle(device, coracle):
    """Whole KPFCNN forward (pyramid from the oracle so that only the network is under test, then end to end)."""
    from d3feat_amd.datasets.common import FragmentDataset
    from d3feat_amd.models.KPFCNN_model import KernelPointFCNN
    from d3feat_amd.models.variables import build_variables
    from d3feat_amd.utils.config import threedmatch_config
    from oracle import network_np as onp
    cfg = threedmatch_config()
    cloud = surface_cloud(41, n_raw=50000)
    limits = np.asarray([37, 35, 36, 38, 38], np.int32)
    W = build_variables(cfg, seed=42, randomize_bn=True).values
    inp = _pair_inputs(coracle, cfg, cloud, limits)
    want_d, want_s = onp.forward(cfg, W, inp)
    ds = FragmentDataset([cloud], fast=True)
    ds.neighborhood_limits = limits
    gen, _, _ = ds.get_batch_gen('test', cfg)
    flat = ds.get_tf_mapping(cfg)(*ds._to_device(next(iter(gen()))))
    model = KernelPointFCNN(flat, cfg, weights=W)
    d, s = model.out_features.cpu().numpy(), model.out_scores.cpu().numpy()
    assert d.shape == want_d.shape == (2 * len(cloud), 32) and s.shape == want_s.shape
    assert np.isfinite(d).all() and np.isfinite(s).all()
    _close(d, want_d)
    _close(s, want_s)
    # exact-shape inputs give the same result as the padded fast path
    ds2 = FragmentDataset([cloud], fast=False)
    ds2.neighborhood_limits = limits
    flat2 = ds2.get_tf_mapping(cfg)(*ds2._to_device(next(iter(gen()))))
    m2 = KernelPointFCNN(flat2, cfg, weights=W)
    assert torch.equal(m2.out_features, model.out_features) and torch.equal(m2.out_scores, model.out_scores)


@pytest.mark.parametrize("C1,C2,N", [(128, 64, 64), (1024, 2048, 512), (64, 0, 32)])
def test_gemm_upsample_cat_equals_materialised(device, C1, C2, N):
    """The fused decoder contraction ([gathered | skip] @ W) is the same arithmetic as gather+concat followed by the GEMM:
    identical k order -> bit-identical results; shadow indices read the zero row."""
    from d3feat_amd import ops
    rng = np.random.default_rng(C1 + C2)
    n1, m = 700, 2500
    x = _t(rng.standard_normal((n1, C1)).astype(np.float32), device)
    skip = _t(rng.standard_normal((m, C2)).astype(np.float32), device) if C2 else None
    idx = rng.integers(0, n1 + 1, (m, 3)).astype(np.int32)        # n1 = shadow index
    idx[::17, 0] = n1
    W = _t((rng.standard_normal((C1 + C2, N)) / np.sqrt(C1 + C2)).astype(np.float32), device)
    cs = _t(rng.random(N).astype(np.float32) + 0.5, device)
    ch = _t(rng.standard_normal(N).astype(np.float32), device)
    u = ops.UpsampleCat(x, _t(idx, device), skip)
    assert u.shape == (m, C1 + C2)
    got = ops.gemm_upsample_cat(u, W, col_scale=cs, col_shift=ch, leaky=True)
    want = ops.gemm(u.materialize(), W, col_scale=cs, col_shift=ch, leaky=True)
    assert torch.equal(got, want)
    ref = np.concatenate([np.concatenate([x.cpu().numpy(), np.zeros((1, C1), np.float32)])[idx[:, 0]]] +
                         ([skip.cpu().numpy()] if C2 else []), 1).astype(np.float64) @ W.cpu().numpy().astype(np.float64)
    ref = ref * cs.cpu().numpy() + ch.cpu().numpy()
    ref = np.where(ref > 0, ref, 0.2 * ref)
    _close(got.cpu().numpy(), ref, 2e-5, relative=True)


def test_kpconv_fused_256_equals_the_two_kernel_form(device):
    """The one-kernel KPConv exists for Cin = Cout = 256 too (1024-thread workgroups); KPConv_ops does not choose it (it measured
    no faster than aggregation + contraction), so it is exercised directly: same result as the two-kernel form to fp32 summation
    order, on real neighbour lists."""
    from d3feat_amd import ops
    from conftest import surface_cloud
    rng = np.random.default_rng(256)
    s0 = surface_cloud(256, n_raw=12000)
    pts = torch.from_numpy(s0).to(device)
    lens = [len(s0)]
    nb = ops.batch_radius_neighbors(pts, pts, lens, lens, 0.075, 40)[0]
    f = torch.from_numpy(rng.standard_normal((len(s0), 256)).astype(np.float32)).to(device)
    W = torch.from_numpy((rng.standard_normal((15, 256, 256)) / 60).astype(np.float32)).to(device)
    kp = (rng.standard_normal((15, 3)) * 0.02).astype(np.float32)
    kp[0] = 0
    assert ops.kpconv_fused_supported(256, 256, 15, "linear", "sum", available=True)
    assert not ops.kpconv_fused_supported(256, 256, 15, "linear", "sum")
    got = ops.kpconv_fused(pts, pts, nb, f, kp, W, 0.03, leaky=True)
    wf, inv = ops.kpconv_aggregate(pts, pts, nb, f, kp, 0.03)
    want = ops.gemm(wf, W.reshape(15 * 256, 256), row_scale=inv, leaky=True)
    torch.cuda.synchronize()
    assert (got - want).abs().max().item() <= 1e-4 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("cin", [32, 64, 128, 256])
def test_kpconv_shadow_slots_do_not_see_a_non_finite_row_zero(device, coracle, cin):
    """A shadow neighbour contributes the reference's ZERO feature row (kernels/convolution_ops.py:234).  The aggregation kernels
    fetch it as an out-of-range buffer load (exact zeros from the hardware's range check) -- not as row 0 times a zero influence,
    which turned every query with a shadow slot into NaN as soon as row 0 held Inf / NaN.  With NaN and Inf planted in
    row 0 the output is non-finite exactly where the reference's is: at the queries that have row 0 as a REAL neighbour."""
    from d3feat_amd.kernels import convolution_ops as conv_ops
    from d3feat_amd.kernels.kernel_points import create_kernel_points
    from oracle import network_np as onp
    s0 = surface_cloud(300 + cin, n_raw=20000)
    rng = np.random.default_rng(cin)
    lens = np.asarray([len(s0)], np.int32)
    nb = coracle.batch_neighbors(s0, s0, lens, lens, np.float32(0.075))[:, :37]
    assert (nb == len(s0)).any()                                   # shadow slots exist
    f = rng.standard_normal((len(s0), cin)).astype(np.float32)
    f[0, 0], f[0, 1] = np.nan, np.inf
    W = (rng.standard_normal((15, cin, cin)) * np.sqrt(2.0 / cin)).astype(np.float32)
    KP = create_kernel_points(0.045, 15, 1, 3, 'center', rng=np.random.default_rng(1)).reshape(15, 3).astype(np.float32)
    with np.errstate(invalid="ignore"):
        want = onp.KPConv_ops(s0, s0, nb, f, KP, W, 0.03, 'linear', 'sum').numpy()
    got = conv_ops.KPConv_ops(_t(s0, device), _t(s0, device), _t(nb.astype(np.int32), device), _t(f, device), KP,
                              _t(W, device), 0.03, 'linear', 'sum').cpu().numpy()
    touched = (nb == 0).any(1)                                     # queries with row 0 among their real neighbours
    assert touched.sum() < 100 and np.isfinite(want[~touched]).all()
    assert np.isfinite(got[~touched]).all(), int((~np.isfinite(got[~touched])).any(1).sum())
    _close(got[~touched], want[~touched])
    assert not np.isfinite(got[touched]).all(1).any()             # where the reference is non-finite, so is the kernel
