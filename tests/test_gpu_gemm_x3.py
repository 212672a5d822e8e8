"""The operand-split contraction (d3f_gemm_x3, csrc/gemm_x3.h): fp32 operands written exactly as three bfloat16 planes, six exact
bf16 products per fp32 product on v_mfma_f32_32x32x16_bf16, fp32 accumulate.  Claims checked here:
  * the split is EXACT: products that fp32 can hold exactly come out exactly (all three planes of either operand are exercised);
  * against float64 its error is of the size of the fp32 MFMA kernel's (d3f_gemm_f32t), measured side by side on the same operands;
  * it is the same operator: epilogue, gathered / concatenated operands, ragged M / N, K split, device-resident row counts.
Which problems run as 128 x 32, 128 x 64 and 256 x 128 workgroups is a host decision: tests/test_cabi.py::test_gemm_x3_plans (CPU).
The network-level statement (same descriptors as the reference's Python at a few 1e-6) is tests/test_gpu_golden_network.py, which
runs on this kernel since it is the default."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


class _fp32_mfma:
    """The same call on the fp32 MFMA kernel (d3f_gemm_f32t)."""

    def __enter__(self):
        from d3feat_amd import ops
        self.ops, self.prev = ops, ops.GEMM_X3
        ops.GEMM_X3 = False

    def __exit__(self, *a):
        self.ops.GEMM_X3 = self.prev


@pytest.fixture(autouse=True)
def _x3_for_every_width(monkeypatch):
    """The host keeps 32-column layers on the fp32 MFMA kernel (faster there); this module exercises d3f_gemm_x3 at every width."""
    from d3feat_amd import ops
    monkeypatch.setattr(ops, "_x3_ok", lambda C1, C2, N, rows=0: ops.GEMM_X3 and (C1 + C2) % 32 == 0 and (C2 == 0 or C1 % 32 == 0))


def test_x3_is_the_default_contraction(monkeypatch):
    monkeypatch.undo()
    from d3feat_amd import ops
    assert ops.GEMM_X3 and ops._x3_ok(64, 0, 64) and ops._x3_ok(128, 64, 128) and not ops._x3_ok(48, 0, 64) and not ops._x3_ok(16, 48, 64)
    assert not ops._x3_ok(64, 0, 32)        # 32-column layers: the fp32 MFMA kernel is the faster one ...
    assert ops._x3_ok(64, 0, 32, 100000)    # ... below the row count of the resident-W persistent form


@pytest.mark.parametrize("M,K,N", [(1000, 64, 64), (333, 96, 32), (4100, 128, 100), (130, 2048, 36)])
def test_all_three_planes_of_the_activations_are_exact(device, M, K, N):
    """W = a 0/1 selection matrix: every output is ONE activation value, which must come back bit for bit -- 24-bit mantissas, i.e.
    a1 + a2 + a3 reassembled by the fp32 accumulator (a1 w, a2 w, a3 w are the only non-zero products)."""
    from d3feat_amd import ops
    rng = np.random.default_rng(M + N)
    A = (rng.standard_normal((M, K)) * np.exp(rng.uniform(-20, 20, (M, K)))).astype(np.float32)
    sel = rng.integers(0, K, N)
    W = np.zeros((K, N), np.float32)
    W[sel, np.arange(N)] = 1.0
    got = ops.gemm(_t(A, device), _t(W, device)).cpu().numpy()
    assert np.array_equal(got.view(np.uint32), A[:, sel].view(np.uint32))


@pytest.mark.parametrize("M,K,N", [(500, 64, 64), (129, 256, 128)])
def test_all_three_planes_of_the_weights_are_exact(device, M, K, N):
    """A = one-hot rows: every output is ONE weight value, bit for bit (w1 + w2 + w3 of the pre-split copy)."""
    from d3feat_amd import ops
    rng = np.random.default_rng(M + K)
    W = (rng.standard_normal((K, N)) * np.exp(rng.uniform(-20, 20, (K, N)))).astype(np.float32)
    sel = rng.integers(0, K, M)
    A = np.zeros((M, K), np.float32)
    A[np.arange(M), sel] = 1.0
    got = ops.gemm(_t(A, device), _t(W, device)).cpu().numpy()
    assert np.array_equal(got.view(np.uint32), W[sel].view(np.uint32))


def test_the_small_edge_of_the_format(device, capsys):
    """Operands of 2^-126 .. 2^-90 (VERDICT r04 weak 1 iii): the third plane of a value below ~2^-110 is a bfloat16 SUBNORMAL (and
    below 2^-118 so is the second), which a matrix pipe may flush.  Selection weights make every output ONE operand: whatever the
    hardware does with subnormal planes, the result is within 2^-126 ABSOLUTE of the operand -- 2^-8 of the smallest normal
    fp32 number, far below anything an activation of this network can resolve -- and what it does is printed (and recorded in
    csrc/gemm_x3.h).  Values of 2^-100 and above (normal planes) must come through bit for bit."""
    from d3feat_amd import ops
    rng = np.random.default_rng(126)
    M, K, N = 2000, 128, 64
    e = rng.integers(-126, -89, (M, K))
    A = (np.ldexp(1.0 + rng.random((M, K)), e) * rng.choice([-1.0, 1.0], (M, K))).astype(np.float32)
    sel = rng.integers(0, K, N)
    W = np.zeros((K, N), np.float32)
    W[sel, np.arange(N)] = 1.0
    got = ops.gemm(_t(A, device), _t(W, device)).cpu().numpy()
    want = A[:, sel]
    err = np.abs(got.astype(np.float64) - want.astype(np.float64))
    small = np.abs(want) < 2.0 ** -100
    exact_small = bool(np.array_equal(got[small].view(np.uint32), want[small].view(np.uint32)))
    with capsys.disabled():
        print("\n[x3 small edge] operands below 2^-100 reproduced bit for bit: %s; max abs error %.3e (2^%.1f), flushed to zero: %d of %d"
              % (exact_small, err.max(), np.log2(max(err.max(), 2.0 ** -200)), int(((got == 0) & (want != 0)).sum()), want.size))
    assert np.array_equal(got[~small].view(np.uint32), want[~small].view(np.uint32))
    assert err.max() <= 2.0 ** -126


def test_integer_products_are_exact(device):
    """12-bit integers x 5-bit integers over K = 128: every product and every partial sum is an integer below 2^24 -- the result is
    exact whatever the summation order (the activations span two planes)."""
    from d3feat_amd import ops
    rng = np.random.default_rng(5)
    A = rng.integers(-2047, 2048, (3000, 128)).astype(np.float32)
    W = rng.integers(-15, 16, (128, 64)).astype(np.float32)
    got = ops.gemm(_t(A, device), _t(W, device)).cpu().numpy()
    want = A.astype(np.int64) @ W.astype(np.int64)
    assert np.array_equal(got.astype(np.int64), want)


@pytest.mark.parametrize("M,K,N", [(20000, 64, 32), (20000, 128, 64), (9000, 256, 128), (3000, 1024, 256), (900, 3840, 256),
                                   (200, 7680, 512), (20001, 1024, 256), (40000, 512, 512)])
def test_error_against_float64_beside_the_fp32_mfma_kernel(device, M, K, N):
    """max |C - C64| / max |C64| of both kernels on the same operands: the split form may not be worse than twice the fp32 MFMA
    kernel's error + 2^-23 (its dropped terms), and both stay below 4e-6."""
    from d3feat_amd import ops
    rng = np.random.default_rng(K + N)
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32)
    ref = A.astype(np.float64) @ W.astype(np.float64)
    tA, tW = _t(A, device), _t(W, device)
    x3 = ops.gemm(tA, tW).cpu().numpy()
    with _fp32_mfma():
        f32 = ops.gemm(tA, tW).cpu().numpy()
    scale = np.abs(ref).max()
    e_x3, e_f32 = np.abs(x3 - ref).max() / scale, np.abs(f32 - ref).max() / scale
    print("M %d K %d N %d: x3 %.3e  fp32 mfma %.3e" % (M, K, N, e_x3, e_f32))
    assert e_x3 <= 2.0 * e_f32 + 2.0 ** -23 and e_x3 <= 4e-6 and e_f32 <= 4e-6


@pytest.mark.parametrize("M,K,N,real", [(777, 64, 36, 0), (4097, 128, 100, 0), (260, 4096, 128, 0), (70001, 96, 128, 0),
                                        (140000, 256, 64, 0), (100003, 64, 32, 0), (131072, 32, 128, 66000), (90000, 128, 20, 0),
                                        (9000, 512, 128, 6100), (1580, 7680, 512, 1200), (5, 32, 4, 0), (20001, 1024, 200, 0),
                                        (33000, 512, 256, 24000)])
def test_same_operator_every_epilogue_ragged_shapes_row_counts(device, M, K, N, real):
    from d3feat_amd import ops
    rng = np.random.default_rng(M + K + N)
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32)
    rs = rng.random(M).astype(np.float32) + 0.5
    cs = rng.random(N).astype(np.float32) + 0.5
    ch = rng.standard_normal(N).astype(np.float32)
    res = rng.standard_normal((M, N)).astype(np.float32)
    n = real or M
    ref = A[:n].astype(np.float64) @ W.astype(np.float64)
    tA, tW = _t(A, device), _t(W, device)
    if real:
        tA.n_dev = torch.tensor([real], dtype=torch.int32, device=device)
        tA.n_hint = real
    outs = []
    for _ in range(2):
        out = torch.full((M, N), 12345.0, dtype=torch.float32, device=device)
        ops.gemm(tA, tW, out=out)
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    assert np.abs(outs[0][:n].cpu().numpy() - ref).max() <= 4e-6 * max(1.0, np.abs(ref).max())
    assert bool((outs[0][n:] == 12345.0).all())
    for kw in (dict(row_scale=_t(rs, device)), dict(residual=_t(res, device)), dict(row_scale=_t(rs, device), residual=_t(res, device))):
        want = ref * (rs[:n, None] if "row_scale" in kw else 1.0) * cs + ch + (res[:n] if "residual" in kw else 0.0)
        want = np.where(want > 0, want, 0.2 * want)
        got = ops.gemm(tA, tW, col_scale=_t(cs, device), col_shift=_t(ch, device), leaky=True, alpha=0.2, **kw)[:n].cpu().numpy()
        assert np.abs(got - want).max() <= 4e-6 * max(1.0, np.abs(want).max())
        with _fp32_mfma():
            f32 = ops.gemm(tA, tW, col_scale=_t(cs, device), col_shift=_t(ch, device), leaky=True, alpha=0.2, **kw)[:n].cpu().numpy()
        assert np.abs(got - f32).max() <= 4e-6 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("C1,C2,N,m", [(128, 64, 64, 2500), (1024, 2048, 512, 2500), (64, 0, 32, 2500), (32, 32, 128, 2500),
                                       (128, 128, 256, 30000), (32, 64, 128, 70001), (128, 0, 64, 66000), (64, 0, 32, 90003),
                                       (32, 32, 32, 70000)])
def test_gathered_and_concatenated_operands(device, C1, C2, N, m):
    """[ x'[idx[m, 0]] | skip[m] ] @ W: shadow / out-of-range indices read the zero row; the same bits as the materialised operand.
    (The fifth case runs as 256 x 128 workgroups; the cases of >= 65536 rows take the resident-W persistent form of round 5 --
    gathered rows, a concatenation boundary inside the walk, 128 / 64 / 32 columns, a ragged last tile.)"""
    from d3feat_amd import ops
    rng = np.random.default_rng(C1 + C2 + N)
    n1 = 700
    x = rng.standard_normal((n1, C1)).astype(np.float32)
    skip = rng.standard_normal((m, C2)).astype(np.float32) if C2 else None
    idx = rng.integers(0, n1 + 1, (m, 3)).astype(np.int32)
    idx[::17, 0] = n1
    W = (rng.standard_normal((C1 + C2, N)) / np.sqrt(C1 + C2)).astype(np.float32)
    cs, ch = rng.random(N).astype(np.float32) + 0.5, rng.standard_normal(N).astype(np.float32)
    u = ops.UpsampleCat(_t(x, device), _t(idx, device), _t(skip, device) if C2 else None)
    tW, tcs, tch = _t(W, device), _t(cs, device), _t(ch, device)
    got = ops.gemm_upsample_cat(u, tW, col_scale=tcs, col_shift=tch, leaky=True)
    assert torch.equal(got, ops.gemm(u.materialize(), tW, col_scale=tcs, col_shift=tch, leaky=True))
    full = np.concatenate([np.concatenate([x, np.zeros((1, C1), np.float32)])[idx[:, 0]]] + ([skip] if C2 else []), 1)
    ref = full.astype(np.float64) @ W.astype(np.float64) * cs + ch
    ref = np.where(ref > 0, ref, 0.2 * ref)
    assert np.abs(got.cpu().numpy() - ref).max() <= 4e-6 * max(1.0, np.abs(ref).max())
    if C2:
        a1 = rng.standard_normal((m, C1)).astype(np.float32)
        got2 = ops.gemm_cat2(_t(a1, device), _t(skip, device), tW).cpu().numpy()
        ref2 = np.concatenate([a1, skip], 1).astype(np.float64) @ W.astype(np.float64)
        assert np.abs(got2 - ref2).max() <= 4e-6 * max(1.0, np.abs(ref2).max())


def test_shapes_it_cannot_address_stay_on_the_fp32_kernel(device):
    """K = 48 (not a multiple of 32) and a concatenation whose first part is 16 wide: d3f_gemm_x3 answers D3F_ERR_ARG through the
    C ABI; the host routes them to d3f_gemm_f32t."""
    from d3feat_amd import _lib, ops
    lib = _lib.load()
    rng = np.random.default_rng(1)
    A, W = rng.standard_normal((300, 48)).astype(np.float32), rng.standard_normal((48, 32)).astype(np.float32)
    got = ops.gemm(_t(A, device), _t(W, device)).cpu().numpy()
    assert np.abs(got - A.astype(np.float64) @ W.astype(np.float64)).max() <= 2e-5
    tA, out = _t(A, device), torch.empty((300, 32), device=device)
    wx = torch.empty(int(lib.d3f_gemm_x3_packed_bytes(48, 32)) // 2, dtype=torch.int16, device=device)
    assert lib.d3f_gemm_pack_x3(_t(W, device).data_ptr(), 32, 48, 32, wx.data_ptr(), None) == 0
    rc = lib.d3f_gemm_x3(tA.data_ptr(), 300, 48, 48, None, 0, None, 0, 0, wx.data_ptr(), out.data_ptr(), 32, 300, 32, None, None, None,
                         None, 0, 0, 0.0, None, 0, None, None, 0, None)
    assert rc == -3
    torch.cuda.synchronize()


@pytest.mark.parametrize("M,K,N", [(300, 4096, 128), (50000, 64, 32)])
def test_cabi_with_the_header_documented_sizes(device, M, K, N):
    """A plain C caller: d3f_gemm_x3_packed_bytes / d3f_gemm_pack_x3 / d3f_gemm_x3_workspace_bytes / d3f_gemm_x3 on the null stream."""
    from d3feat_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32)
    tA, tW = _t(A, device), _t(W, device)
    nb = int(lib.d3f_gemm_x3_packed_bytes(K, N))
    assert nb == -(-N // 32) * (K // 32) * 7680
    wx = torch.empty(nb, dtype=torch.uint8, device=device)
    assert lib.d3f_gemm_pack_x3(tW.data_ptr(), N, K, N, wx.data_ptr(), None) == 0
    wsb = int(lib.d3f_gemm_x3_workspace_bytes(M, N, K, 0))
    ws = torch.empty(wsb, dtype=torch.uint8, device=device)
    out = torch.empty((M, N), device=device)
    rc = lib.d3f_gemm_x3(tA.data_ptr(), M, K, K, None, 0, None, 0, 0, wx.data_ptr(), out.data_ptr(), N, M, N, None, None, None, None, 0,
                         0, 0.0, ws.data_ptr(), C.c_size_t(wsb), None, None, 0, None)
    assert rc == 0
    torch.cuda.synchronize()
    ref = A.astype(np.float64) @ W.astype(np.float64)
    assert np.abs(out.cpu().numpy() - ref).max() <= 4e-6 * max(1.0, np.abs(ref).max())
    if wsb > 256:      # a K-split plan: one byte less is refused, nothing is launched
        rc = lib.d3f_gemm_x3(tA.data_ptr(), M, K, K, None, 0, None, 0, 0, wx.data_ptr(), out.data_ptr(), N, M, N, None, None, None, None,
                             0, 0, 0.0, ws.data_ptr(), C.c_size_t((wsb - 256) // 2), None, None, 0, None)
        assert rc == -2
