"""The fused KPConv kernels' contraction in the operand-split form (csrc/kpconv.hip, round 5; the scheme of csrc/gemm_x3.h inside
kpconv_fused_kernel): fp32 in, fp32 out -- measured against a float64 evaluation of kernels/convolution_ops.py:161-255 next to the
v_mfma_f32_16x16x4_f32 form on the same operands, and through exactness cases (operands whose products and sums are exact in
either form)."""
import numpy as np
import pytest
import torch

from conftest import surface_cloud

pytestmark = pytest.mark.gpu


def _kpconv64(pts, nb, f, kp, W, extent):
    """float64 KPConv_ops (linear influence, sum aggregation) + neighbour-count division, on the device."""
    n = pts.shape[0]
    P = torch.cat([pts.double(), torch.full((1, 3), 1e6, dtype=torch.float64, device=pts.device)])
    F = torch.cat([f.double(), torch.zeros((1, f.shape[1]), dtype=torch.float64, device=f.device)])
    idx = nb.long()
    rel = P[idx] - pts.double()[:, None, :]                                    # [n, K, 3]
    d = (rel[:, :, None, :] - kp.double()[None, None]).pow(2).sum(-1).add(1e-10).sqrt()      # [n, K, 15]
    w = (1.0 - d / (2.0 * extent)).clamp(min=0.0)                              # (the reference's linear influence: :215)
    wf = torch.einsum("nkp,nkc->npc", w, F[idx])                               # [n, 15, Cin]
    out = torch.einsum("npc,pco->no", wf, W.double())
    cnt = (F[idx].sum(-1) > 0).sum(-1).clamp(min=1).double()
    return out / cnt[:, None]


def _operands(cin, device, seed, n_raw=30000):
    from d3feat_amd import ops
    rng = np.random.default_rng(seed)
    s0 = surface_cloud(seed, n_raw=n_raw)
    pts = torch.from_numpy(s0).to(device)
    lens = [len(s0)]
    nb = ops.batch_radius_neighbors(pts, pts, lens, lens, 0.075, 42)[0]
    f = torch.from_numpy(rng.standard_normal((len(s0), cin)).astype(np.float32)).to(device)
    W = torch.from_numpy((rng.standard_normal((15, cin, cin)) * np.sqrt(2.0 / (15 * cin))).astype(np.float32)).to(device)
    kp = (rng.standard_normal((15, 3)) * 0.025).astype(np.float32)
    kp[0] = 0
    return pts, nb, f, W, kp


def _both(ops, *a, **k):
    keep = ops.KP_X3
    fn = ops.kpconv_fused32 if a[3].shape[1] == 32 else ops.kpconv_fused       # (Cin = 32: the level-0 kernel)
    try:
        ops.KP_X3 = True
        x3 = fn(*a, **k)
        ops.KP_X3 = False
        f32 = fn(*a, **k)
    finally:
        ops.KP_X3 = keep
    torch.cuda.synchronize()
    return x3, f32


@pytest.mark.parametrize("cin", [32, 64, 128, 256])
def test_split_contraction_error_against_float64(device, cin):
    from d3feat_amd import ops
    pts, nb, f, W, kp = _operands(cin, device, 640 + cin)
    x3, f32 = _both(ops, pts, pts, nb, f, kp, W, 0.03)
    ref = _kpconv64(pts, nb, f, torch.from_numpy(kp).to(device), W, 0.03)
    scale = ref.abs().max().item()
    e3, e32 = (x3.double() - ref).abs().max().item() / scale, (f32.double() - ref).abs().max().item() / scale
    print("Cin %d: split form %.2e, fp32 MFMA form %.2e of max |out| = %.3g" % (cin, e3, e32, scale))
    # both carry the influences' v_sqrt / fp32 aggregation error (~1e-6); the split contraction must not add to it
    assert e3 <= 5e-6 and e32 <= 5e-6 and e3 <= 1.5 * e32 + 2e-7
    assert (x3 - f32).abs().max().item() <= 2e-6 * scale


@pytest.mark.parametrize("cin", [32, 64, 128])
def test_split_contraction_exactness_cases(device, cin):
    """K_values = a selection: output column o of kernel point p copies weighted-feature channel (o + p) % Cin -- one nonzero
    product per (k, column), weights 1.0: every plane product is exact and the sum over the 15 kernel points is the same fp32
    sum in both forms up to order; with ONE kernel point selected the output IS the weighted feature, bit for bit."""
    from d3feat_amd import ops
    pts, nb, f, _, kp = _operands(cin, device, 77 + cin, n_raw=15000)
    W1 = torch.zeros((15, cin, cin), dtype=torch.float32, device=device)
    W1[3] = torch.eye(cin, device=device)                      # out[:, o] = wf[:, 3, o] / count
    x3, f32 = _both(ops, pts, pts, nb, f, kp, W1, 0.03)
    assert torch.equal(x3, f32)                                 # one exact product per output: both forms copy the same float
    wf, inv = ops.kpconv_aggregate(pts, pts, nb, f, kp, 0.03)
    want = wf.view(-1, 15, cin)[:, 3, :] * inv[:, None]
    assert (x3 - want).abs().max().item() <= 1e-6 * max(1.0, want.abs().max().item())   # (aggregation order differs: vec4 kernel)
    # integer-valued operands: every product and partial sum is an integer below 2^24 -> exact in any order, in both forms
    fi = torch.randint(-3, 4, f.shape, device=device).float()
    Wi = torch.randint(-2, 3, (15, cin, cin), device=device).float()
    kp0 = np.zeros((15, 3), np.float32)
    kp0[1:] = 10.0                                              # only the centre kernel point has influence ...
    x3, f32 = _both(ops, pts, pts, nb, fi, kp0, Wi, 1e3)        # ... and a huge extent makes it ~1 - d/1000: not an integer:
    assert (x3 - f32).abs().max().item() <= 2e-6 * max(1.0, f32.abs().max().item())


def test_split_contraction_epilogue_and_capacity_rows(device):
    """Epilogue operands (batch-norm scale / shift, residual, LeakyReLU) and a capacity-mode call (device-resident row count
    smaller than the grid): the two forms agree to fp32 rounding; rows beyond the real count are not written."""
    from d3feat_amd import ops
    pts, nb, f, W, kp = _operands(64, device, 5, n_raw=20000)
    rng = np.random.default_rng(9)
    cs = torch.from_numpy((rng.random(64) + 0.5).astype(np.float32)).to(device)
    ch = torch.from_numpy(rng.standard_normal(64).astype(np.float32)).to(device)
    res = torch.from_numpy(rng.standard_normal((pts.shape[0], 64)).astype(np.float32)).to(device)
    x3, f32 = _both(ops, pts, pts, nb, f, kp, W, 0.03, col_scale=cs, col_shift=ch, residual=res, leaky=True)
    assert (x3 - f32).abs().max().item() <= 2e-6 * max(1.0, f32.abs().max().item())


@pytest.mark.parametrize("cin,num_kp,influence,mode", [
    (32, 4, "linear", "sum"), (32, 6, "gaussian", "sum"), (32, 13, "linear", "closest"), (32, 15, "constant", "sum"),
    (32, 15, "gaussian", "closest"), (64, 6, "linear", "sum"), (64, 13, "gaussian", "closest"), (128, 4, "constant", "closest")])
def test_split_contraction_generic_modes_against_the_oracle(device, coracle, cin, num_kp, influence, mode):
    """with KP_X3 on, KPConv_ops sends EVERY Cin = Cout = 32 / 64 / 128 call through the operand-split fused kernels --
    also the generic instances (num_kp != 15: partial passes of the split tile, other influences / aggregations) that the
    production model never reaches.  Each is checked here against the float64-free CPU oracle (oracle/network_np.KPConv_ops =
    kernels/convolution_ops.py:161-255) and against the fp32 MFMA form of the same kernel."""
    from d3feat_amd import ops
    from d3feat_amd.kernels import convolution_ops as conv_ops
    from oracle import network_np as onp
    s0 = surface_cloud(40 + num_kp, n_raw=12000)
    rng = np.random.default_rng(100 * cin + num_kp)
    lens = np.asarray([len(s0)], np.int32)
    nb = coracle.batch_neighbors(s0, s0, lens, lens, np.float32(0.075))[:, :34].astype(np.int32)
    f = rng.standard_normal((len(s0), cin)).astype(np.float32)
    W = (rng.standard_normal((num_kp, cin, cin)) * np.sqrt(2.0 / (num_kp * cin))).astype(np.float32)
    KP = (rng.standard_normal((num_kp, 3)) * 0.03).astype(np.float32)
    KP[0] = 0
    want = onp.KPConv_ops(s0, s0, nb, f, KP, W, 0.03, influence, mode).numpy()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    keep = ops.KP_X3
    try:
        ops.KP_X3 = True
        x3 = conv_ops.KPConv_ops(t(s0), t(s0), t(nb), t(f), KP, t(W), 0.03, influence, mode).cpu().numpy()
        ops.KP_X3 = False
        f32 = conv_ops.KPConv_ops(t(s0), t(s0), t(nb), t(f), KP, t(W), 0.03, influence, mode).cpu().numpy()
    finally:
        ops.KP_X3 = keep
    scale = max(1.0, float(np.abs(want).max()))
    assert np.abs(x3 - want).max() <= 1e-4 * scale, np.abs(x3 - want).max()
    assert np.abs(f32 - want).max() <= 1e-4 * scale
    assert np.abs(x3 - f32).max() <= 5e-6 * scale


def test_aggregation_on_the_matrix_cores_equals_the_vector_form(device):
    """d3f_kpconv_fused32_mfma (round 6): the per-query [15 x K] x [K x 32] aggregation as v_mfma_f32_16x16x1_4b_f32 rank-1 updates --
    the same fp32 multiply-adds in the same neighbour order as the vector form's FMA chains, then the same split contraction over a
    channel-major k order: results equal to fp32 rounding of the contraction's summation order, and within 5e-6 of float64.
    Epilogue operands, shadow neighbours (a ragged last tile, all-shadow rows) and a capacity-mode call included."""
    from d3feat_amd import ops
    pts, nb, f, W, kp = _operands(32, device, 911, n_raw=30000)
    n = pts.shape[0]
    nb = nb.clone()
    nb[5, :] = n                                   # an all-shadow row
    nb[7, 3:] = n                                  # three neighbours only
    rng = np.random.default_rng(2)
    cs = torch.from_numpy((rng.random(32) + 0.5).astype(np.float32)).to(device)
    ch = torch.from_numpy(rng.standard_normal(32).astype(np.float32)).to(device)
    res = torch.from_numpy(rng.standard_normal((n, 32)).astype(np.float32)).to(device)
    keep = ops.KP_MFMA
    try:
        for kw in (dict(), dict(col_scale=cs, col_shift=ch, residual=res, leaky=True)):
            ops.KP_MFMA = True
            a = ops.kpconv_fused32(pts, pts, nb, f, kp, W, 0.03, **kw)
            ops.KP_MFMA = False
            b = ops.kpconv_fused32(pts, pts, nb, f, kp, W, 0.03, **kw)
            torch.cuda.synchronize()
            scale = max(1.0, b.abs().max().item())
            assert (a - b).abs().max().item() <= 2e-6 * scale, (a - b).abs().max().item()
            assert torch.equal(a[5], b[5])
        ops.KP_MFMA = True
        a = ops.kpconv_fused32(pts, pts, nb, f, kp, W, 0.03)
        ref = _kpconv64(pts, nb, f, torch.from_numpy(kp).to(device), W, 0.03)
        assert (a.double() - ref).abs().max().item() <= 5e-6 * ref.abs().max().item()
        # capacity mode: the device-resident row count is smaller than the grid; rows beyond it are not written
        m = n - 1000
        q2 = pts.clone()
        q2.n_dev = torch.tensor([m], dtype=torch.int32, device=device)
        q2.n_hint = m
        a2 = ops.kpconv_fused32(q2, pts, nb, f, kp, W, 0.03)
        torch.cuda.synchronize()
        assert torch.equal(a2[:m], a[:m])
    finally:
        ops.KP_MFMA = keep
