"""Cross-check of the floating-point oracle (oracle/network_np.py, the restatement of the reference's TF graph).

The reference ships no test or golden vector for this part and TensorFlow cannot be installed here:
PARITY UNPINNED by the reference.  What can be done is done here: every op of the restatement is re-derived
independently -- scalar float64 loops written from the formulas in the reference's docstrings/comments
(kernels/convolution_ops.py:161-255, models/D3Feat.py:65-115), not from the vectorised restatement -- on small cases,
and the full forward is checked for the structural properties the reference's graph guarantees."""
import numpy as np
import pytest
import torch

from conftest import GOLDEN
import os


def _kpconv_loops(q, s, idx, f, KP, W, extent, influence="linear", mode="sum"):
    n, K = idx.shape
    P, Cin, Cout = W.shape
    out = np.zeros((n, Cout))
    for i in range(n):
        wf = np.zeros((P, Cin))
        cnt = 0
        for k in range(K):
            j = idx[i, k]
            if j >= len(s):                       # shadow neighbour: point at 1e6, zero features
                continue
            rel = s[j].astype(np.float64) - q[i]
            d2 = ((rel[None, :] - KP.astype(np.float64)) ** 2).sum(1)
            if influence == "linear":
                h = np.maximum(1.0 - np.sqrt(d2 + 1e-10) / (2.0 * extent), 0.0)
            elif influence == "constant":
                h = np.ones(P)
            else:
                sig = extent * 0.3
                h = np.exp(-d2 / (2 * sig ** 2 + 1e-9))
            if mode == "closest":
                m = np.zeros(P)
                m[np.argmin(d2)] = 1.0
                h = h * m
            wf += h[:, None] * f[j][None, :]
            cnt += 1 if f[j].astype(np.float64).sum() > 0 else 0
        out[i] = np.einsum("pc,pco->o", wf, W.astype(np.float64)) / max(cnt, 1)
    return out


@pytest.mark.parametrize("influence,mode", [("linear", "sum"), ("constant", "sum"), ("gaussian", "sum"), ("linear", "closest")])
def test_kpconv_ops_restatement_vs_scalar_loops(coracle, influence, mode):
    from oracle import network_np as onp
    rng = np.random.default_rng(0)
    s = np.load(os.path.join(GOLDEN, "demo_bin0_sub003.npy"))[:1500]
    lens = np.asarray([len(s)], np.int32)
    nb = coracle.batch_neighbors(s[:60], s, np.asarray([60], np.int32), lens, np.float32(0.075))[:, :30]
    f = rng.standard_normal((len(s), 6)).astype(np.float32)
    W = rng.standard_normal((15, 6, 5)).astype(np.float32)
    KP = np.load(os.path.join(GOLDEN, "kitti_kernel_points.npz"))["layer_0__simple_0__kernel_points"] * np.float32(0.1)
    want = _kpconv_loops(s[:60], s, nb, f, KP, W, 0.03, influence, mode)
    got = onp.KPConv_ops(s[:60], s, nb, f, KP, W, 0.03, influence, mode).numpy()
    assert np.abs(got - want).max() <= 2e-5 * max(1.0, np.abs(want).max())


def test_detection_head_restatement_vs_scalar_loops(coracle):
    from oracle import network_np as onp
    rng = np.random.default_rng(1)
    s = np.load(os.path.join(GOLDEN, "demo_bin0_sub003.npy"))[:400]
    L = np.asarray([250, 150], np.int32)
    nb = coracle.batch_neighbors(s, s, L, L, np.float32(0.09))[:, :12]
    x = rng.standard_normal((400, 8)).astype(np.float32)
    in_b = onp.stack_batch_inds(L)
    got = onp.detection_head(torch.from_numpy(x), nb, in_b, L).numpy()[:, 0]
    xd = x.astype(np.float64)
    m = [max(xd[:250].max(), 0.0 if 400 in in_b[0] else -np.inf), max(xd[250:].max(), 0.0 if 400 in in_b[1] else -np.inf)]
    y = np.concatenate([xd[:250] / (m[0] + 1e-6), xd[250:] / (m[1] + 1e-6)])
    want = np.zeros(400)
    for i in range(400):
        rows = [y[j] for j in nb[i] if j < 400]
        cnt = max(sum(1 for r in rows if np.float32(r.astype(np.float32).sum()) != 0), 1)
        mean = (np.sum(rows, 0) if rows else np.zeros(8)) / cnt
        alpha = np.log1p(np.exp(y[i] - mean))
        beta = y[i] / (1e-6 + y[i].max())
        want[i] = (alpha * beta).max()
    assert np.abs(got - want).max() <= 1e-5 * max(1.0, np.abs(want).max())


def test_pools_restatement():
    from oracle import network_np as onp
    x = torch.tensor([[1., -5.], [3., 2.], [-2., 7.]])
    inds = np.asarray([[0, 1, 3], [3, 3, 3], [2, 3, 3]])
    # shadow row (index 3) = per-column minimum (-2, -5): max-pool ignores it unless the row is all-shadow
    assert torch.equal(onp.ind_max_pool(x, inds), torch.tensor([[3., 2.], [-2., -5.], [-2., 7.]]))
    assert torch.equal(onp.closest_pool(x, inds), torch.tensor([[1., -5.], [0., 0.], [-2., 7.]]))


def test_forward_structure_on_reference_geometry(coracle):
    """Full forward of the restatement on a crop of the demo cloud: shapes, unit-norm descriptors, finite positive
    scores, and the self-pair mirror (both halves of a stacked self-pair give identical rows)."""
    from d3feat_amd.models.variables import build_variables
    from d3feat_amd.utils.config import threedmatch_config
    from oracle import network_np as onp
    cfg = threedmatch_config()
    sub = np.load(os.path.join(GOLDEN, "demo_bin0_sub003.npy"))
    c = sub[np.linalg.norm(sub - np.median(sub, axis=0), axis=1) < 0.9]
    assert 500 < len(c) < 5000
    pts = np.concatenate([c, c])
    lens = np.asarray([len(c)] * 2, np.int32)
    inp = onp.descriptor_input(cfg, pts, np.ones((len(pts), 1), np.float32), lens, [37, 35, 36, 38, 38],
                               lambda q, s, ql, sl, r: coracle.batch_neighbors(q, s, ql, sl, r),
                               lambda p, l, dl: coracle.batch_grid_subsampling(p, l, dl))
    W = build_variables(cfg, seed=42, randomize_bn=True).values
    trace = {}
    d, s = onp.forward(cfg, W, inp, trace)
    assert d.shape == (2 * len(c), 32) and s.shape == (2 * len(c), 1)
    assert np.isfinite(d).all() and np.isfinite(s).all()
    assert np.abs(np.linalg.norm(d, axis=1) - 1).max() < 1e-5
    assert (s >= 0).all()
    assert np.abs(d[: len(c)] - d[len(c):]).max() < 1e-6 and np.abs(s[: len(c)] - s[len(c):]).max() < 1e-6
    assert trace["layer_0/simple_0"].shape[1] == 64 and trace["layer_4/resnetb_0"].shape[1] == 2048
    assert trace["uplayer_0/last_unary_1"].shape[1] == 32
