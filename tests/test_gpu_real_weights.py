"""The reference's REAL trained tensors through the oracle and the HIP path (SURVEY.md "hard part 7": tolerances on
realistic activation scale, not only on He-initialised random weights).

tests/golden/kitti_epoch61_weights.npz (tools/make_golden_weights.py) holds the 34 weight tensors the reference ships for
its KITTI model (results_kitti/Log_11011605/kernel_points/epoch61/*.npy, written by utils/trainer.py:503-557) and
kitti_kernel_points.npz the 10 trained kernel-point dispositions.  What the dump lacks (batch-norm statistics, the last
block's conv2/conv3/shortcut and uplayer_3) is filled with seeded values.  KITTI configuration (dl = 0.30 m), two DIFFERENT
LiDAR-like frames per stack (datasets/KITTI.py:94-106).  Bar: descriptors and scores within 1e-4 ABSOLUTE of the oracle.
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
TOL = 1e-4


def real_kitti_weights(cfg, seed=11):
    from d3feat_amd.models.variables import build_variables
    W = build_variables(cfg, seed=seed, randomize_bn=True).values
    real = np.load(os.path.join(GOLDEN, "kitti_epoch61_weights.npz"))
    kps = np.load(os.path.join(GOLDEN, "kitti_kernel_points.npz"))
    n = 0
    for src in (real, kps):
        for k in src.files:
            name = k.replace("__", "/")
            assert name in W and W[name].shape == src[k].shape, (name, src[k].shape)
            W[name] = np.ascontiguousarray(src[k], np.float32)
            n += 1
    assert n == 44
    return W


def test_real_kitti_weights_forward_vs_oracle(device, coracle):
    from d3feat_amd.engine import FragmentEngine
    from d3feat_amd.utils.config import kitti_config
    from d3feat_amd.utils.synthetic import lidar_sweep
    from oracle import parity as par
    torch.set_num_threads(min(os.cpu_count() or 1, 64))     # more intra-op threads than that made the CPU graph slower
    cfg = kitti_config()
    W = real_kitti_weights(cfg)
    limits = np.asarray([25, 25, 25, 25, 25], np.int32)
    raws = [lidar_sweep(s, 120000) for s in (5, 105)]
    subs = [coracle.grid_subsampling(r, np.float32(0.3)) for r in raws]
    ref = par.fragment_reference(cfg, W, None, limits, co=coracle, clouds=subs)
    eng = FragmentEngine(cfg, W, limits, raw_cap=250000, n0_cap=40000, level_ratio=0.6, slots=1, device=device, two_clouds=True)
    p, d, s = (t.cpu().numpy() for t in eng.run(tuple(torch.from_numpy(r).to(device) for r in raws)))
    assert eng.fallbacks == 0
    c = par.compare_fragment(ref, p, d, s)
    assert c["points_equal"], c
    assert c["desc_max_abs"] <= TOL and c["score_max_abs"] <= TOL, c
    # the eager path on the same stack
    pe, de, se = (t.cpu().numpy() for t in eng.run_eager(tuple(torch.from_numpy(r).to(device) for r in raws)))
    c = par.compare_fragment(ref, pe, de, se)
    assert c["points_equal"] and c["desc_max_abs"] <= TOL and c["score_max_abs"] <= TOL, c
    # trained weights produce non-trivial descriptors: unit norm, and scores that actually vary
    assert np.allclose(np.linalg.norm(ref["desc"], axis=1), 1.0, atol=1e-5)
    assert ref["score"].std() > 1e-3


def test_real_weight_layers_absolute_tolerance(device, coracle):
    """Every KPConv of the trained model on real geometry at ITS level's scale, absolute 1e-4 on activations produced from
    unit-scale inputs -- the per-layer error budget behind the end-to-end bound."""
    from d3feat_amd.kernels import convolution_ops as conv_ops
    from d3feat_amd.utils.config import kitti_config
    from d3feat_amd.utils.synthetic import lidar_sweep
    from oracle import network_np as onp
    cfg = kitti_config()
    W = real_kitti_weights(cfg)
    s0 = coracle.grid_subsampling(lidar_sweep(9, 120000), np.float32(0.3))
    lens = np.asarray([len(s0)], np.int32)
    nb = coracle.batch_neighbors(s0, s0, lens, lens, np.float32(0.75))[:, :25]
    rng = np.random.default_rng(0)
    for scope in ("layer_0/resnetb_1/conv2", "layer_1/resnetb_0/conv2", "layer_2/resnetb_0/conv2", "layer_3/resnetb_0/conv2"):
        Wk, KP = W[scope + "/weights"], W[scope + "/kernel_points"]
        cin = Wk.shape[1]
        # the trained kernel points of deeper levels live at that level's scale: rescale the geometry instead of the kernel
        scale = np.float32(2 ** int(scope.split("/")[0].split("_")[1]))
        f = np.maximum(rng.standard_normal((len(s0), cin)), -0.2).astype(np.float32)       # leaky-relu-like inputs
        want = onp.KPConv_ops(s0 * scale, s0 * scale, nb, f, KP, Wk, 0.3 * float(scale), "linear", "sum").numpy()
        got = conv_ops.KPConv_ops(torch.from_numpy(s0 * scale).to(device), torch.from_numpy(s0 * scale).to(device),
                                  torch.from_numpy(nb.astype(np.int32)).to(device), torch.from_numpy(f).to(device), KP,
                                  torch.from_numpy(Wk).to(device), 0.3 * float(scale), "linear", "sum").cpu().numpy()
        err = np.abs(got - want).max()
        # fp32 summation noise of this layer is ~3e-6 at |activation| ~ 8 (oracle vs an fp64 evaluation): absolute bound
        assert err <= TOL, (scope, err, np.abs(want).max())


def test_row_positive_sign_is_order_independent(device):
    """The neighbour count of KPConv_ops (kernels/convolution_ops.py:250-252) tests `sum_c f > 0` per support row.  For rows
    whose fp32 sum is rounding noise the answer depends on the summation order, and the reference's order (Eigen's
    reduction inside tf.reduce_sum) is an implementation detail.  The kernel decides by the sign of the EXACT sum (fp64
    accumulation), which (a) equals every fp32 order on rows that are not rounding noise and (b) is deterministic on the
    adversarial ones below, where two fp32 orders of the very same row disagree with each other."""
    from d3feat_amd import _lib, ops
    rng = np.random.default_rng(1)
    C = 64
    rows = []
    # (a) ordinary rows + exact-cancellation rows whose sum is exactly representable in any order
    rows += [rng.standard_normal(C).astype(np.float32) for _ in range(200)]
    z = np.zeros(C, np.float32); z[3], z[40] = 2.5, -2.5
    rows.append(z)                                                   # exactly 0 -> not positive
    y = z.copy(); y[10] = np.float32(2.0 ** -20)
    rows.append(y)                                                   # 2^-20: exact in every order (fits the 24-bit window)
    rows.append(-y)
    # (b) rounding-noise rows: big +a, -a and a tiny term that an unlucky order absorbs
    for tiny in (1e-10, -1e-10, 3e-9, -3e-9):
        a = np.zeros(C, np.float32); a[0], a[1], a[63] = 1.0, tiny, -1.0
        rows.append(a)
        b = np.zeros(C, np.float32); b[0], b[62], b[63] = 4096.0, -4096.0, tiny * 1e3
        rows.append(b)
    f = np.stack(rows)
    exact = np.asarray([float(np.sum(r.astype(np.float64))) > 0.0 for r in f])
    fwd = np.asarray([np.float32(0) + sum((np.float32(v) for v in r), np.float32(0)) > 0 for r in f])
    rev = np.asarray([sum((np.float32(v) for v in r[::-1]), np.float32(0)) > 0 for r in f])
    n_plain = 203                                                    # ordinary + exactly representable rows
    assert np.array_equal(exact[:n_plain], fwd[:n_plain]) and np.array_equal(exact[:n_plain], rev[:n_plain])
    assert (fwd[n_plain:] != rev[n_plain:]).any(), "the adversarial rows must make two fp32 orders disagree"
    assert (fwd[n_plain:] != exact[n_plain:]).any()
    lib = _lib.load()
    ft = torch.from_numpy(f).to(device)
    pos = torch.empty((len(f),), dtype=torch.uint8, device=device)
    _lib.check(lib.d3f_row_positive(ft.data_ptr(), len(f), C, C, pos.data_ptr(), None, 0, ops._stream(device)), "row_positive")
    assert np.array_equal(pos.cpu().numpy().astype(bool), exact)


def test_lazy_variables_equal_build_variables(device, coracle):
    """KernelPointFCNN(weights=None, seed=s) creates its variables while running; they must be the tensors
    build_variables(seed=s) creates without running (same creation order as models/network_blocks.py:321-368)."""
    from d3feat_amd.datasets.common import FragmentDataset
    from d3feat_amd.models.KPFCNN_model import KernelPointFCNN
    from d3feat_amd.models.variables import build_variables
    from d3feat_amd.utils.config import threedmatch_config
    from conftest import surface_cloud
    cfg = threedmatch_config()
    sub = surface_cloud(3, n_raw=20000)
    ds = FragmentDataset([sub], fast=True)
    ds.device = device
    ds.neighborhood_limits = np.asarray([37, 35, 36, 38, 38], np.int32)
    gen, _, _ = ds.get_batch_gen("test", cfg)
    flat = ds.get_tf_mapping(cfg)(*ds._to_device(next(iter(gen()))))
    model = KernelPointFCNN(flat, cfg, weights=None, seed=5, device=device)
    want = build_variables(cfg, seed=5).values
    got = model.weights()
    assert set(got) == set(want)
    for k in want:
        assert np.array_equal(got[k], want[k]), k
