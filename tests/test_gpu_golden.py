"""HIP kernels vs the committed golden vectors (tests/golden/: outputs of the reference's own C++ on its demo data,
tools/make_golden.py) -- no oracle in between.  Bit-exact."""
import hashlib
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, bits

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.fixture(scope="module")
def gold():
    g = dict(np.load(os.path.join(GOLDEN, "preprocess.npz")))
    g["head"] = np.load(os.path.join(GOLDEN, "demo_bin0_head.npy"))
    g["sub0"] = np.load(os.path.join(GOLDEN, "demo_bin0_sub003.npy"))
    return g


@pytest.mark.parametrize("key,dl", [("head_sub_003", 0.03), ("head_sub_005", 0.05)])
def test_grid_subsampling_golden(device, gold, key, dl):
    from d3feat_amd import tf_custom_ops as tfo
    got = tfo.grid_subsampling(_t(gold["head"], device), dl).cpu().numpy()
    assert got.shape == gold[key].shape and np.array_equal(bits(got), bits(gold[key]))


def test_batch_grid_subsampling_golden(device, gold):
    from d3feat_amd import tf_custom_ops as tfo
    p, l = tfo.batch_grid_subsampling(_t(gold["head"], device), _t(gold["head_batch_lens_in"], device), 0.04)
    assert np.array_equal(l.cpu().numpy(), gold["head_batch_lens_out"])
    assert np.array_equal(bits(p.cpu().numpy()), bits(gold["head_batch_sub_004"]))


def test_cpp_wrapper_compute_golden(device, gold):
    """grid_subsampling.compute(points, features=, classes=, sampleDl=) of cpp_wrappers (wrapper.cpp:58-286)."""
    from d3feat_amd.cpp_wrappers.cpp_subsampling import grid_subsampling
    p, f, c = grid_subsampling.compute(gold["head"], features=gold["wrap_features_in"], classes=gold["wrap_labels_in"],
                                       sampleDl=0.04, verbose=0)
    assert np.array_equal(bits(p), bits(gold["wrap_sub_004"]))
    assert np.array_equal(bits(f), bits(gold["wrap_sub_features"]))
    assert c.ndim == 2 and np.array_equal(c, gold["wrap_sub_labels"])
    only = grid_subsampling.compute(gold["head"], sampleDl=0.04)
    assert isinstance(only, np.ndarray) and np.array_equal(bits(only), bits(gold["wrap_sub_004"]))
    with pytest.raises(RuntimeError):
        grid_subsampling.compute(gold["head"], features=gold["wrap_features_in"][:10], sampleDl=0.04)
    with pytest.raises(RuntimeError):
        grid_subsampling.compute(gold["head"], sampleDl=0.04, method="nope")


def test_neighbors_head_golden(device, gold):
    from d3feat_amd import tf_custom_ops as tfo
    hs = gold["head_sub_003"]
    hl = np.asarray([len(hs)], np.int32)
    got = tfo.batch_ordered_neighbors(_t(hs, device), _t(hs, device), _t(hl, device), _t(hl, device), np.float32(0.075))
    assert np.array_equal(got.cpu().numpy(), gold["head_nbr_ordered"].astype(np.int32))
    on = tfo.ordered_neighbors(_t(hs[:500], device), _t(hs, device), np.float32(0.075)).cpu().numpy()
    assert np.array_equal(on, gold["head_ordered_neighbors_q500"].astype(np.int32))


def test_neighbors_demo_pair_golden(device, gold):
    """The level-0 conv search of the demo self-pair (28 014 stacked points, Kmax 74): whole matrix by sha256 against the
    reference's stable path; equal to the active nanoflann path after patching its two recorded tie rows."""
    from d3feat_amd import tf_custom_ops as tfo
    sub0 = gold["sub0"]
    pts = _t(np.concatenate([sub0, sub0]), device)
    pl = _t(np.asarray([len(sub0)] * 2, np.int32), device)
    got = tfo.batch_ordered_neighbors(pts, pts, pl, pl, np.float32(0.03 * 2.5)).cpu().numpy()
    assert got.shape == (28014, 74)
    assert hashlib.sha256(got.tobytes()).digest() == gold["demo_nbr_ordered_sha256"].tobytes()
    patched = got.copy()
    patched[gold["demo_nbr_tie_rows"]] = gold["demo_nbr_tie_rows_nanoflann"]
    assert hashlib.sha256(patched.tobytes()).digest() == gold["demo_nbr_nanoflann_sha256"].tobytes()


def test_pyramid_and_calibration_golden(device, gold):
    """init_test_input_pipeline-style calibration on the demo cloud: pyramid sizes, Kmax per level, point bits per level
    and the neighbour-count histograms equal the reference's."""
    from d3feat_amd.datasets.common import FragmentDataset
    from d3feat_amd.utils.config import threedmatch_config
    cfg = threedmatch_config()
    ds = FragmentDataset([gold["sub0"]])
    ds.neighborhood_limits = np.full(5, 905, np.int32)
    hists = ds.calibrate_neighbors(cfg, samples_threshold=10 ** 9)
    assert np.array_equal(hists, gold["pyr_hist_bin0"])
    gen, _, _ = ds.get_batch_gen("test", cfg)
    ds.neighborhood_limits = np.full(5, 905, np.int32)
    flat = ds.get_tf_mapping(cfg)(*ds._to_device(next(iter(gen()))))
    assert [flat[l].shape[0] for l in range(5)] == list(gold["pyr_sizes"])
    assert [flat[5 + l].shape[1] for l in range(5)] == list(gold["pyr_kmax_conv"])
    assert [flat[10 + l].shape[1] for l in range(4)] == list(gold["pyr_kmax_pool"][:4])
    assert [flat[15 + l].shape[1] for l in range(4)] == list(gold["pyr_kmax_up"][:4])
    for l in range(5):
        assert hashlib.sha256(flat[l].cpu().numpy().tobytes()).digest() == gold["pyr_points_sha256_%d" % l].tobytes()


def test_trained_kernel_points_kpconv(device, gold, coracle):
    """KPConv with a TRAINED kernel-point disposition (results_kitti/.../layer_0_resnetb_1_conv2.ply) on demo geometry."""
    from d3feat_amd.kernels import convolution_ops as conv_ops
    from oracle import network_np as onp
    kp = np.load(os.path.join(GOLDEN, "kitti_kernel_points.npz"))["layer_0__resnetb_1__conv2__kernel_points"]
    kp = (kp * np.float32(0.1)).astype(np.float32)            # KITTI dl 0.30 -> 3DMatch dl 0.03
    s = gold["sub0"][:6000]
    L = np.asarray([len(s)], np.int32)
    nb = coracle.batch_neighbors(s, s, L, L, np.float32(0.075))[:, :37].astype(np.int32)
    rng = np.random.default_rng(0)
    f = rng.standard_normal((len(s), 32)).astype(np.float32)
    W = (rng.standard_normal((15, 32, 32)) * 0.25).astype(np.float32)
    want = onp.KPConv_ops(s, s, nb, f, kp, W, 0.03, "linear", "sum").numpy()
    got = conv_ops.KPConv_ops(_t(s, device), _t(s, device), _t(nb, device), _t(f, device), kp, _t(W, device), 0.03,
                              "linear", "sum").cpu().numpy()
    assert np.abs(got - want).max() <= 1e-4 * max(1.0, np.abs(want).max())
