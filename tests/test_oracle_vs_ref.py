"""Pins the oracle (oracle/d3f_oracle.c, the plain-C restatement the GPU parity tests compare against):

  1. against the committed golden vectors under tests/golden/ -- outputs of the REFERENCE'S OWN C++ on its own demo
     data, produced by tools/make_golden.py (runs everywhere, including the GPU box);
  2. against the reference's own C++ compiled in place (oracle/_ref) on fresh seeded inputs, when that library is
     present (i.e. where /root/reference existed at build time).
All comparisons are bit-exact (integer / index work, and fp32 values whose every rounding step is specified).
"""
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN, bits, small_cloud


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest()


@pytest.fixture(scope="module")
def gold():
    g = dict(np.load(os.path.join(GOLDEN, "preprocess.npz")))
    g["head"] = np.load(os.path.join(GOLDEN, "demo_bin0_head.npy"))
    g["sub0"] = np.load(os.path.join(GOLDEN, "demo_bin0_sub003.npy"))
    return g


# ---- 1. golden vectors --------------------------------------------------------------------------------------------

def test_golden_manifest_is_consistent():
    import json
    man = json.load(open(os.path.join(GOLDEN, "MANIFEST.json")))
    for f, want in man["files"].items():
        assert hashlib.sha256(open(os.path.join(GOLDEN, f), "rb").read()).hexdigest() == want, f


@pytest.mark.parametrize("key,dl", [("head_sub_003", 0.03), ("head_sub_005", 0.05)])
def test_grid_subsampling_matches_reference_output(coracle, gold, key, dl):
    got = coracle.grid_subsampling(gold["head"], dl)
    assert got.shape == gold[key].shape
    assert np.array_equal(bits(got), bits(gold[key]))          # values AND row order (unordered_map iteration order)


def test_batch_grid_subsampling_matches_reference_output(coracle, gold):
    p, l = coracle.batch_grid_subsampling(gold["head"], gold["head_batch_lens_in"], 0.04)
    assert np.array_equal(l, gold["head_batch_lens_out"])
    assert np.array_equal(bits(p), bits(gold["head_batch_sub_004"]))


def test_wrapper_core_features_and_labels_match_reference_output(coracle, gold):
    p, f, c = coracle.grid_subsampling(gold["head"], 0.04, gold["wrap_features_in"], gold["wrap_labels_in"])
    assert np.array_equal(bits(p), bits(gold["wrap_sub_004"]))
    assert np.array_equal(bits(f), bits(gold["wrap_sub_features"]))   # true division, unlike the barycentres
    assert np.array_equal(c, gold["wrap_sub_labels"])                  # largest label id present in the voxel


def test_neighbors_head_cloud_match_both_reference_paths(coracle, gold):
    hs = gold["head_sub_003"]
    hl = np.asarray([len(hs)], np.int32)
    got = coracle.batch_neighbors(hs, hs, hl, hl, np.float32(0.075))
    stable = gold["head_nbr_ordered"].astype(np.int32)
    assert np.array_equal(got, stable)                                  # == batch_ordered_neighbors (neighbors.cpp:125-208)
    nano = gold["head_nbr_nanoflann"].astype(np.int32)                  # active path: equal up to order inside d2 ties
    assert nano.shape == got.shape
    rows = np.nonzero(np.any(nano != got, axis=1))[0]
    def d2_of(row, r):
        d = (hs[r] - hs[np.minimum(row, len(hs) - 1)]).astype(np.float32)
        d = d * d
        return (d[:, 0] + d[:, 1]) + d[:, 2]
    for r in rows:
        assert sorted(nano[r]) == sorted(got[r])
        # the two orders hold bit-equal d2 column by column: they differ only by a permutation inside a tie
        assert np.array_equal(bits(d2_of(nano[r], r)), bits(d2_of(got[r], r)))
    # OrderedNeighbors (single cloud, pad -1)
    on = gold["head_ordered_neighbors_q500"].astype(np.int32)
    mine = coracle.batch_neighbors(hs[:500], hs, np.asarray([500], np.int32), hl, np.float32(0.075))
    mine = np.where(mine == len(hs), -1, mine)
    assert np.array_equal(mine, on)


def test_neighbors_demo_pair_match_reference(coracle, gold):
    sub0 = gold["sub0"]
    assert len(sub0) == int(gold["demo_sub_counts"][0]) == 14007
    pts = np.concatenate([sub0, sub0])
    pl = np.asarray([len(sub0)] * 2, np.int32)
    got = coracle.batch_neighbors(pts, pts, pl, pl, np.float32(0.03 * 2.5))
    assert got.shape[1] == int(gold["demo_nbr_kmax"][0]) == 74
    assert _sha(got) == gold["demo_nbr_ordered_sha256"].tobytes()       # the whole matrix of the stable path
    assert np.array_equal(got[: len(sub0), :40], gold["demo_nbr_ordered_first40"].astype(np.int32))
    assert np.array_equal(np.sum(got < len(pts), axis=1), gold["demo_nbr_counts"].astype(np.int64))
    # the active nanoflann path differs from it on exactly the recorded tie rows, and only by a swap inside the tie
    tie = gold["demo_nbr_tie_rows"]
    assert len(tie) == 2 and tie[1] == tie[0] + len(sub0)               # one per copy of the cloud
    assert np.array_equal(got[tie], gold["demo_nbr_tie_rows_ordered"])
    nano_rows = gold["demo_nbr_tie_rows_nanoflann"]
    patched = got.copy()
    patched[tie] = nano_rows
    assert _sha(patched) == gold["demo_nbr_nanoflann_sha256"].tobytes()
    for a, b in zip(got[tie], nano_rows):
        assert sorted(a) == sorted(b) and np.sum(a != b) == 2


def test_pyramid_and_calibration_match_reference(coracle, gold):
    """5-level pyramid of the demo self-pair with the oracle's two ops == the reference's (sizes, widths, point bits,
    neighbour-count histograms); limits from both demo pairs' histograms = [37, 35, 36, 38, 38]."""
    from d3feat_amd.utils.config import threedmatch_config
    from oracle import network_np as onp
    cfg = threedmatch_config()
    sub0 = gold["sub0"]
    pts = np.concatenate([sub0, sub0])
    pl = np.asarray([len(sub0)] * 2, np.int32)
    hist_n = onp.hist_size(cfg)
    assert hist_n == 905
    inp = onp.descriptor_input(cfg, pts, np.ones((len(pts), 1), np.float32), pl, np.full(5, hist_n, np.int32),
                               lambda q, s, ql, sl, r: coracle.batch_neighbors(q, s, ql, sl, r),
                               lambda p, l, dl: coracle.batch_grid_subsampling(p, l, dl))
    assert [p.shape[0] for p in inp["points"]] == list(gold["pyr_sizes"]) == [28014, 7612, 2102, 602, 182]
    assert [m.shape[1] for m in inp["neighbors"]] == list(gold["pyr_kmax_conv"])
    assert [m.shape[1] for m in inp["pools"]] == list(gold["pyr_kmax_pool"])
    assert [m.shape[1] for m in inp["upsamples"]] == list(gold["pyr_kmax_up"])
    for l in range(5):
        assert _sha(inp["points"][l]) == gold["pyr_points_sha256_%d" % l].tobytes()
    assert np.array_equal(onp.neighbor_histograms(inp["neighbors"], hist_n), gold["pyr_hist_bin0"])
    assert list(gold["calib_limits_demo_pair"]) == [37, 35, 36, 38, 38]


def test_pyramid_constants_bits():
    """the exact fp32 radii / cell sizes (double arithmetic, one cast)."""
    from d3feat_amd.utils.config import kitti_config, threedmatch_config
    from oracle import network_np as onp
    c = onp.pyramid_constants(threedmatch_config())
    assert [int(np.float32(x["r"]).view(np.uint32)) for x in c] == [0x3d99999a, 0x3e19999a, 0x3e99999a, 0x3f19999a, 0x3f99999a]
    assert [int(np.float32(x["dl_pool"]).view(np.uint32)) for x in c[:4]] == [0x3d75c28f, 0x3df5c28f, 0x3e75c28f, 0x3ef5c28f]
    k = onp.pyramid_constants(kitti_config())
    assert [int(np.float32(x["dl_pool"]).view(np.uint32)) for x in k[:4]] == [0x3f19999a, 0x3f99999a, 0x4019999a, 0x4099999a]


# ---- 2. live reference library ------------------------------------------------------------------------------------

@pytest.mark.parametrize("n,dl", [(1, 0.1), (13, 0.05), (14, 0.05), (30, 0.01), (5000, 0.03), (60000, 0.02)])
def test_grid_subsampling_vs_live_reference(coracle, reflib, n, dl):
    p = small_cloud(100 + n, n)
    assert np.array_equal(bits(coracle.grid_subsampling(p, dl)), bits(reflib.grid_subsampling(p, dl)))


def test_grid_subsampling_negative_coordinates_vs_live_reference(coracle, reflib):
    p = small_cloud(9, 20000) - np.float32(3.7)
    assert np.array_equal(bits(coracle.grid_subsampling(p, 0.045)), bits(reflib.grid_subsampling(p, 0.045)))


def test_batch_grid_subsampling_vs_live_reference(coracle, reflib):
    p = np.concatenate([small_cloud(1, 9000), small_cloud(2, 1), small_cloud(3, 4000, (0.5, 0.5, 0.5))])
    lens = np.asarray([9000, 1, 4000], np.int32)
    a, al = coracle.batch_grid_subsampling(p, lens, 0.06)
    b, bl = reflib.batch_grid_subsampling(p, lens, 0.06)
    assert np.array_equal(al, bl) and np.array_equal(bits(a), bits(b))


def test_wrapper_core_vs_live_reference(coracle, refwrap):
    rng = np.random.default_rng(4)
    p = small_cloud(5, 12000)
    f = rng.standard_normal((12000, 5)).astype(np.float32)
    c = rng.integers(-3, 9, (12000, 1)).astype(np.int32)
    a = coracle.grid_subsampling(p, 0.07, f, c)
    b = refwrap.grid_subsampling(p, 0.07, f, c)
    for x, y in zip(a, b):
        assert np.array_equal(bits(x), bits(y))


@pytest.mark.parametrize("seed,r", [(0, 0.08), (1, 0.15)])
def test_batch_neighbors_vs_live_reference(coracle, reflib, seed, r):
    q = small_cloud(seed, 3000, (1, 1, 0.3))
    s = small_cloud(seed + 50, 5000, (1, 1, 0.3))
    ql, sl = np.asarray([1000, 2000], np.int32), np.asarray([3500, 1500], np.int32)
    got = coracle.batch_neighbors(q, s, ql, sl, np.float32(r))
    assert np.array_equal(got, reflib.batch_ordered_neighbors(q, s, ql, sl, np.float32(r)))
    assert np.array_equal(got, coracle.batch_neighbors(q, s, ql, sl, np.float32(r), grid=False))
    nano = reflib.batch_nanoflann_neighbors(q, s, ql, sl, np.float32(r))
    assert nano.shape == got.shape
    assert np.array_equal(np.sort(nano, axis=1), np.sort(got, axis=1))   # same sets; order differs only inside d2 ties
