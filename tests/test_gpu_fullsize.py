"""Parity AT the benchmarked configuration (BASELINE.json configs[1]) and on the reference's own
demo geometry (configs[0]) -- the sizes bench.py times, not the reduced ones of the other test files:

  * config #2 full size: seeds 0..3, 300 000 raw points per fragment (edge 1.68 m -> ~29 k points per cloud at 0.03 m),
    neighborhood_limits calibrated on exactly these fragments (datasets/common.py:572-673), through
    FragmentEngine(batch=4) -- four fragments per HIP-graph replay, two replays in flight, the execution bench.py times --
    against the oracle run fragment by fragment the way the reference's tester does (utils/tester.py:196-213):
    every level of the pyramid bit-exact, descriptors and scores within 1e-4 ABSOLUTE;
  * demo self-pair: tests/golden/demo_bin0_sub003.npy (the reference's own grid_subsampling of demo_data/cloud_bin_0.ply,
    14 007 points) stacked with itself, limits of the golden calibration, exact-shape path and engine path vs oracle.
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _oracle_limits(cfg, subs, coracle):
    """calibrate_neighbors on the CPU: untruncated pyramids of every self-pair, histograms summed (common.py:629-670)."""
    from oracle import network_np as onp
    from oracle import parity as par
    hist_n = onp.hist_size(cfg)
    full = np.full(cfg.num_layers, hist_n, np.int32)
    hists = 0
    for s in subs:
        ref = par.fragment_reference(cfg, None, None, full, co=coracle, forward=False, clouds=[s, s])
        hists = hists + onp.neighbor_histograms(ref["inp"]["neighbors"], hist_n)
    return onp.limits_from_histograms(hists), hists


@pytest.mark.timeout(900)
def test_config2_full_size_engine_vs_oracle(device, coracle):
    from d3feat_amd import tf_custom_ops as tfo
    from d3feat_amd.datasets.common import FragmentDataset
    from d3feat_amd.engine import FragmentEngine
    from d3feat_amd.models.variables import build_variables
    from d3feat_amd.utils.config import threedmatch_config
    from d3feat_amd.utils.synthetic import room_fragment
    from oracle import network_np as onp
    from oracle import parity as par
    torch.set_num_threads(min(os.cpu_count() or 1, 64))     # more intra-op threads than that made the CPU graph slower
    cfg = threedmatch_config()
    W = build_variables(cfg, seed=42).values           # bench.py's weights
    raws_host = [room_fragment(s, n_raw=300000, edge=1.68) for s in range(4)]
    raws = [torch.from_numpy(r).to(device) for r in raws_host]
    # ---- calibration: GPU histograms == oracle histograms, hence identical limits
    subs = [tfo.grid_subsampling(r, cfg.first_subsampling_dl).cpu().numpy() for r in raws]
    for s, r in zip(subs, raws_host):
        assert np.array_equal(s.view(np.uint32), coracle.grid_subsampling(r, 0.03).view(np.uint32))
    assert all(27000 < len(s) < 32000 for s in subs), [len(s) for s in subs]
    cal = FragmentDataset(subs)
    hist_n = onp.hist_size(cfg)
    cal.neighborhood_limits = np.full(cfg.num_layers, hist_n, np.int32)
    hists = cal.calibrate_neighbors(cfg, samples_threshold=10 ** 9)
    want_limits, want_hists = _oracle_limits(cfg, subs, coracle)
    assert np.array_equal(hists, want_hists)
    limits = cal.neighborhood_limits
    assert np.array_equal(limits, want_limits)
    # ---- the engine exactly as bench.py builds it (F = 4; two slots here: 6 fragments = one full + one partial replay)
    n0_max = max(len(s) for s in subs)
    eng = FragmentEngine(cfg, W, limits, raw_cap=int(300000 * 1.05) + 1024, n0_cap=(int(n0_max * 1.3) + 1023) // 1024 * 1024,
                         slots=2, device=device, n0_hint=int(np.mean([len(s) for s in subs])), batch=4)
    order = [0, 1, 2, 3, 2, 0]
    eng.submit(0, [raws[i] for i in order[:4]])
    eng.submit(1, [raws[i] for i in order[4:]])
    refs = [par.fragment_reference(cfg, W, raws_host[i], limits, co=coracle) for i in range(4)]
    L = cfg.num_layers
    for slot, members in ((0, order[:4]), (1, order[4:])):
        outs = eng.fetch(slot)
        assert eng.fallbacks == 0
        sl = eng.slots[slot]
        flat = eng.reference_order_flat(slot)      # (the slot's pyramid lives in the internal cell order: renumbered back)
        totals = [int(sl.flat[l].n_dev.item()) for l in range(L)]
        lens = [x.tolist() for x in sl.level_lengths]            # per level: [n_1, n_1, n_2, n_2, ...] (+ stand-ins)
        for j, fi in enumerate(members):
            ref = refs[fi]
            offsets = [sum(lens[l][: 2 * j]) for l in range(L)]
            for l in range(L):
                assert lens[l][2 * j] == lens[l][2 * j + 1] == ref["inp"]["points"][l].shape[0] // 2
            par.check_pyramid_slice(flat, ref, L, offsets, totals, fast=True)
            p, d, s = (t.cpu().numpy() for t in outs[j])
            c = par.compare_fragment(ref, p, d, s, nb0=flat[L].cpu().numpy(), row0=offsets[0], total=totals[0])
            assert c["points_equal"] and c["idx_equal"], c
            assert c["desc_max_abs"] <= TOL and c["score_max_abs"] <= TOL, c
    # ---- the packed record view is the same data
    eng.submit(0, [raws[1]])
    rec = eng.fetch(0, packed=True)[0].cpu().numpy()
    c = par.compare_fragment(refs[1], rec[:, :3], rec[:, 3:35], rec[:, 35:36])
    assert c["points_equal"] and c["desc_max_abs"] <= TOL and c["score_max_abs"] <= TOL, c


@pytest.mark.parametrize("which", [0, 1])
def test_demo_self_pair_full_forward_vs_oracle(device, coracle, which):
    """BASELINE configs[0] geometry: the reference's own subsampling of its demo clouds (both), each as a self-pair, limits from the golden
    calibration of the demo pair; exact-shape eager path (the reference's tensor shapes) and the graph engine."""
    from d3feat_amd.datasets.common import FragmentDataset
    from d3feat_amd.engine import FragmentEngine
    from d3feat_amd.models.KPFCNN_model import KernelPointFCNN
    from d3feat_amd.models.variables import build_variables
    from d3feat_amd.utils.config import threedmatch_config
    from oracle import parity as par
    cfg = threedmatch_config()
    W = build_variables(cfg, seed=3, randomize_bn=True).values
    sub = np.load(os.path.join(GOLDEN, "demo_bin%d_sub003.npy" % which))
    assert len(sub) == (14007, 13530)[which]
    limits = np.load(os.path.join(GOLDEN, "preprocess.npz"))["calib_limits_demo_pair"].astype(np.int32)
    assert limits.tolist() == [37, 35, 36, 38, 38]
    ref = par.fragment_reference(cfg, W, None, limits, co=coracle, clouds=[sub, sub])
    L = cfg.num_layers
    # exact-shape path: every matrix has the reference's own shape
    ds = FragmentDataset([sub], fast=False)
    ds.device = device
    ds.neighborhood_limits = limits
    gen, _, _ = ds.get_batch_gen("test", cfg)
    flat = ds.get_tf_mapping(cfg)(*ds._to_device(next(iter(gen()))))
    for l in range(L):
        assert np.array_equal(flat[l].cpu().numpy().view(np.uint32), ref["inp"]["points"][l].view(np.uint32))
        for name, off in (("neighbors", L), ("pools", 2 * L), ("upsamples", 3 * L)):
            w = ref["inp"][name][l]
            if w.shape[0]:
                assert np.array_equal(flat[off + l].cpu().numpy(), w), (name, l)
    assert np.array_equal(flat[4 * L + 2].cpu().numpy(), ref["inp"]["in_batches"])
    model = KernelPointFCNN(flat, cfg, weights=W, device=device)
    d, s = model.out_features.cpu().numpy(), model.out_scores.cpu().numpy()
    assert np.abs(d - ref["desc"]).max() <= TOL and np.abs(s - ref["score"]).max() <= TOL
    # the same cloud through the engine: its stage 0 voxelises the (already voxelised) cloud once more, with the grid origin of
    # THIS cloud -- a different point set / order than `sub` -- so the oracle runs the same stage 0 (fragment_reference(raw))
    eng = FragmentEngine(cfg, W, limits, raw_cap=20000, n0_cap=16000, slots=1, device=device)
    p, d, s = (t.cpu().numpy() for t in eng.run(torch.from_numpy(sub).to(device)))
    assert eng.fallbacks == 0
    ref2 = par.fragment_reference(cfg, W, sub, limits, co=coracle)
    c = par.compare_fragment(ref2, p, d, s)
    assert c["points_equal"] and c["desc_max_abs"] <= TOL and c["score_max_abs"] <= TOL, c
