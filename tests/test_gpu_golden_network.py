"""The HIP path against what the REFERENCE'S OWN PYTHON computed (tests/golden/network_*.npz: kernels/convolution_ops.py,
models/network_blocks.py, models/D3Feat.py, datasets/common.py executed unmodified by tools/make_golden_network.py; nothing of
oracle/network_np.py is involved here -- these tests compare the GPU with the reference directly).

Bar (BASELINE.json north_star): descriptors / scores within 1e-4 absolute, index matrices bit-exact (the reference's active
nanoflann path leaves the order inside runs of bit-equal distances unspecified: such rows are compared as the same set with the
same distance column by column, the contract of SURVEY.md section 8c)."""
import numpy as np
import pytest
import torch

from oracle.golden_network import GoldenNetwork, ops_fixture

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _dev(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def _flat(g, device):
    """The positional list of datasets/common.py:1410-1413 + the dataset tail (datasets/ThreeDMatch.py:322), from the fixture."""
    i = g.inputs
    flat = [_dev(p, device) for p in i["points"]] + [_dev(m, device) for m in i["neighbors"]]
    flat += [_dev(m, device) for m in i["pools"]] + [_dev(m, device) for m in i["upsamples"]]
    flat += [_dev(i["features"], device), _dev(i["batch_weights"], device), _dev(i["in_batches"], device),
             _dev(i["out_batches"], device)]
    flat += [_dev(i["stack_lengths"], device), torch.zeros(1, dtype=torch.int32), torch.zeros(1, dtype=torch.int32),
             ["a", "b"], _dev(i["points"][0], device)]
    return flat


def _record_blocks():
    """Wrap the package's get_block_ops so that every block's output is kept under its variable scope."""
    from d3feat_amd import ops
    from d3feat_amd.models import D3Feat as d3, network_blocks as nb
    got = {}
    orig = nb.get_block_ops

    def get_block_ops(name):
        fn = orig(name)

        def run(*a, **k):
            out = fn(*a, **k)
            scope = "/".join(nb._vs()._scope)
            got[scope] = out.materialize() if isinstance(out, ops.UpsampleCat) else out
            return out
        return run
    nb.get_block_ops = d3.get_block_ops = get_block_ops
    return got, lambda: (setattr(nb, "get_block_ops", orig), setattr(d3, "get_block_ops", orig))


@pytest.mark.parametrize("name", ["3dmatch", "kitti"])
def test_forward_equals_the_reference_python(device, name):
    """assemble_FCNN_blocks of the reference (models/D3Feat.py:5-115) vs the HIP model on the reference's own inputs: every block
    output the HIP path materialises, descriptors, scores.  3dmatch: self-pair, seeded weights; kitti: two different clouds of
    unequal length, the reference's real trained tensors."""
    from d3feat_amd.models.KPFCNN_model import KernelPointFCNN
    g = GoldenNetwork(name)
    cfg = g.config()
    got, restore = _record_blocks()
    try:
        model = KernelPointFCNN(_flat(g, device), cfg, weights=dict(g.W))
    finally:
        restore()
    d, s = model.out_features.cpu().numpy(), model.out_scores.cpu().numpy()
    assert d.shape == g.descriptors.shape and s.shape == g.scores.shape
    ed, es = np.abs(d - g.descriptors).max(), np.abs(s - g.scores).max()
    assert ed <= TOL and es <= TOL, (ed, es)
    checked = 0
    for scope in g.block_order:
        if scope not in got:
            continue                      # nearest_upsample blocks stay lazy on the HIP path (contracted by the next unary)
        rows, want = g.block(scope)
        have = got[scope].cpu().numpy()[rows]
        scale = max(1.0, float(np.abs(want).max()))
        assert have.shape == want.shape and np.abs(have - want).max() <= TOL * scale, (scope, np.abs(have - want).max(), scale)
        checked += 1
    assert checked >= 15


def test_every_row_of_every_block_on_a_larger_crop(device):
    """tests/golden/network_3dmatch_4k.npz (round 5): a 4000-point crop as a self-pair -- 8000 stacked rows, sixteen 256-thread
    tiles per level-0 kernel instead of four -- with, for EVERY row of every block output, the reference's (sum over channels, sum
    of magnitudes): an error that depends on the row (a tile edge, a shadow slot, the last workgroup) cannot hide behind the
    sampled rows of the small fixtures.  Two blocks are kept whole, descriptors and scores too."""
    from d3feat_amd.models.KPFCNN_model import KernelPointFCNN
    g = GoldenNetwork("3dmatch_4k")
    cfg = g.config()
    got, restore = _record_blocks()
    try:
        model = KernelPointFCNN(_flat(g, device), cfg, weights=dict(g.W))
    finally:
        restore()
    d, s = model.out_features.cpu().numpy(), model.out_scores.cpu().numpy()
    assert d.shape == g.descriptors.shape == (8000, 32)
    ed, es = np.abs(d - g.descriptors).max(), np.abs(s - g.scores).max()
    assert ed <= TOL and es <= TOL, (ed, es)
    checked = 0
    for scope in g.block_order:
        if scope not in got:
            continue
        have = got[scope].cpu().numpy().astype(np.float64)
        want = g.rowsum(scope)
        assert have.shape[0] == want.shape[0], scope
        scale = max(1.0, float(want[:, 1].max()) / have.shape[1])          # the block's typical magnitude
        # a row sum of C terms, each within 1e-4 of the block's scale in the worst case and ~1e-6 in practice
        err = np.abs(have.sum(1) - want[:, 0]).max()
        assert err <= TOL * scale * np.sqrt(have.shape[1]), (scope, err, scale)
        assert np.abs(np.abs(have).sum(1) - want[:, 1]).max() <= TOL * scale * np.sqrt(have.shape[1]), scope
        checked += 1
    assert checked >= 15
    for scope in g.whole_scopes():
        want = g.whole(scope)
        have = got[scope].cpu().numpy()
        assert have.shape == want.shape and np.abs(have - want).max() <= TOL * max(1.0, float(np.abs(want).max())), scope
    assert len(g.whole_scopes()) == 2


@pytest.mark.parametrize("name", ["3dmatch", "kitti"])
def test_every_kpconv_layer_equals_the_reference_python(device, name):
    """The 10 KPConv_ops calls of the reference's run, one by one through kernels.convolution_ops.KPConv_ops (every kernel form:
    Cin = 1, fused 32, fused 64 / 128, aggregate + contraction 256 / 512), each fed the ORACLE-FREE input it has in the HIP run."""
    from d3feat_amd.kernels import convolution_ops as co
    from d3feat_amd.models.KPFCNN_model import KernelPointFCNN
    g = GoldenNetwork(name)
    cfg = g.config()
    outs = []
    orig = co.KPConv_ops

    def rec(*a, **k):
        k2 = dict(k)
        k2["epilogue"] = None                     # the raw convolution, as convolution_ops.py:161-255 returns it
        outs.append(orig(*a, **k2))
        return orig(*a, **k)
    co.KPConv_ops = rec
    try:
        KernelPointFCNN(_flat(g, device), cfg, weights=dict(g.W))
    finally:
        co.KPConv_ops = orig
    scopes = g.kpconv_scopes()
    assert len(outs) == len(scopes) == 10
    for o, scope in zip(outs, scopes):
        rows, want = g.kpconv(scope)
        have = o.cpu().numpy()[rows]
        scale = max(1.0, float(np.abs(want).max()))
        assert np.abs(have - want).max() <= TOL * scale, (scope, np.abs(have - want).max(), scale)


@pytest.mark.parametrize("influence", ["constant", "linear", "gaussian"])
@pytest.mark.parametrize("mode", ["sum", "closest"])
def test_kpconv_modes_equal_the_reference_python(device, influence, mode):
    """All influence x aggregation modes of kernels/convolution_ops.py:208-232, strided and not."""
    from d3feat_amd.kernels import convolution_ops as co
    z = ops_fixture()
    ext = float(z["ops_extent"])
    s, f, w = _dev(z["ops_s"], device), _dev(z["ops_f"], device), _dev(z["ops_w"], device)
    for tag, q, idx in (("pool", z["ops_q"], z["ops_idx_pool"]), ("self", z["ops_s"], z["ops_idx_self"])):
        got = co.KPConv_ops(_dev(q, device), s, _dev(idx, device), f, z["ops_kp"], w, ext, influence, mode).cpu().numpy()
        want = z["kpconv_%s/%s/%s" % (tag, influence, mode)]
        assert np.abs(got - want).max() <= TOL * max(1.0, float(np.abs(want).max())), (tag, np.abs(got - want).max())


def test_pools_and_unary_equal_the_reference_python(device):
    from d3feat_amd.kernels import convolution_ops as co
    from d3feat_amd.models import network_blocks as nb
    z = ops_fixture()
    g = GoldenNetwork("3dmatch")
    f = _dev(z["ops_f"], device)
    assert np.array_equal(nb.ind_max_pool(f, _dev(z["ops_idx_pool"], device)).cpu().numpy(), z["ind_max_pool"])
    nq = z["ops_q"].shape[0]
    assert np.array_equal(nb.closest_pool(f[:nq].contiguous(), _dev(g.inputs["upsamples"][0], device)).cpu().numpy(), z["closest_pool"])
    u = co.unary_convolution(f, _dev(z["ops_w2"], device)).cpu().numpy()
    assert np.abs(u - z["unary"]).max() <= 1e-5
    assert np.array_equal(nb.leaky_relu(_dev(z["unary"], device)).cpu().numpy(), z["leaky"])


def _equal_up_to_ties(got, want, q, s, what):
    from oracle.parity import equal_up_to_ties
    ok, ties = equal_up_to_ties(got, want, q, s)
    assert ok, what
    return ties


@pytest.mark.parametrize("name", ["3dmatch", "kitti"])
def test_pyramid_equals_the_reference_python(device, name):
    """Dataset.tf_descriptor_input of the reference (datasets/common.py:1301-1413, its C++ ops underneath) vs the HIP pyramid in
    its exact-shape form: points bit-equal, index matrices equal (up to the order inside bit-equal-distance runs), in_batches /
    out_batches / batch_weights equal."""
    from d3feat_amd.datasets.common import FragmentDataset
    g = GoldenNetwork(name)
    cfg = g.config()
    ds = FragmentDataset([], fast=False)
    ds.device = device
    ds.neighborhood_limits = g.limits
    pts = _dev(g.inputs["points"][0], device)
    lens = _dev(g.inputs["stack_lengths"], device)
    feats = torch.ones((pts.shape[0], 1), dtype=torch.float32, device=device)
    batch_inds = ds.tf_get_batch_inds(lens)
    assert np.array_equal(batch_inds.cpu().numpy(), g.z["batch_inds"])
    flat = ds.tf_descriptor_input(cfg, pts, feats, lens, batch_inds, exact_shapes=True, up_first_column_only=False)
    L, ties = g.L, 0
    for l in range(L):
        assert np.array_equal(flat[l].cpu().numpy().view(np.uint32), g.inputs["points"][l].view(np.uint32)), l
    for l in range(L):
        for key, off in (("neighbors", L), ("pools", 2 * L), ("upsamples", 3 * L)):
            want = g.inputs[key][l]
            have = flat[off + l].cpu().numpy()
            assert have.shape == want.shape, (key, l, have.shape, want.shape)
            if want.shape[0] == 0:
                continue
            q = g.inputs["points"][l + 1 if key == "pools" else l]
            s = g.inputs["points"][l + 1 if key == "upsamples" else l]
            ties += _equal_up_to_ties(have, want, q, s, (key, l))
    assert ties <= 64
    assert np.array_equal(flat[4 * L + 2].cpu().numpy(), g.inputs["in_batches"])
    assert np.array_equal(flat[4 * L + 3].cpu().numpy(), g.inputs["out_batches"])
    assert np.array_equal(flat[4 * L + 1].cpu().numpy().view(np.uint32), g.inputs["batch_weights"].view(np.uint32))


def test_engine_replay_equals_the_reference_python(device):
    """The production execution -- FragmentEngine: capacity-mode pyramid + network captured as ONE HIP graph and replayed -- on the
    fixture's cloud (already at 0.03 m: stage0=False, the cloud is stacked with itself like datasets/ThreeDMatch.py:190-192) against
    the descriptors and scores the reference's Python produced; also as one of three fragments of a batched replay."""
    from d3feat_amd.engine import FragmentEngine
    g = GoldenNetwork("3dmatch")
    cfg = g.config()
    cloud = g.clouds()[0]
    assert np.array_equal(cloud.view(np.uint32), g.clouds()[1].view(np.uint32))
    for batch in (1, 3):
        eng = FragmentEngine(cfg, dict(g.W), g.limits, n0_cap=1536, level_ratio=0.45, slots=1, device=device, batch=batch, stage0=False)
        other = np.ascontiguousarray(cloud[::-1] + np.float32(0.5))          # a different stack mate (batch 3)
        frs = [_dev(cloud, device)] if batch == 1 else [_dev(other, device), _dev(cloud, device), _dev(other[:700], device)]
        eng.submit(0, frs)
        outs = eng.fetch(0)
        assert eng.fallbacks == 0
        p, d, s = (t.cpu().numpy() for t in outs[0 if batch == 1 else 1])
        assert np.array_equal(p.view(np.uint32), g.inputs["points"][0].view(np.uint32))
        ed, es = np.abs(d - g.descriptors).max(), np.abs(s - g.scores).max()
        assert ed <= TOL and es <= TOL, (batch, ed, es)
