"""The floating-point oracle and the pyramid restatement against what the REFERENCE'S OWN PYTHON computed.

tests/golden/network_*.npz were produced by tools/make_golden_network.py: /root/reference's kernels/convolution_ops.py,
models/network_blocks.py, models/D3Feat.py and datasets/common.py imported UNMODIFIED and executed under oracle/tf_eager (numpy
float32 eager `tensorflow`), custom ops served by the reference's C++.  This pins oracle/network_np.py (every block output, all
influence / aggregation modes, descriptors, scores) and oracle/network_np.descriptor_input (+ the C geometry restatement) to the
reference itself; tests/test_gpu_golden_network.py holds the HIP path to the same fixtures.

Tolerance: both sides are float32 with different summation orders (numpy pairwise / BLAS there, torch here): 2e-6 relative to
the tensor's largest magnitude, per block; 1e-5 absolute on descriptors (unit vectors) and scores."""
import numpy as np
import pytest

from oracle import network_np as onp
from oracle.golden_network import GoldenNetwork, ops_fixture

REL = 2e-6


def _close(got, want, what, rel=REL):
    scale = max(1.0, float(np.abs(want).max()))
    err = float(np.abs(np.asarray(got, np.float64) - want).max())
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert err <= rel * scale, "%s: max abs %.3e (scale %.2f)" % (what, err, scale)


@pytest.mark.parametrize("name", ["3dmatch", "kitti"])
def test_network_restatement_equals_the_reference_python(name):
    g = GoldenNetwork(name)
    cfg = g.config()
    trace = {}
    desc, score = onp.forward(cfg, g.W, g.inputs, trace=trace)
    assert list(trace.keys()) == g.block_order          # same blocks, same order, same variable scopes
    for scope in g.block_order:
        rows, want = g.block(scope)
        _close(trace[scope].numpy()[rows], want, scope)
    _close(desc, g.descriptors, "descriptors", rel=1e-5)
    _close(score, g.scores, "scores", rel=1e-5)
    assert np.abs(score - g.scores).max() <= 1e-5 and np.abs(desc - g.descriptors).max() <= 1e-5


def test_network_restatement_on_the_larger_crop_every_row():
    """network_3dmatch_4k.npz: 8000 stacked rows, the row sums of EVERY row of every block, two blocks whole (round 5)."""
    g = GoldenNetwork("3dmatch_4k")
    trace = {}
    desc, score = onp.forward(g.config(), g.W, g.inputs, trace=trace)
    assert list(trace.keys()) == g.block_order
    for scope in g.block_order:
        have = trace[scope].numpy().astype(np.float64)
        want = g.rowsum(scope)
        scale = max(1.0, float(want[:, 1].max()) / have.shape[1])
        assert np.abs(have.sum(1) - want[:, 0]).max() <= 2e-5 * scale * np.sqrt(have.shape[1]), scope
        rows, w = g.block(scope)
        _close(have[rows], w, scope)
    for scope in g.whole_scopes():
        _close(trace[scope].numpy(), g.whole(scope), scope)
    assert np.abs(score - g.scores).max() <= 1e-5 and np.abs(desc - g.descriptors).max() <= 1e-5


@pytest.mark.parametrize("name", ["3dmatch", "kitti"])
def test_every_kpconv_layer_equals_the_reference_python(name):
    """Each of the 10 KPConv ops alone, fed the reference's own block inputs is not possible (inputs are not in the fixture), so
    the raw KPConv outputs recorded inside the reference's run are compared with the restatement's at the same place."""
    g = GoldenNetwork(name)
    cfg = g.config()
    got = {}
    orig = onp.KPConv_ops

    def rec(*a, **k):
        out = orig(*a, **k)
        got[len(got)] = out
        return out
    onp.KPConv_ops = rec
    try:
        onp.forward(cfg, g.W, g.inputs)
    finally:
        onp.KPConv_ops = orig
    scopes = g.kpconv_scopes()
    assert len(scopes) == len(got) == 10
    for i, scope in enumerate(scopes):
        rows, want = g.kpconv(scope)
        _close(got[i].numpy()[rows], want, "kpconv " + scope)


def _equal_up_to_ties(got, want, q, s, what):
    from oracle.parity import equal_up_to_ties
    ok, ties = equal_up_to_ties(got, want, q, s)
    assert ok, what
    return ties


@pytest.mark.parametrize("name", ["3dmatch", "kitti"])
def test_pyramid_restatement_equals_the_reference_python(name, coracle):
    """datasets/common.py:1301-1413 executed by the reference vs oracle/network_np.descriptor_input over the C restatement: every
    matrix of every level bit-equal, incl. in_batches / out_batches (:453-496) and batch_weights."""
    g = GoldenNetwork(name)
    cfg = g.config()
    pts, lens = g.inputs["points"][0], g.inputs["stack_lengths"]
    inp = onp.descriptor_input(cfg, pts, np.ones((len(pts), 1), np.float32), lens, g.limits,
                               lambda q, s, ql, sl, r: coracle.batch_neighbors(q, s, ql, sl, r),
                               lambda p, l, dl: coracle.batch_grid_subsampling(p, l, dl))
    tie_rows = 0
    for l in range(g.L):
        assert np.array_equal(inp["points"][l].view(np.uint32), g.inputs["points"][l].view(np.uint32)), l
        for key in ("neighbors", "pools", "upsamples"):
            got, want = inp[key][l], g.inputs[key][l]
            assert got.shape == want.shape, (key, l)
            if want.shape[0] == 0:
                continue
            q = g.inputs["points"][l + 1 if key == "pools" else l]
            s = g.inputs["points"][l + 1 if key == "upsamples" else l]
            tie_rows += _equal_up_to_ties(got, want, q, s, (key, l))
    assert tie_rows <= 64           # rows holding a run of bit-equal distances: a fraction of a percent of ~8 000 rows
    for key in ("in_batches", "out_batches", "stack_lengths"):
        assert np.array_equal(inp[key], g.inputs[key]), key
    assert np.array_equal(inp["batch_weights"].view(np.uint32), g.inputs["batch_weights"].view(np.uint32))
    assert np.array_equal(onp.get_batch_inds(lens), g.z["batch_inds"])


@pytest.mark.parametrize("influence", ["constant", "linear", "gaussian"])
@pytest.mark.parametrize("mode", ["sum", "closest"])
def test_kpconv_modes_equal_the_reference_python(influence, mode):
    """All modes of kernels/convolution_ops.py:208-232, strided (queries != supports) and not."""
    z = ops_fixture()
    ext = float(z["ops_extent"])
    for tag, q, idx in (("pool", z["ops_q"], z["ops_idx_pool"]), ("self", z["ops_s"], z["ops_idx_self"])):
        got = onp.KPConv_ops(q, z["ops_s"], idx, z["ops_f"], z["ops_kp"], z["ops_w"], ext, influence, mode).numpy()
        _close(got, z["kpconv_%s/%s/%s" % (tag, influence, mode)], "%s %s %s" % (tag, influence, mode), rel=4e-6)


def test_pools_and_unary_equal_the_reference_python():
    import torch
    z = ops_fixture()
    f = torch.from_numpy(z["ops_f"])
    assert np.array_equal(onp.ind_max_pool(f, z["ops_idx_pool"]).numpy(), z["ind_max_pool"])
    nq = z["ops_q"].shape[0]
    g = GoldenNetwork("3dmatch")
    assert np.array_equal(onp.closest_pool(f[:nq], g.inputs["upsamples"][0]).numpy(), z["closest_pool"])
    _close(onp.unary_convolution(f, torch.from_numpy(z["ops_w2"])).numpy(), z["unary"], "unary")
    assert np.array_equal(onp.leaky_relu(torch.from_numpy(z["unary"])).numpy(), z["leaky"])


def test_fixture_variables_are_the_checkpoint_names():
    """The variable names the reference's code created under the stand-in are exactly the names of its released checkpoints
    (tests/golden/checkpoint_index.json, decoded from results/Log_contraloss/snapshots/snap-54.index)."""
    import json
    import os
    from conftest import GOLDEN
    g = GoldenNetwork("3dmatch")
    idx = json.load(open(os.path.join(GOLDEN, "checkpoint_index.json")))
    want = {n[len("KernelPointNetwork/"):]: list(e["shape"]) for n, e in idx.items() if n.startswith("KernelPointNetwork/")}
    have = {k: list(v.shape) for k, v in g.W.items()}
    assert have == {k: want[k] for k in have}
    assert set(have) == set(want)
