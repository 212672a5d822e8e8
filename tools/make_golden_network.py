#!/usr/bin/env python3
"""tests/golden/network_*.npz: what the REFERENCE'S OWN PYTHON computes for the floating-point half of the hot path.

The reference's model code is pure Python over stock TensorFlow-1 symbols.  This script puts /root/reference (NOT compat/,
NOT d3feat_amd/) on sys.path together with oracle/tf_eager -- a numpy float32 eager stand-in named `tensorflow` -- imports

    kernels/convolution_ops.py, models/network_blocks.py, models/D3Feat.py, datasets/common.py, utils/config.py

UNMODIFIED (their sha256 go into the MANIFEST) and executes

  * `Dataset.tf_get_batch_inds` + `Dataset.tf_descriptor_input` (datasets/common.py:408-496,1301-1413) with the custom ops served by
    the reference's own C++ (oracle/_ref), and
  * `assemble_FCNN_blocks` (models/D3Feat.py:5-115) in inference mode (dropout_prob = 1.0 -> training False, :22)

on small crops of the reference's demo clouds, recording the inputs, every block's output (by wrapping the function
`get_block_ops` returns -- instrumentation, the reference's code is untouched), every raw KPConv output (wrapping
`conv_ops.KPConv_ops`), descriptors and scores.  Three fixtures:

  network_3dmatch.npz   3DMatch configuration (results/Log_contraloss/parameters.txt), self-pair [c; c] as
                        datasets/ThreeDMatch.py:190-192 stacks it, seeded weights, non-trivial batch-norm statistics
  network_kitti.npz     KITTI configuration (results_kitti/Log_11011605/parameters.txt), two DIFFERENT clouds of unequal
                        length, the reference's REAL trained tensors (kernel_points/epoch61) where its dump has them
  network_ops.npz       direct calls: KPConv_ops in all influence x aggregation modes of convolution_ops.py:208-232,
                        ind_max_pool, closest_pool, unary_convolution, batch_norm + leaky_relu

Large weight tensors are not stored: they are derived from (name, seed) by oracle/seeded_variables.py on both sides, or come from
tests/golden/kitti_epoch61_weights.npz; the fixture holds a digest of each.

    python tools/make_golden_network.py        # needs /root/reference and oracle/_ref (make -C oracle ref)
"""
import hashlib
import json
import os
import sys
import tempfile
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
SEED = 20260926
ROWS = 256          # rows of every block output kept in the fixture

REF_FILES = ["kernels/convolution_ops.py", "kernels/kernel_points.py", "models/network_blocks.py", "models/D3Feat.py",
             "datasets/common.py", "utils/config.py", "utils/ply.py"]


def setup_imports():
    # the stand-in first, then the reference, then this repo's root (for `oracle.*` only: the root has no packages named
    # utils / models / kernels / datasets, so every such import resolves to the reference)
    sys.path[:0] = [os.path.join(ROOT, "oracle", "tf_eager"), REF, ROOT]
    for bad in ("compat", "d3feat_amd"):
        assert not any(p.rstrip("/").endswith(bad) for p in sys.path)
    # the compiled CPython extension datasets/common.py:29 imports is absent from the checkout (and unused on this path)
    for name in ("cpp_wrappers", "cpp_wrappers.cpp_subsampling", "cpp_wrappers.cpp_subsampling.grid_subsampling"):
        sys.modules[name] = types.ModuleType(name)
    # the reference's top-level directories have no __init__.py (namespace packages), so a REGULAR package of the same name
    # in site-packages (HuggingFace `datasets`) would win: bind the names to the reference's directories explicitly
    for name in ("datasets", "kernels", "models", "utils"):
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(REF, name)]
        sys.modules[name] = m
    import tensorflow as tf
    assert "d3f-numpy-eager" in tf.__version__
    return tf


def crop(cloud, centre_index, n):
    """The n points nearest to cloud[centre_index], in their original order."""
    d = np.sum((cloud - cloud[centre_index]) ** 2, axis=1)
    keep = np.sort(np.argsort(d, kind="stable")[:n])
    return np.ascontiguousarray(cloud[keep], np.float32)


class Recorder:
    def __init__(self, tf, conv_ops, network_blocks, d3feat):
        self.tf, self.blocks, self.kpconv = tf, {}, {}
        import tensorflow
        self._scope = tensorflow._scope
        orig_get = network_blocks.get_block_ops
        orig_kp = conv_ops.KPConv_ops

        def get_block_ops(name):
            fn = orig_get(name)

            def run(*a, **k):
                out = fn(*a, **k)
                self.blocks["/".join(self._scope[1:])] = np.ascontiguousarray(out, np.float32)
                return out
            return run

        def KPConv_ops(*a, **k):
            out = orig_kp(*a, **k)
            self.kpconv["/".join(self._scope[1:])] = np.ascontiguousarray(out, np.float32)
            return out
        network_blocks.get_block_ops = get_block_ops
        d3feat.get_block_ops = get_block_ops
        conv_ops.KPConv_ops = KPConv_ops
        self.restore = lambda: (setattr(network_blocks, "get_block_ops", orig_get), setattr(d3feat, "get_block_ops", orig_get),
                                setattr(conv_ops, "KPConv_ops", orig_kp))


def run_case(tf, mods, config, clouds, limits, external=None, tag="", whole=(), rowsums=False, sample_rows=None):
    """clouds: list of stage-0 clouds forming ONE stack (self-pair: [c, c]).  rowsums: additionally keep, for EVERY row of every block
    output, (sum over channels, sum of magnitudes) in float64 -- 16 bytes per row instead of the row: a row-dependent error cannot hide
    behind the sampled rows; whole: scopes whose output is kept whole."""
    from oracle import seeded_variables as sv
    conv_ops, network_blocks, d3feat, common = mods
    L = config.num_layers
    ds = common.Dataset(tag)
    ds.neighborhood_limits = [int(v) for v in limits]
    stacked_points = np.concatenate(clouds, 0).astype(np.float32)
    stack_lengths = np.asarray([len(c) for c in clouds], np.int32)
    stacked_features = tf.ones((tf.shape(stacked_points)[0], 1), dtype=tf.float32)     # datasets/ThreeDMatch.py:293
    batch_inds = ds.tf_get_batch_inds(stack_lengths)
    flat = ds.tf_descriptor_input(config, stacked_points, stacked_features, stack_lengths, batch_inds)
    # models/KPFCNN_model.py:88-109: the flat list -> dictionary
    inputs = dict(points=flat[:L], neighbors=flat[L:2 * L], pools=flat[2 * L:3 * L], upsamples=flat[3 * L:4 * L],
                  features=flat[4 * L], batch_weights=flat[4 * L + 1], in_batches=flat[4 * L + 2], out_batches=flat[4 * L + 3],
                  stack_lengths=stack_lengths)
    spec = []
    stored = {}

    def hook(full, default):
        name = full.split("/", 1)[1] if full.startswith("KernelPointNetwork/") else full
        leaf = name.rsplit("/", 1)[-1]
        key = name.replace("/", "__")
        if external is not None and any(key in src.files for src in external):
            v = next(np.ascontiguousarray(src[key], np.float32) for src in external if key in src.files)
            kind = "external"
        elif leaf == "weights":
            v, kind = sv.seeded_weights(name, default.shape, SEED), "seeded"
        elif leaf in ("gamma", "beta", "moving_mean", "moving_variance"):
            v, kind = sv.seeded_bn(name, default.shape, SEED), "stored"
        else:                                   # kernel_points: what the reference's own load_kernels created
            v, kind = np.ascontiguousarray(default, np.float32), "stored"
        assert v.shape == default.shape, (name, v.shape, default.shape)
        if kind == "stored":
            stored["var/" + name] = v
        spec.append([name, list(v.shape), kind, sv.digest(v)])
        return v

    tf.reset_default_graph()
    tf.set_variable_hook(hook)
    tf.set_random_seed(SEED)
    np.random.seed(SEED % (2 ** 31))             # kernels/kernel_points.py draws its rotations from the global state
    rec = Recorder(tf, conv_ops, network_blocks, d3feat)
    try:
        with tf.variable_scope("KernelPointNetwork"):                       # models/KPFCNN_model.py:129
            desc, score = d3feat.assemble_FCNN_blocks(inputs, config, 1.0)  # dropout_prob fed as 1.0 (utils/tester.py:198)
    finally:
        rec.restore()
        tf.set_variable_hook(None)
    out = dict(stored)
    out["seed"] = np.int64(SEED)
    out["varspec"] = np.asarray(json.dumps(spec))
    out["limits"] = np.asarray(ds.neighborhood_limits, np.int32)
    out["stack_lengths"] = stack_lengths
    out["batch_inds"] = np.asarray(batch_inds, np.int32)
    for l in range(L):
        out["points_%d" % l] = np.asarray(inputs["points"][l], np.float32)
        out["neighbors_%d" % l] = np.asarray(inputs["neighbors"][l], np.int32)
        out["pools_%d" % l] = np.asarray(inputs["pools"][l], np.int32)
        out["upsamples_%d" % l] = np.asarray(inputs["upsamples"][l], np.int32)
    out["features"] = np.asarray(inputs["features"], np.float32)
    out["batch_weights"] = np.asarray(inputs["batch_weights"], np.float32)
    out["in_batches"] = np.asarray(inputs["in_batches"], np.int32)
    out["out_batches"] = np.asarray(inputs["out_batches"], np.int32)
    # block / KPConv outputs are kept at a seeded subset of ROWS (all channels): a transcription error shows in every row, and
    # the full tensors would be 8 MB per fixture; descriptors and scores are kept whole
    rrng = np.random.default_rng(SEED + 7)
    for pre, d in (("block/", rec.blocks), ("kpconv/", rec.kpconv)):
        for k, v in d.items():
            rows = np.sort(rrng.choice(v.shape[0], size=min(v.shape[0], sample_rows or ROWS), replace=False)).astype(np.int32)
            out[pre + k] = np.ascontiguousarray(v[rows])
            out["rows/" + pre + k] = rows
    if rowsums:
        for k, v in rec.blocks.items():
            v64 = v.astype(np.float64)
            out["rowsum/" + k] = np.stack([v64.sum(1), np.abs(v64).sum(1)], 1)
    for k in whole:
        out["whole/" + k] = np.ascontiguousarray(rec.blocks[k])
    out["block_order"] = np.asarray(json.dumps(list(rec.blocks.keys())))
    out["descriptors"] = np.asarray(desc, np.float32)
    out["scores"] = np.asarray(score, np.float32)
    assert out["descriptors"].shape == (len(stacked_points), 32) and out["scores"].shape == (len(stacked_points), 1)
    assert np.isfinite(out["descriptors"]).all() and np.isfinite(out["scores"]).all()
    return out


def run_ops(tf, mods, base):
    """Direct calls of the reference's operator functions (all modes)."""
    conv_ops, network_blocks, _, _ = mods
    rng = np.random.default_rng(SEED + 1)
    out = {}
    q, s = base["points_1"], base["points_0"]
    idx_pool, idx_self = base["pools_0"], base["neighbors_0"]
    cin, cout = 8, 16
    f = rng.standard_normal((len(s), cin)).astype(np.float32)
    f[rng.random(len(s)) < 0.3] *= -1.0                       # rows whose channel sum is negative: the `> 0` count (:250-252)
    f[rng.random(len(s)) < 0.05] = 0.0
    extent = np.float32(0.03) * 1.0
    kp = (rng.standard_normal((15, 3)) * 0.03).astype(np.float32)
    kp[0] = 0
    w = (rng.standard_normal((15, cin, cout)) * 0.2).astype(np.float32)
    out.update(dict(ops_q=q, ops_s=s, ops_idx_pool=idx_pool, ops_idx_self=idx_self, ops_f=f, ops_kp=kp, ops_w=w,
                    ops_extent=np.float32(extent)))
    for infl in ("constant", "linear", "gaussian"):
        for agg in ("sum", "closest"):
            out["kpconv_pool/%s/%s" % (infl, agg)] = np.asarray(
                conv_ops.KPConv_ops(q, s, idx_pool, f, kp, w, float(extent), infl, agg), np.float32)
            out["kpconv_self/%s/%s" % (infl, agg)] = np.asarray(
                conv_ops.KPConv_ops(s, s, idx_self, f, kp, w, float(extent), infl, agg), np.float32)
    out["ind_max_pool"] = np.asarray(network_blocks.ind_max_pool(f, idx_pool), np.float32)
    out["closest_pool"] = np.asarray(network_blocks.closest_pool(f[:len(q)], base["upsamples_0"]), np.float32)
    w2 = (rng.standard_normal((cin, cout)) * 0.3).astype(np.float32)
    out["ops_w2"] = w2
    out["unary"] = np.asarray(conv_ops.unary_convolution(f, w2), np.float32)
    out["leaky"] = np.asarray(network_blocks.leaky_relu(out["unary"]), np.float32)
    return out


def live_check(n_points):
    """No fixture: the reference's Python (under the stand-in) and oracle/network_np.py on a LARGER crop of the demo cloud, same inputs and
    weights, in this process; prints one JSON line with the largest differences (tests/test_oracle_vs_reference_python.py)."""
    tf = setup_imports()
    work = tempfile.mkdtemp(prefix="d3f_golden_net_")
    os.chdir(work)
    import kernels.convolution_ops as conv_ops
    import models.network_blocks as network_blocks
    import models.D3Feat as d3feat
    import datasets.common as common
    from utils.config import Config
    mods = (conv_ops, network_blocks, d3feat, common)
    cfg = Config()
    cfg.load(os.path.join(REF, "results", "Log_contraloss"))
    bin0 = np.load(os.path.join(OUT, "demo_bin0_sub003.npy"))
    c = crop(bin0, 7000, n_points)
    global ROWS
    ROWS = 1 << 30                                    # keep whole block outputs
    a = run_case(tf, mods, cfg, [c, c], [37, 35, 36, 38, 38], tag="live")
    from oracle import network_np as onp
    from oracle import seeded_variables as sv
    W = sv.resolve(a)
    L = cfg.num_layers
    inputs = dict(points=[a["points_%d" % l] for l in range(L)], neighbors=[a["neighbors_%d" % l] for l in range(L)],
                  pools=[a["pools_%d" % l] for l in range(L)], upsamples=[a["upsamples_%d" % l] for l in range(L)],
                  features=a["features"], batch_weights=a["batch_weights"], in_batches=a["in_batches"], out_batches=a["out_batches"],
                  stack_lengths=a["stack_lengths"])
    trace = {}
    desc, score = onp.forward(cfg, W, inputs, trace=trace)
    worst = 0.0
    for scope in json.loads(str(a["block_order"])):
        want = a["block/" + scope]
        worst = max(worst, float(np.abs(trace[scope].numpy() - want).max() / max(1.0, np.abs(want).max())))
    print(json.dumps({"points_per_cloud": int(n_points), "rows": int(len(a["points_0"])), "blocks": len(trace),
                      "block_rel_max": worst, "desc_max_abs": float(np.abs(desc - a["descriptors"]).max()),
                      "score_max_abs": float(np.abs(score - a["scores"]).max())}))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--live-check":
        return live_check(int(sys.argv[2]))
    tf = setup_imports()
    from oracle import clib
    assert clib.ref_available(), "build oracle/_ref first: make -C oracle ref"
    work = tempfile.mkdtemp(prefix="d3f_golden_net_")
    os.chdir(work)                                # kernels/kernel_points.py:190 writes kernels/dispositions/ under the cwd
    import kernels.convolution_ops as conv_ops
    import models.network_blocks as network_blocks
    import models.D3Feat as d3feat
    import datasets.common as common
    from utils.config import Config
    for m in (conv_ops, network_blocks, d3feat, common):
        assert os.path.realpath(m.__file__).startswith(REF + "/"), m.__file__
    mods = (conv_ops, network_blocks, d3feat, common)

    bin0 = np.load(os.path.join(OUT, "demo_bin0_sub003.npy"))
    bin1 = np.load(os.path.join(OUT, "demo_bin1_sub003.npy"))

    cfg = Config()
    cfg.load(os.path.join(REF, "results", "Log_contraloss"))
    c = crop(bin0, 4000, 1000)
    a = run_case(tf, mods, cfg, [c, c], [37, 35, 36, 38, 38], tag="3dmatch")
    np.savez_compressed(os.path.join(OUT, "network_3dmatch.npz"), **a)

    kcfg = Config()
    kcfg.load(os.path.join(REF, "results_kitti", "Log_11011605"))
    k0 = (crop(bin0, 9000, 900) * np.float32(10)).astype(np.float32)
    k1 = (crop(bin1, 2500, 1100) * np.float32(10)).astype(np.float32)
    ext = [np.load(os.path.join(OUT, "kitti_epoch61_weights.npz")), np.load(os.path.join(OUT, "kitti_kernel_points.npz"))]
    b = run_case(tf, mods, kcfg, [k0, k1], [30, 28, 27, 26, 25], external=ext, tag="kitti")
    n_ext = sum(1 for s in json.loads(str(b["varspec"])) if s[2] == "external")
    assert n_ext == 44, n_ext
    np.savez_compressed(os.path.join(OUT, "network_kitti.npz"), **b)

    o = run_ops(tf, mods, a)
    np.savez_compressed(os.path.join(OUT, "network_ops.npz"), **o)

    # a larger crop (4000 points, 8000 stacked rows): every row of every block through its row sums, two blocks whole (round 5)
    c4 = crop(bin0, 7000, 4000)
    order = json.loads(str(a["block_order"]))
    big = run_case(tf, mods, cfg, [c4, c4], [37, 35, 36, 38, 38], tag="3dmatch_4k", rowsums=True, whole=(order[3], order[7]),
                   sample_rows=48)
    np.savez_compressed(os.path.join(OUT, "network_3dmatch_4k.npz"), **big)

    man_p = os.path.join(OUT, "MANIFEST.json")
    man = json.load(open(man_p))
    for fn in ("network_3dmatch.npz", "network_kitti.npz", "network_ops.npz", "network_3dmatch_4k.npz"):
        man["files"][fn] = hashlib.sha256(open(os.path.join(OUT, fn), "rb").read()).hexdigest()
        man.setdefault("generated_by_also", {})[fn] = "tools/make_golden_network.py"
        print("%-22s %.2f MB" % (fn, os.path.getsize(os.path.join(OUT, fn)) / 1e6))
    man["network_reference_sources"] = {f: hashlib.sha256(open(os.path.join(REF, f), "rb").read()).hexdigest() for f in REF_FILES}
    man["network_generated_with"] = "oracle/tf_eager (numpy %s float32 eager stand-in for tensorflow 1.12)" % np.__version__
    json.dump(man, open(man_p, "w"), indent=1)
    print("blocks recorded:", json.loads(str(a["block_order"])))


if __name__ == "__main__":
    main()
