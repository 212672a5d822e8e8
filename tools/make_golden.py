#!/usr/bin/env python3
"""Generate tests/golden/* from the REFERENCE ITSELF (run in the build container, where /root/reference exists).

The reference ships no tests and no golden vectors for this path, so the pins are made here:
its own C++ -- compiled in place by oracle/Makefile into oracle/_ref, nothing copied -- is run on (a) a slice of
its own demo cloud and (b) the whole demo pair, and the outputs are committed as small fixtures.  The oracle's
C restatement (oracle/d3f_oracle.c), and through it the HIP kernels, are then checked against these files on
machines where /root/reference does not exist (the GPU box).

    python tools/make_golden.py            # rewrites tests/golden/

Contents of tests/golden/:
  demo_bin0_head.npy        f32[20000,3]  first 20 000 raw points of demo_data/cloud_bin_0.ply (input fixture)
  demo_bin0_sub003.npy      f32[14007,3]  reference grid_subsampling(cloud_bin_0, 0.03): real 3DMatch geometry at the
                                          network's input resolution (input fixture for the neighbour / network tests)
  demo_bin1_sub003.npy      f32[13530,3]  the same for cloud_bin_1 (config #1's second cloud)
  preprocess.npz            reference outputs (see keys below)
  checkpoint_index.json     variable names / shapes / offsets decoded from results/Log_contraloss/snapshots/snap-54.index
  parameters_3dmatch.txt    results/Log_contraloss/parameters.txt (config data, for the Config.load round trip)
  parameters_kitti.txt      results_kitti/Log_11011605/parameters.txt
  kitti_kernel_points.npz   the 10 trained kernel-point dispositions of results_kitti/.../epoch61/*.ply (real K_points)
  MANIFEST.json             sha256 of every file + of the reference sources that produced them
"""
import hashlib
import json
import os
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def file_sha(p):
    return hashlib.sha256(open(p, "rb").read()).hexdigest()


def main():
    from oracle import clib
    from oracle import network_np as onp
    from d3feat_amd.utils import ply, tf_checkpoint
    from d3feat_amd.utils.config import Config
    clib.build(ref=True)
    assert clib.ref_available(), "oracle/_ref could not be built"
    ref, refw = clib.RefLib(), clib.RefWrapLib()
    os.makedirs(OUT, exist_ok=True)
    g = {}

    raw0 = ply.read_ply_xyz(os.path.join(REF, "demo_data", "cloud_bin_0.ply"))
    raw1 = ply.read_ply_xyz(os.path.join(REF, "demo_data", "cloud_bin_1.ply"))
    head = np.ascontiguousarray(raw0[:20000])
    np.save(os.path.join(OUT, "demo_bin0_head.npy"), head)

    # ---- grid subsampling (tf_custom_ops/tf_subsampling/grid_subsampling/grid_subsampling.cpp:5-97, :101-149) ------
    g["head_sub_003"] = ref.grid_subsampling(head, 0.03)
    g["head_sub_005"] = ref.grid_subsampling(head, 0.05)
    lens = np.asarray([12000, 7000, 1000], np.int32)
    bp, bl = ref.batch_grid_subsampling(head, lens, 0.04)
    g["head_batch_lens_in"], g["head_batch_sub_004"], g["head_batch_lens_out"] = lens, bp, bl
    # cpp_wrappers core with features and labels (cpp_wrappers/cpp_subsampling/grid_subsampling/grid_subsampling.cpp:5-105)
    rng = np.random.default_rng(0)
    feats = rng.standard_normal((len(head), 3)).astype(np.float32)
    labels = rng.integers(0, 7, (len(head), 2)).astype(np.int32)
    wp, wf, wc = refw.grid_subsampling(head, 0.04, feats, labels)
    g["wrap_features_in"], g["wrap_labels_in"] = feats, labels
    g["wrap_sub_004"], g["wrap_sub_features"], g["wrap_sub_labels"] = wp, wf, wc

    # ---- the whole demo pair at the network's input resolution -----------------------------------------------------
    sub0 = ref.grid_subsampling(raw0, 0.03)
    sub1 = ref.grid_subsampling(raw1, 0.03)
    np.save(os.path.join(OUT, "demo_bin0_sub003.npy"), sub0)
    np.save(os.path.join(OUT, "demo_bin1_sub003.npy"), sub1)
    g["demo_raw_counts"] = np.asarray([len(raw0), len(raw1)], np.int64)
    g["demo_sub_counts"] = np.asarray([len(sub0), len(sub1)], np.int64)
    g["demo_bin1_sub003_sha256"] = np.frombuffer(bytes.fromhex(sha(sub1)), np.uint8)

    # ---- radius neighbours (tf_custom_ops/tf_neighbors/neighbors/neighbors.cpp:211-332 active path, :125-208 stable) --
    pts = np.concatenate([sub0, sub0])
    pl = np.asarray([len(sub0)] * 2, np.int32)
    r0 = np.float32(0.03 * 2.5)
    nano = ref.batch_nanoflann_neighbors(pts, pts, pl, pl, r0)
    brute = ref.batch_ordered_neighbors(pts, pts, pl, pl, r0)
    assert nano.shape == brute.shape
    g["demo_nbr_kmax"] = np.asarray([nano.shape[1]], np.int64)
    g["demo_nbr_counts"] = np.sum(nano < len(pts), axis=1).astype(np.int16)
    g["demo_nbr_nanoflann_sha256"] = np.frombuffer(bytes.fromhex(sha(nano)), np.uint8)
    g["demo_nbr_ordered_sha256"] = np.frombuffer(bytes.fromhex(sha(brute)), np.uint8)
    # rows where the active (unstable sort) path and the stable path disagree: equal-d2 ties only
    diff_rows = np.nonzero(np.any(nano != brute, axis=1))[0]
    g["demo_nbr_tie_rows"] = diff_rows.astype(np.int32)
    g["demo_nbr_tie_rows_nanoflann"] = nano[diff_rows]
    g["demo_nbr_tie_rows_ordered"] = brute[diff_rows]
    # first cloud's block of the stable matrix, 40 columns (indices < 65536 -> uint16 keeps the file small)
    blk = brute[: len(sub0), :40].copy()
    assert blk.max() < 65536
    g["demo_nbr_ordered_first40"] = blk.astype(np.uint16)
    # head cloud: small enough to keep the whole matrices of both reference paths
    hs = g["head_sub_003"]
    hl = np.asarray([len(hs)], np.int32)
    g["head_nbr_nanoflann"] = ref.batch_nanoflann_neighbors(hs, hs, hl, hl, r0).astype(np.int16)
    g["head_nbr_ordered"] = ref.batch_ordered_neighbors(hs, hs, hl, hl, r0).astype(np.int16)
    on = ref.ordered_neighbors(hs[:500], hs, r0)   # OrderedNeighbors: one cloud, pad -1 (neighbors.cpp:58-123)
    g["head_ordered_neighbors_q500"] = on.astype(np.int16)

    # ---- 5-level pyramid of the demo self-pair (call pattern of datasets/common.py:1301-1413) ---------------------
    cfg = Config()
    cfg.load(os.path.join(REF, "results", "Log_contraloss"))
    hist_n = onp.hist_size(cfg)
    full = np.full(cfg.num_layers, hist_n, np.int32)
    inp = onp.descriptor_input(cfg, pts, np.ones((len(pts), 1), np.float32), pl, full,
                               lambda q, s, ql, sl, r: ref.batch_nanoflann_neighbors(q, s, ql, sl, r),
                               lambda p, l, dl: ref.batch_grid_subsampling(p, l, dl))
    g["pyr_sizes"] = np.asarray([p.shape[0] for p in inp["points"]], np.int64)
    g["pyr_kmax_conv"] = np.asarray([m.shape[1] for m in inp["neighbors"]], np.int64)
    g["pyr_kmax_pool"] = np.asarray([m.shape[1] for m in inp["pools"]], np.int64)
    g["pyr_kmax_up"] = np.asarray([m.shape[1] for m in inp["upsamples"]], np.int64)
    for l in range(cfg.num_layers):
        g["pyr_points_sha256_%d" % l] = np.frombuffer(bytes.fromhex(sha(inp["points"][l])), np.uint8)
    hists = onp.neighbor_histograms(inp["neighbors"], hist_n)
    # both self-pairs of the demo are histogrammed by calibrate_neighbors (datasets/common.py:629-670)
    pts1 = np.concatenate([sub1, sub1])
    pl1 = np.asarray([len(sub1)] * 2, np.int32)
    inp1 = onp.descriptor_input(cfg, pts1, np.ones((len(pts1), 1), np.float32), pl1, full,
                                lambda q, s, ql, sl, r: ref.batch_nanoflann_neighbors(q, s, ql, sl, r),
                                lambda p, l, dl: ref.batch_grid_subsampling(p, l, dl))
    hists_both = hists + onp.neighbor_histograms(inp1["neighbors"], hist_n)
    g["pyr_hist_bin0"] = hists
    g["calib_limits_demo_pair"] = onp.limits_from_histograms(hists_both)
    g["pyr_sizes_bin1"] = np.asarray([p.shape[0] for p in inp1["points"]], np.int64)
    np.savez_compressed(os.path.join(OUT, "preprocess.npz"), **g)

    # ---- checkpoint index, configs, trained kernel points ------------------------------------------------------------
    idx = tf_checkpoint.read_index(os.path.join(REF, "results", "Log_contraloss", "snapshots", "snap-54.index"))
    json.dump({k: dict(dtype=e.dtype, shape=list(e.shape), offset=e.offset, size=e.size) for k, e in idx.items()},
              open(os.path.join(OUT, "checkpoint_index.json"), "w"), indent=0)
    shutil.copyfile(os.path.join(REF, "results", "Log_contraloss", "parameters.txt"), os.path.join(OUT, "parameters_3dmatch.txt"))
    shutil.copyfile(os.path.join(REF, "results_kitti", "Log_11011605", "parameters.txt"), os.path.join(OUT, "parameters_kitti.txt"))
    for f in ("parameters_3dmatch.txt", "parameters_kitti.txt"):
        os.chmod(os.path.join(OUT, f), 0o644)
    dumps = tf_checkpoint.load_weight_dumps(os.path.join(REF, "results_kitti", "Log_11011605", "kernel_points", "epoch61"))
    np.savez_compressed(os.path.join(OUT, "kitti_kernel_points.npz"),
                        **{k.replace("/", "__"): v for k, v in dumps.items() if k.endswith("kernel_points")})

    srcs = ["tf_custom_ops/tf_neighbors/neighbors/neighbors.cpp", "tf_custom_ops/tf_subsampling/grid_subsampling/grid_subsampling.cpp",
            "tf_custom_ops/cpp_utils/cloud/cloud.cpp", "tf_custom_ops/cpp_utils/nanoflann/nanoflann.hpp",
            "cpp_wrappers/cpp_subsampling/grid_subsampling/grid_subsampling.cpp", "demo_data/cloud_bin_0.ply",
            "demo_data/cloud_bin_1.ply"]
    man = {"generated_by": "tools/make_golden.py", "reference_sources": {s: file_sha(os.path.join(REF, s)) for s in srcs},
           "files": {f: file_sha(os.path.join(OUT, f)) for f in sorted(os.listdir(OUT)) if f != "MANIFEST.json"},
           "gxx": os.popen("g++ --version").read().splitlines()[0]}
    json.dump(man, open(os.path.join(OUT, "MANIFEST.json"), "w"), indent=1)
    for f in sorted(os.listdir(OUT)):
        print("%10d  %s" % (os.path.getsize(os.path.join(OUT, f)), f))
    print("pyramid sizes", g["pyr_sizes"], "kmax", g["pyr_kmax_conv"], g["pyr_kmax_pool"], g["pyr_kmax_up"])
    print("limits", g["calib_limits_demo_pair"], "tie rows", diff_rows)


if __name__ == "__main__":
    main()
