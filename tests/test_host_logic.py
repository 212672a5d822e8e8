"""Host-side logic that needs no GPU: Config (parameters.txt compatibility), variable naming against the reference's
checkpoint index, the checkpoint bundle reader, PLY I/O, kernel-point dispositions, fragment sharding."""
import json
import os
import shutil

import numpy as np
import pytest

from conftest import GOLDEN


def test_config_load_matches_shipped_parameters(tmp_path):
    from d3feat_amd.utils.config import Config, kitti_config, threedmatch_config
    for fname, builtin in (("parameters_3dmatch.txt", threedmatch_config()), ("parameters_kitti.txt", kitti_config())):
        d = tmp_path / fname.split(".")[0]
        d.mkdir()
        shutil.copyfile(os.path.join(GOLDEN, fname), d / "parameters.txt")
        c = Config()
        c.load(str(d))
        assert c.architecture == builtin.architecture
        assert c.num_layers == 5 and c.num_kernel_points == 15 and c.first_features_dim == 64
        for k in ("first_subsampling_dl", "density_parameter", "KP_extent", "KP_influence", "convolution_mode",
                  "fixed_kernel_points", "use_batch_norm", "in_features_dim", "dataset"):
            assert getattr(c, k) == getattr(builtin, k), k
        # save -> load round trip keeps every field the inference path reads
        out = tmp_path / (fname + ".rt")
        out.mkdir()
        c.save(str(out))
        c2 = Config()
        c2.load(str(out))
        for k in ("architecture", "num_layers", "first_features_dim", "first_subsampling_dl", "density_parameter",
                  "KP_extent", "KP_influence", "convolution_mode", "num_kernel_points", "use_batch_norm"):
            assert getattr(c2, k) == getattr(c, k), k


def test_variable_names_and_shapes_match_reference_checkpoint():
    """build_variables creates exactly the inference variables of results/Log_contraloss/snapshots/snap-54.index."""
    from d3feat_amd.models.variables import build_variables
    from d3feat_amd.utils.config import threedmatch_config
    from d3feat_amd.utils.tf_checkpoint import ROOT_SCOPE, is_model_variable
    idx = json.load(open(os.path.join(GOLDEN, "checkpoint_index.json")))
    want = {k[len(ROOT_SCOPE):]: tuple(v["shape"]) for k, v in idx.items() if is_model_variable(k)}
    vs = build_variables(threedmatch_config(), seed=0)
    got = {k: tuple(v.shape) for k, v in vs.values.items()}
    assert set(got) == set(want), (sorted(set(got) - set(want))[:5], sorted(set(want) - set(got))[:5])
    assert got == want
    assert sum(int(np.prod(s)) for s in got.values()) == 14103938            # = payload of the data shard / 4
    n_bn = sum(1 for k in got if k.endswith("gamma"))
    assert n_bn == 37


def test_checkpoint_bundle_reader_roundtrip(tmp_path):
    """Write a tiny bundle in the TF layout by hand, read it back (index parsing, offsets, crc, name filtering)."""
    import struct
    from d3feat_amd.utils import tf_checkpoint as tc

    def varint(x):
        out = b""
        while True:
            b = x & 0x7F
            x >>= 7
            out += bytes([b | (0x80 if x else 0)])
            if not x:
                return out

    rng = np.random.default_rng(0)
    tensors = {"KernelPointNetwork/layer_0/simple_0/weights": rng.standard_normal((15, 1, 4)).astype(np.float32),
               "KernelPointNetwork/layer_0/simple_0/weights/Momentum": np.zeros((15, 1, 4), np.float32),
               "KernelPointNetwork/uplayer_0/last_unary_1/weights": rng.standard_normal((4, 2)).astype(np.float32),
               "global_step": np.asarray(7, np.int64)}
    data, entries, off = b"", [], 0
    for name in sorted(tensors):
        raw = tensors[name].tobytes()
        shape = b"".join(b"\x12" + varint(len(b"\x08" + varint(d))) + b"\x08" + varint(d) for d in tensors[name].shape)
        dt = 1 if tensors[name].dtype == np.float32 else 9
        val = b"\x08" + varint(dt) + b"\x12" + varint(len(shape)) + shape + b"\x20" + varint(off) + b"\x28" + varint(len(raw))
        val += b"\x35" + struct.pack("<I", tc.masked_crc32c(raw))
        entries.append((name.encode(), val))
        data += raw
        off += len(raw)

    def block(kvs):
        body = b""
        for k, v in kvs:                       # no prefix sharing, one restart at 0 (valid per the format)
            body += varint(0) + varint(len(k)) + varint(len(v)) + k + v
        return body + struct.pack("<I", 0) + struct.pack("<I", 1)

    blk = block([(b"", b"\x08\x01")] + entries)
    file = blk + b"\x00" + b"\x00" * 4
    handle = varint(0) + varint(len(blk))
    iblk = block([(b"~", handle)])
    ioff = len(file)
    file += iblk + b"\x00" + b"\x00" * 4
    moff = len(file)
    mblk = block([])
    file += mblk + b"\x00" + b"\x00" * 4
    footer = varint(moff) + varint(len(mblk)) + varint(ioff) + varint(len(iblk))
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", 0xdb4775248b80fb57)
    prefix = str(tmp_path / "snap-1")
    open(prefix + ".index", "wb").write(file + footer)
    idx = tc.read_index(prefix + ".index")
    assert list(idx) == sorted(tensors)
    assert idx["KernelPointNetwork/layer_0/simple_0/weights"].shape == (15, 1, 4)
    with pytest.raises(FileNotFoundError):
        tc.load_checkpoint(prefix)
    open(prefix + ".data-00000-of-00001", "wb").write(data)
    w = tc.load_checkpoint(prefix, verify_crc=True)
    assert set(w) == {"layer_0/simple_0/weights", "uplayer_0/last_unary_1/weights"}
    assert np.array_equal(w["layer_0/simple_0/weights"], tensors["KernelPointNetwork/layer_0/simple_0/weights"])


def test_checkpoint_index_golden_layout():
    idx = json.load(open(os.path.join(GOLDEN, "checkpoint_index.json")))
    keys = list(idx)
    assert keys == sorted(keys) and len(keys) == 196
    last = idx["KernelPointNetwork/uplayer_3/unary_0/weights"]
    assert last["shape"] == [3072, 512] and last["offset"] == 50124296 and last["offset"] + last["size"] == 56415752
    # data is packed back to back in key order
    off = 0
    for k in keys:
        assert idx[k]["offset"] == off
        off += idx[k]["size"]


def test_ply_roundtrip(tmp_path):
    from d3feat_amd.utils.ply import read_ply, read_ply_xyz, write_ply
    rng = np.random.default_rng(1)
    pts = rng.standard_normal((50, 3)).astype(np.float32)
    lab = rng.integers(0, 9, 50).astype(np.int32)
    fn = str(tmp_path / "c.ply")
    assert write_ply(fn, [pts, lab], ["x", "y", "z", "label"])
    d = read_ply(fn)
    assert d.dtype.names == ("x", "y", "z", "label")
    assert np.array_equal(read_ply_xyz(fn), pts) and np.array_equal(d["label"], lab)
    assert not write_ply(fn, [pts, lab[:10]], ["x", "y", "z", "label"])
    assert not write_ply(fn, [pts], ["x", "y"])


def test_kernel_points_disposition():
    """kernels/kernel_points.py: 15 points, one at the centre, inside the sphere of the requested radius; the
    trained KITTI dispositions (tests/golden/kitti_kernel_points.npz) have the same structure."""
    from d3feat_amd.kernels.kernel_points import create_kernel_points
    kp = create_kernel_points(0.045, 15, 1, 3, "center", rng=np.random.default_rng(3)).reshape(15, 3)
    assert kp.shape == (15, 3)
    r = np.linalg.norm(kp, axis=1)
    assert r.min() < 0.1 * 0.045 and r.max() <= 0.045 * 1.05
    d = np.linalg.norm(kp[:, None] - kp[None], axis=-1) + np.eye(15)
    assert d.min() > 0.3 * 0.045                                       # well spread
    g = np.load(os.path.join(GOLDEN, "kitti_kernel_points.npz"))
    assert len(g.files) == 10
    for k in g.files:
        assert g[k].shape == (15, 3)


def test_shard_fragments_partitions():
    from d3feat_amd.parallel import shard_fragments
    for ws in (1, 2, 3, 8):
        parts = [shard_fragments(37, r, ws) for r in range(ws)]
        assert sorted(sum(parts, [])) == list(range(37))
        sizes = np.random.default_rng(ws).integers(15000, 45000, 37)
        parts = [shard_fragments(37, r, ws, sizes=sizes) for r in range(ws)]
        assert sorted(sum(parts, [])) == list(range(37))
        loads = [int(sizes[p].sum()) for p in parts]
        assert max(loads) - min(loads) <= sizes.max()                 # LPT bound


def test_stack_batch_inds_host_logic():
    """datasets/common.py:453-496 quirk: an extra pad column iff all lengths are equal."""
    from oracle import network_np as onp
    a = onp.stack_batch_inds([3, 3])
    assert a.shape == (2, 4) and a[0, -1] == 6 and a[1, 0] == 3
    b = onp.stack_batch_inds([4, 2])
    assert b.shape == (2, 4) and list(b[1]) == [4, 5, 6, 6]


def test_kitti_bin_reader_and_result_files(tmp_path):
    """Formats either side of the path: velodyne .bin in (datasets/KITTI.py:131), the tester's three .npy files out
    (utils/tester.py:208-229: first cloud of the self-pair, ascending score)."""
    from d3feat_amd.utils.results import read_kitti_bin, save_3dmatch_results, select_first_cloud
    rng = np.random.default_rng(3)
    sweep = rng.standard_normal((1000, 4)).astype(np.float32)
    p = tmp_path / "000000.bin"
    sweep.tofile(p)
    xyz = read_kitti_bin(str(p))
    assert xyz.dtype == np.float32 and xyz.flags["C_CONTIGUOUS"] and np.array_equal(xyz, sweep[:, :3])
    (tmp_path / "bad.bin").write_bytes(b"\0" * 20)
    with pytest.raises(ValueError):
        read_kitti_bin(str(tmp_path / "bad.bin"))
    n = 57
    pts = rng.standard_normal((2 * n, 3)).astype(np.float32)
    desc = rng.standard_normal((2 * n, 32)).astype(np.float32)
    score = rng.random((2 * n, 1)).astype(np.float32)
    kp, ft, sc = select_first_cloud(pts, desc, score, n)
    order = np.argsort(score[:n], axis=0).squeeze()                     # the reference's own expression (tester.py:209)
    assert np.array_equal(sc, score[order]) and np.array_equal(kp, pts[order]) and np.array_equal(ft, desc[order])
    assert np.all(np.diff(sc[:, 0]) >= 0)
    paths = save_3dmatch_results(str(tmp_path / "out"), b"7-scenes-redkitchen/seq-01/cloud_bin_12.ply", pts, desc, score, n)
    assert [os.path.relpath(q, str(tmp_path / "out")) for q in paths] == [
        "descriptors/7-scenes-redkitchen/cloud_bin_12.D3Feat.npy", "keypoints/7-scenes-redkitchen/cloud_bin_12.npy",
        "scores/7-scenes-redkitchen/cloud_bin_12.npy"]
    assert np.array_equal(np.load(paths[0]), ft) and np.array_equal(np.load(paths[1]), kp) and np.array_equal(np.load(paths[2]), sc)


def test_sort_form_of_the_voxel_pass_equals_the_reference(coracle):
    """The stage-0 subsampler's sort form (csrc/grid_subsample.hip: gs_sortkey / stable radix sort / gs_heads / gs_voxels /
    gs_accum) restated in numpy: a STABLE sort of the voxel keys makes every voxel one run in input order, the run's head is the
    voxel's first occurrence, voxel ids are the ranks of the heads in input order.  Its voxels, barycentres (in-order fp32 sums x
    float(1 / count)) and first-occurrence numbering must be exactly the reference's (grid_subsampling.cpp:24-94); the output
    ROW order is then derived from the keys in that numbering by the same order rounds as in the hash form."""
    rng = np.random.default_rng(17)
    for n, dl, spread in [(20000, 0.05, 1.0), (5000, 0.03, 0.4), (1, 0.1, 1.0), (300, 0.5, 1.0)]:
        pts = (rng.random((n, 3)) * spread + rng.uniform(-2, 2, 3)).astype(np.float32)
        pts[: min(n, 7)] = pts[0]
        dlf = np.float32(dl)
        inv = np.float32(1.0) / dlf
        org = (np.floor(pts.min(0) * inv) * dlf).astype(np.float32)
        nx = np.floor((pts.max(0)[0] - org[0]) / dlf).astype(np.int64) + 1
        ny = np.floor((pts.max(0)[1] - org[1]) / dlf).astype(np.int64) + 1
        ijk = np.floor((pts - org) / dlf).astype(np.int64)
        key = ijk[:, 0] + nx * ijk[:, 1] + nx * ny * ijk[:, 2]
        order = np.argsort(key, kind="stable")                       # the radix sort: runs in input order
        ks = key[order]
        head = np.concatenate([[True], ks[1:] != ks[:-1]])
        starts = np.flatnonzero(head)
        counts = np.diff(np.concatenate([starts, [n]]))
        first = order[starts]                                        # first occurrence of every voxel (stable => smallest index)
        vid = np.argsort(np.argsort(first))                          # voxel id = rank of the head in input order (the scan)
        bary = np.empty((len(starts), 3), np.float32)
        for r, (s0, c) in enumerate(zip(starts, counts)):
            acc = np.zeros(3, np.float32)
            for i in order[s0:s0 + c]:
                acc = (acc + pts[i]).astype(np.float32)
            bary[vid[r]] = acc * np.float32(1.0 / float(c))
        want = coracle.grid_subsampling(pts, dl)
        assert want.shape == bary.shape
        canon = lambda a: a[np.lexsort((a.view(np.uint32)[:, 2], a.view(np.uint32)[:, 1], a.view(np.uint32)[:, 0]))]
        assert np.array_equal(canon(want).view(np.uint32), canon(bary).view(np.uint32))
        # first-occurrence numbering: voxel v's first point is the v-th distinct key met when walking the input
        seen, walk = set(), []
        for k in key:
            if int(k) not in seen:
                seen.add(int(k))
                walk.append(int(k))
        keys_by_vid = np.empty(len(starts), np.int64)
        keys_by_vid[vid] = ks[starts]
        assert keys_by_vid.tolist() == walk


def test_bench_roofline_selection_never_picks_a_multi_launch_group():
    """bench.py's headline `roofline` object: families are tagged multi-launch by the op RECORD NAME, the
    dominant family is the largest SINGLE-kernel one, whatever the labels look like and however large the launch-gap-dominated
    groups time."""
    import bench
    from d3feat_amd.utils.config import threedmatch_config
    cfg = threedmatch_config()
    timed = []
    for _ in range(2):      # two passes: the subsampling group is the largest "time", then the grid builds, then the contractions
        timed += [("grid_subsample", dict(N=1200000, M=117000), 2.0)] * 5
        timed += [("nb_grid_build", dict(Ns=235000), 0.9)] * 5
        timed += [("gemm_f32", dict(M=235000, N=64, K=128), 0.06)] * 26
        timed += [("kpconv_fused32", dict(Nq=235000, Ns=235000, K=42, Cin=32, Cout=32), 0.4)] * 2
        timed += [("nb_search", dict(Nq=235000, Ns=235000, width=42, first_only=False), 0.12)] * 9
        timed += [("unknown_record", {}, 50.0)]
    fam = bench.accumulate_families(cfg, timed)
    assert "unknown_record" not in fam
    assert fam["grid_subsample"]["multi"] and fam["nb_grid_build"]["multi"]
    assert not any(v["multi"] for k, v in fam.items() if k not in ("grid_subsample", "nb_grid_build"))
    order, dominant = bench.select_dominant(fam)
    assert order[0] == "grid_subsample" and order[1] == "nb_grid_build"
    assert dominant == "gemm_dma_kernel", dominant                    # 2 x 26 x 0.06 = 3.12 ms
    # the searches: 2 x 9 x 0.12 = 2.16 ms < 3.12 ms; a label that happens to contain "launches)" changes nothing
    fam["weird (9 launches)"] = dict(fam["gemm_dma_kernel"], ms=100.0, multi=False)
    assert bench.select_dominant(fam)[1] == "weird (9 launches)"
    g = fam["gemm_dma_kernel"]
    assert g["launches"] == 52 and abs(g["flops"] - 52 * 2.0 * 235000 * 64 * 128) < 1
    # the operand-split contraction is its own family, priced against a sixth of the dense bf16 matrix peak
    fam2 = bench.accumulate_families(cfg, [("gemm_x3", dict(M=235000, N=64, K=128), 0.05)] * 3 + [("gemm_f32", dict(M=235000, N=32, K=64), 0.03)])
    assert set(fam2) == {"gemm_x3_kernel", "gemm_dma_kernel"} and fam2["gemm_x3_kernel"]["launches"] == 3
    assert abs(bench.matrix_peak("gemm_x3_kernel") - bench.MFMA_BF16_PEAK_TF / 6) < 1e-9 and bench.matrix_peak("gemm_dma_kernel") == 157.3
    assert bench.matrix_peak("kpconv_fused32_kernel") == 157.3
    # a run made of groups only has no headline kernel rather than a wrong one
    assert bench.select_dominant({k: v for k, v in fam.items() if v["multi"]})[1] is None
