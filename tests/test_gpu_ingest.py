"""Stage-0 ingestion on the GPU: xyz decoded from raw PLY / KITTI records on the device equals the
host readers bit for bit, and a fragment submitted as file records gives the engine's result for the decoded cloud."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _write_ply(path, pts, dtype, endian, extra):
    fields = [("x", dtype), ("y", dtype), ("z", dtype)]
    if extra:
        fields = [("nx", "f4")] + fields[:1] + [("red", "u1")] + fields[1:] + [("label", "i4")]
    e = "<" if endian == "little" else ">"
    rec = np.zeros(len(pts), dtype=[(n, e + t) for n, t in fields])
    for i, ax in enumerate("xyz"):
        rec[ax] = pts[:, i]
    names = {"f4": "float", "f8": "double", "u1": "uchar", "i4": "int"}
    with open(path, "wb") as f:
        f.write(b"ply\n" + ("format binary_%s_endian 1.0\n" % endian).encode())
        f.write(("element vertex %d\n" % len(pts)).encode())
        for n, t in fields:
            f.write(("property %s %s\n" % (names[t], n)).encode())
        f.write(b"end_header\n")
        f.write(rec.tobytes())


@pytest.mark.parametrize("dtype,endian,extra", [("f4", "little", False), ("f8", "little", True), ("f4", "big", True),
                                                ("f8", "big", False)])
def test_decode_ply_records_bit_exact(device, tmp_path, dtype, endian, extra):
    from d3feat_amd import ops
    from d3feat_amd.utils.ply import ply_vertex_count, read_ply_records, read_ply_xyz
    rng = np.random.default_rng(0)
    pts = (rng.standard_normal((5003, 3)) * 3).astype(np.float64 if dtype == "f8" else np.float32)
    path = str(tmp_path / "c.ply")
    _write_ply(path, pts, dtype, endian, extra)
    want = read_ply_xyz(path)
    assert ply_vertex_count(path) == 5003
    raw, layout = read_ply_records(path)
    got = ops.decode_xyz_records(raw, layout).cpu().numpy()
    assert got.dtype == np.float32 and np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_decode_kitti_records_and_engine_ingest(device, tmp_path):
    from d3feat_amd import ops
    from d3feat_amd.engine import FragmentEngine
    from d3feat_amd.models.variables import build_variables
    from d3feat_amd.utils.config import threedmatch_config
    from d3feat_amd.utils.results import read_kitti_bin, read_kitti_records
    from d3feat_amd.utils.synthetic import room_fragment
    raw_pts = room_fragment(5, n_raw=30000, edge=1.0)
    rec = np.concatenate([raw_pts, np.random.default_rng(1).random((len(raw_pts), 1), dtype=np.float32)], 1)
    path = str(tmp_path / "000000.bin")
    rec.tofile(path)
    raw, layout = read_kitti_records(path)
    got = ops.decode_xyz_records(raw, layout).cpu().numpy()
    assert np.array_equal(got.view(np.uint32), read_kitti_bin(path).view(np.uint32))
    cfg = threedmatch_config()
    W = build_variables(cfg, seed=42).values
    eng = FragmentEngine(cfg, W, np.asarray([37, 35, 36, 38, 38], np.int32), raw_cap=40000, n0_cap=12000, slots=1, device=device)
    a = tuple(t.clone() for t in eng.run(ops.RawRecords(raw, layout)))
    b = eng.run(torch.from_numpy(raw_pts).to(device))
    assert eng.fallbacks == 0
    assert all(torch.equal(x, y) for x, y in zip(a, b))
