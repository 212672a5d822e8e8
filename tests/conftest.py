import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def coracle():
    from oracle.clib import COracle
    return COracle()


@pytest.fixture(scope="session")
def reflib():
    from oracle import clib
    if not clib.ref_available():
        pytest.skip("oracle/_ref not built (needs /root/reference at build time)")
    return clib.RefLib()


@pytest.fixture(scope="session")
def refwrap():
    from oracle import clib
    if not clib.ref_available():
        pytest.skip("oracle/_ref not built (needs /root/reference at build time)")
    return clib.RefWrapLib()


@pytest.fixture(scope="session")
def device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda", 0)


GOLDEN = os.path.join(ROOT, "tests", "golden")


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


def small_cloud(seed, n, scale=(2.0, 1.5, 0.7)):
    rng = np.random.default_rng(seed)
    return (rng.random((n, 3)) * np.asarray(scale) - 0.3).astype(np.float32)


def surface_cloud(seed, n_raw=60000, dl=0.03):
    """A small room fragment subsampled by the C oracle (surfaces give realistic neighbour counts)."""
    from d3feat_amd.utils.synthetic import room_fragment
    from oracle.clib import COracle
    raw = room_fragment(seed, n_raw=n_raw, edge=1.0)
    return COracle().grid_subsampling(raw, dl)


def write_tf_bundle(prefix, tensors, crc=True):
    """Write `tensors` {full variable name: ndarray} as a TensorFlow checkpoint bundle (<prefix>.index + .data-00000-of-00001)
    in the on-disk layout d3feat_amd.utils.tf_checkpoint reads: one SSTable data block (no prefix sharing, a single restart),
    BundleEntryProto values with masked crc32c, index block, empty meta-index, 48-byte footer."""
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
    data, entries, off = [], [], 0
    for name in sorted(tensors):
        arr = np.ascontiguousarray(tensors[name])
        raw = arr.tobytes()
        shape = b"".join(b"\x12" + varint(len(b"\x08" + varint(d))) + b"\x08" + varint(d) for d in arr.shape)
        dt = 1 if arr.dtype == np.float32 else 9
        val = b"\x08" + varint(dt) + b"\x12" + varint(len(shape)) + shape + b"\x20" + varint(off) + b"\x28" + varint(len(raw))
        val += b"\x35" + struct.pack("<I", tc.masked_crc32c(raw) if crc else 0)   # crc=False: large test bundles
        entries.append((name.encode(), val))
        data.append(raw)
        off += len(raw)

    def block(kvs):
        body = b""
        for k, v in kvs:
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
    open(prefix + ".index", "wb").write(file + footer)
    open(prefix + ".data-00000-of-00001", "wb").write(b"".join(data))
