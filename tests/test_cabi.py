"""The drop-in boundary: libd3feat_amd.so loads on a machine without a GPU and exports exactly the entry points that
include/d3feat_amd.h declares, with the argument lists the ctypes binding (d3feat_amd/_lib.py) assumes.
No compute call is made here (no GPU in the CPU test tier)."""
import ctypes
import os
import re
import subprocess

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "d3feat_amd.h")

_C2CT = {"int": ctypes.c_int, "float": ctypes.c_float, "size_t": ctypes.c_size_t, "uint64_t": ctypes.c_uint64}


def _declarations():
    """-> {name: (return type, [argument C types])} parsed from the header."""
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    out = {}
    for m in re.finditer(r"\b(int|size_t)\s+(d3f_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        types = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    types.append("ptr")
                else:
                    types.append(a.replace("const ", "").split()[0])
        out[name] = (ret, types)
    return out


@pytest.fixture(scope="module")
def lib():
    from d3feat_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "d3feat_amd", "csrc")])
    return _lib.load()


def test_header_declares_the_expected_entry_points():
    decl = _declarations()
    for must in ("d3f_batch_grid_subsample", "d3f_batch_radius_neighbors", "d3f_neighbor_grid_build",
                 "d3f_neighbor_grid_search", "d3f_kpconv_aggregate", "d3f_kpconv_fused_c1", "d3f_gemm_f32",
                 "d3f_ind_max_pool", "d3f_closest_pool_cat", "d3f_detect_head", "d3f_affine_act", "d3f_version"):
        assert must in decl, must
    # plain C ABI: no torch / HIP types anywhere in the signatures
    code = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    assert "torch" not in code and "hipStream_t" not in code and "#include <hip" not in code


def test_library_exports_every_declared_symbol(lib):
    for name in _declarations():
        assert hasattr(lib, name), "libd3feat_amd.so does not export %s" % name


def test_binding_matches_header():
    from d3feat_amd import _lib
    decl = _declarations()
    assert set(decl) == set(_lib.SIGNATURES), set(decl) ^ set(_lib.SIGNATURES)
    for name, (ret, types) in decl.items():
        res, args = _lib.SIGNATURES[name]
        assert res is _C2CT[ret], name
        assert len(args) == len(types), "%s: header has %d arguments, binding %d" % (name, len(types), len(args))
        for i, (t, a) in enumerate(zip(types, args)):
            want = ctypes.c_void_p if t == "ptr" else _C2CT[t]
            assert a is want, "%s argument %d: header %s, binding %s" % (name, i, t, a)


def test_exports_are_unmangled_c_symbols(lib):
    from d3feat_amd import _lib
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r"\bT (d3f_[a-z0-9_]+)\b", nm))
    assert set(_declarations()) <= exported


def test_host_side_argument_checks_need_no_gpu(lib):
    """Entry points validate sizes before touching the device: invalid arguments come back as D3F_ERR_ARG."""
    assert lib.d3f_version() >= 100
    assert lib.d3f_gemm_f32(None, 0, None, 0, None, 0, -1, 4, 4, None, None, None, None, 0, 0, 0.2, None, 0, None, 0, None) == -3
    assert lib.d3f_batch_grid_subsample(None, 10, None, 1, 0.1, None, 0, None, 0, None, None, None, None, None, None, 0, None) == -3
    assert lib.d3f_neighbor_grid_build(None, 5, None, 0, 0.1, None, 0, None) == -3
    assert lib.d3f_grid_subsample_workspace_bytes(30000, 2, 0, 0) > 30000 * 4 * 14
    assert lib.d3f_gemm_workspace_bytes(390, 512, 7680, 0) >= 256
    assert lib.d3f_neighbor_grid_bytes(60000, 2) > 60000 * 16
    # the one-kernel KPConv forms address rows with 24-bit multiplies: row counts / leading dimensions beyond that are refused
    # before anything is launched (include/d3feat_amd.h; the two-kernel form has no such limit)
    import ctypes
    buf = ctypes.create_string_buffer(256)
    p = ctypes.addressof(buf)
    kp = (ctypes.c_float * 45)()
    args = lambda Nq, Ns, ld_idx, ldf: (p, Nq, p, Ns, p, ld_idx, 8, p, ldf, p, ctypes.addressof(kp), 15, 0.05, 1, 0, p, None, None, None, 0, 1, 0.2,
                                        p, 32, None, None, None, 0, None)
    assert lib.d3f_kpconv_fused32(*args(100, 1 << 24, 8, 32)) == -3
    assert lib.d3f_kpconv_fused32(*args(100, 1 << 20, 8, 1 << 12)) == -3        # Ns * ldf >= 2^31
    assert lib.d3f_kpconv_fused32(*args(1 << 24, 100, 8, 32)) == -3


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from d3feat_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.D3FeatLibraryError):
        _lib.load()


def test_cpu_tensors_are_rejected():
    """There is no CPU fallback: handing a host tensor to an op raises instead of computing somewhere else."""
    import torch
    from d3feat_amd import _lib, ops
    with pytest.raises(_lib.D3FeatLibraryError):
        ops.gemm(torch.zeros(4, 4), torch.zeros(4, 4))


def _plan(M, N, K, hint=0):
    from d3feat_amd import _lib
    r, c, s = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    assert _lib.load().d3f_gemm_x3_plan(M, N, K, hint, ctypes.byref(r), ctypes.byref(c), ctypes.byref(s)) == 0
    return r.value, c.value, s.value


def test_gemm_x3_plans():
    """The cost model of the launcher (rounds of resident workgroups x k-tiles per slice): which of tests/test_gpu_gemm_x3.py's problems run as
    128 x 32, 128 x 64 and 256 x 128 workgroups, split in K or not -- so that every kernel instantiation is exercised there.  Host code only: no GPU."""
    assert _plan(20000, 32, 64) == (128, 32, 1) and _plan(20000, 64, 128) == (128, 64, 1)
    assert _plan(20001, 256, 1024)[:2] == (256, 128) and _plan(20001, 200, 1024)[:2] == (256, 128) and _plan(30000, 256, 256)[:2] == (256, 128)
    assert _plan(4525, 512, 3072) == (256, 128, 3)                        # the network's level-3 decoder contraction at F = 5
    rows, cols, s = _plan(900, 256, 3840)
    assert (rows, cols) == (128, 64) and s > 1                            # skinny deep layer: split in K
    assert _plan(300000, 64, 256, 171000)[:2] == (128, 64)
    from d3feat_amd import _lib
    assert _lib.load().d3f_gemm_x3_plan(100, 64, 48, 0, None, None, None) == -3
