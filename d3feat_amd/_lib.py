"""ctypes binding of libd3feat_amd.so (the C ABI declared in include/d3feat_amd.h).

The library is the ONLY compute path: there is no CPU or PyTorch fallback.  If the shared object has not been
built (python -c "import __graft_entry__ as g; g.build()"  or  make -C d3feat_amd/csrc) every op raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# D3FEAT_AMD_LIB: another build of the same library (A/B measurements of kernel changes inside one GPU visit)
LIB_PATH = os.environ.get("D3FEAT_AMD_LIB") or os.path.join(_HERE, "lib", "libd3feat_amd.so")

D3F_OK = 0
ERRORS = {-1: "HIP runtime / kernel launch failure", -2: "workspace too small", -3: "invalid argument"}
ST_EMPTY_ELEMENT, ST_NEG_CELL, ST_KEY_RANGE, ST_HIT_OVERFLOW, ST_OUT_OVERFLOW, ST_KEY_WIDTH = 1, 2, 4, 8, 16, 32
PAD_NUM_SUPPORTS = -2147483648
NEIGHBOR_CAP = 1024
MAX_BATCH = 255

_vp, _sz, _i, _f = C.c_void_p, C.c_size_t, C.c_int, C.c_float

# name -> (restype, argtypes): mirrors include/d3feat_amd.h one to one (checked by tests/test_cabi.py)
SIGNATURES = {
    "d3f_version": (_i, []),
    "d3f_trace_marker": (_i, [_i, _vp]),
    "d3f_pack_status": (_i, [_vp, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _vp]),
    "d3f_grid_subsample_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "d3f_batch_grid_subsample": (_i, [_vp, _i, _vp, _i, _f, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "d3f_batch_grid_subsample_async": (_i, [_vp, _i, _vp, _i, _f, _vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "d3f_batch_grid_subsample_async_inplace": (_i, [_vp, _i, _vp, _i, _f, _vp, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "d3f_stack_self_pair": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp]),
    "d3f_radius_neighbors_workspace_bytes": (_sz, [_i, _i, _i]),
    "d3f_batch_radius_neighbors": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _f, _vp, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "d3f_neighbor_grid_bytes": (_sz, [_i, _i]),
    "d3f_neighbor_grid_order_offset": (_sz, [_i, _i]),
    "d3f_neighbor_grid_build": (_i, [_vp, _i, _vp, _i, _f, _vp, _sz, _vp]),
    "d3f_neighbor_grid_search": (_i, [_vp, _sz, _i, _vp, _i, _vp, _i, _f, _i, _vp, _i, _i, _i, _i, _i, _f, _i, _vp, _vp]),
    "d3f_neighbor_grid_search_ordered": (_i, [_vp, _sz, _i, _vp, _i, _vp, _i, _f, _vp, _sz, _vp, _i, _i, _i, _i, _i, _f, _i, _vp, _vp]),
    "d3f_neighbor_grid_inv_offset": (_sz, [_i, _i]),
    "d3f_neighbor_grid_xyz_offset": (_sz, [_i, _i]),
    "d3f_row_positive": (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _vp]),
    "d3f_kpconv_aggregate": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _vp, _i, _i, _vp, _vp, _i, _f, _i, _i, _vp, _vp, _vp, _vp,
                                  _vp, _i, _vp]),
    "d3f_kpconv_fused_c1": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _vp, _i, _vp, _i, _f, _i, _i, _vp, _i, _vp, _vp, _vp, _i, _i,
                                 _f, _vp, _i, _vp, _vp, _vp, _i, _vp]),
    "d3f_kpconv_fused32": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _vp, _i, _vp, _vp, _i, _f, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _f, _vp,
                                _i, _vp, _vp, _vp, _i, _vp]),
    "d3f_kpconv_fused_supported": (_i, [_i, _i, _i, _i, _i]),
    "d3f_kpconv_pack_weights": (_i, [_vp, _i, _i, _vp, _vp]),
    "d3f_kpconv_fused": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _vp, _i, _i, _vp, _vp, _i, _f, _i, _i, _vp, _i, _vp, _vp, _vp, _i, _i,
                              _f, _vp, _i, _vp, _vp, _vp, _i, _vp]),
    "d3f_kpconv_fused32_mfma": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _vp, _i, _vp, _vp, _i, _f, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _f, _vp,
                                _i, _vp, _vp, _vp, _vp]),
    "d3f_kpconv_fused32_x3": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _vp, _i, _vp, _vp, _i, _f, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _f, _vp,
                                   _i, _vp, _vp, _vp, _i, _vp]),
    "d3f_kpconv_packed_x3_bytes": (_sz, [_i, _i]),
    "d3f_kpconv_pack_weights_x3": (_i, [_vp, _i, _i, _vp, _vp]),
    "d3f_kpconv_fused_x3": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _vp, _i, _i, _vp, _vp, _i, _f, _i, _i, _vp, _i, _vp, _vp, _vp, _i, _i,
                                 _f, _vp, _i, _vp, _vp, _vp, _i, _vp]),
    "d3f_gemm_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "d3f_gemm_bf16_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "d3f_gemm_f32": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _f, _vp, _sz, _vp, _i, _vp]),
    "d3f_gemm_upsample_cat_f32": (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _i, _i, _vp, _i, _vp, _i, _i, _i, _vp, _vp, _i, _f, _vp, _sz,
                                       _vp, _vp, _i, _vp]),
    "d3f_ind_max_pool": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    "d3f_closest_pool_cat": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _vp, _i, _i, _vp, _i, _vp, _vp, _vp]),
    "d3f_affine_act": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _i, _i, _f, _vp, _i, _vp, _vp]),
    "d3f_pack_descriptors": (_i, [_vp, _vp, _i, _i, _vp, _i, _vp, _i, _vp, _vp]),
    "d3f_pack_descriptors_to": (_i, [_vp, _vp, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "d3f_feature_nn_workspace_bytes": (_sz, [_i]),
    "d3f_feature_nn": (_i, [_vp, _i, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "d3f_mutual_matches_workspace_bytes": (_sz, [_i]),
    "d3f_mutual_matches": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "d3f_ransac_hypotheses": (_i, [_vp, _i, _vp, _i, _vp, _i, _f, _f, C.c_uint64, C.c_uint64, _i, _vp, _vp, _vp]),
    "d3f_ransac_draw": (_i, [C.c_uint64, C.c_uint64, _i, _i]),
    "d3f_neighbor_grid_score": (_i, [_vp, _sz, _i, _vp, _i, _vp, _i, _f, _vp, _vp, _vp, _vp]),
    "d3f_gemm_pack_bf16": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "d3f_gemm_pack_f32t": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "d3f_gemm_f32t": (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _i, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _f, _vp, _sz,
                           _vp, _vp, _i, _vp]),
    "d3f_gemm_x3_packed_bytes": (_sz, [_i, _i]),
    "d3f_gemm_pack_x3": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "d3f_gemm_x3_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "d3f_gemm_x3_plan": (_i, [_i, _i, _i, _i, _vp, _vp, _vp]),
    "d3f_gemm_x3": (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _i, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _f, _vp, _sz,
                         _vp, _vp, _i, _vp]),
    "d3f_gemm_bf16": (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _i, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _f, _vp, _sz,
                           _vp, _vp, _i, _i, _i, _vp]),
    "d3f_decode_xyz_records": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "d3f_detect_head": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _vp, _vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp]),
}

_lib = None


class D3FeatLibraryError(RuntimeError):
    pass


def load():
    """Load libd3feat_amd.so; raises D3FeatLibraryError when it is missing (no fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise D3FeatLibraryError(
            "d3feat_amd: %s not found -- the HIP extension is the only compute path. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C d3feat_amd/csrc`." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != D3F_OK:
        raise D3FeatLibraryError("d3feat_amd.%s failed: %s (code %d)" % (what, ERRORS.get(rc, "unknown"), rc))
