"""TEST INFRASTRUCTURE ONLY (oracle).  ctypes loaders for

  * oracle/libd3f_oracle.so          -- this repo's plain-C restatement (d3f_oracle.c)   -> `COracle`
  * oracle/_ref/libd3f_ref.so        -- the reference's tf_custom_ops cores              -> `RefLib`
  * oracle/_ref/libd3f_ref_wrap.so   -- the reference's cpp_wrappers core                -> `RefWrapLib`

All functions take / return numpy arrays.  Never imported by d3feat_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def build(ref=True):
    """(Re)build the C restatement and, when /root/reference exists, oracle/_ref."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])
    if ref:
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


class COracle:
    """Plain-C restatement (oracle/d3f_oracle.c)."""

    def __init__(self):
        path = os.path.join(_HERE, "libd3f_oracle.so")
        if not os.path.exists(path):
            build(ref=False)
        self.lib = C.CDLL(path)
        self.lib.orc_grid_subsampling.restype = C.c_int
        self.lib.orc_grid_subsampling.argtypes = [_fp, C.c_int, _fp, C.c_int, _ip, C.c_int, C.c_float, _fp, _fp, _ip]
        self.lib.orc_batch_grid_subsampling.restype = C.c_int
        self.lib.orc_batch_grid_subsampling.argtypes = [_fp, C.c_int, _ip, C.c_int, C.c_float, _fp, _ip]
        self.lib.orc_batch_neighbors.restype = C.c_int
        self.lib.orc_batch_neighbors.argtypes = [_fp, C.c_int, _fp, C.c_int, _ip, _ip, C.c_int, C.c_float, C.c_int,
                                                 C.POINTER(_ip)]
        self.lib.orc_free.argtypes = [C.c_void_p]

    def grid_subsampling(self, points, dl, features=None, classes=None):
        p = _f32(points)
        n = p.shape[0]
        fdim = ldim = 0
        f = c = None
        if features is not None:
            f = _f32(features).reshape(n, -1)
            fdim = f.shape[1]
        if classes is not None:
            c = _i32(classes).reshape(n, -1)
            ldim = c.shape[1]
        op = np.empty((max(n, 1), 3), np.float32)
        of = np.empty((max(n, 1), max(fdim, 1)), np.float32)
        oc = np.empty((max(n, 1), max(ldim, 1)), np.int32)
        m = self.lib.orc_grid_subsampling(p.ctypes.data_as(_fp), n,
                                          f.ctypes.data_as(_fp) if f is not None else None, fdim,
                                          c.ctypes.data_as(_ip) if c is not None else None, ldim,
                                          C.c_float(dl), op.ctypes.data_as(_fp), of.ctypes.data_as(_fp),
                                          oc.ctypes.data_as(_ip))
        res = [op[:m].copy()]
        if features is not None:
            res.append(of.reshape(-1)[: m * fdim].reshape(m, fdim).copy())
        if classes is not None:
            res.append(oc.reshape(-1)[: m * ldim].reshape(m, ldim).copy())
        return res[0] if len(res) == 1 else tuple(res)

    def batch_grid_subsampling(self, points, lens, dl):
        p = _f32(points)
        l = _i32(lens)
        op = np.empty((max(p.shape[0], 1), 3), np.float32)
        ob = np.empty(l.shape[0], np.int32)
        m = self.lib.orc_batch_grid_subsampling(p.ctypes.data_as(_fp), p.shape[0], l.ctypes.data_as(_ip), l.shape[0],
                                                C.c_float(dl), op.ctypes.data_as(_fp), ob.ctypes.data_as(_ip))
        return op[:m].copy(), ob

    def batch_neighbors(self, queries, supports, q_lens, s_lens, radius, grid=True):
        q, s, ql, sl = _f32(queries), _f32(supports), _i32(q_lens), _i32(s_lens)
        out = _ip()
        k = self.lib.orc_batch_neighbors(q.ctypes.data_as(_fp), q.shape[0], s.ctypes.data_as(_fp), s.shape[0],
                                         ql.ctypes.data_as(_ip), sl.ctypes.data_as(_ip), ql.shape[0],
                                         C.c_float(radius), int(bool(grid)), C.byref(out))
        res = np.ctypeslib.as_array(out, shape=(q.shape[0] * k + 1,))[: q.shape[0] * k].reshape(q.shape[0], k).copy()
        self.lib.orc_free(out)
        return res


def ref_available():
    return os.path.exists(os.path.join(_HERE, "_ref", "libd3f_ref.so"))


class RefLib:
    """The reference's own tf_custom_ops C++ (compiled in place, oracle/ref_shim.cpp)."""

    def __init__(self):
        self.lib = C.CDLL(os.path.join(_HERE, "_ref", "libd3f_ref.so"))
        for name in ("ref_batch_nanoflann_neighbors", "ref_batch_ordered_neighbors"):
            fn = getattr(self.lib, name)
            fn.restype = C.c_int
            fn.argtypes = [_fp, C.c_int, _fp, C.c_int, _ip, _ip, C.c_int, C.c_float, C.POINTER(_ip)]
        self.lib.ref_ordered_neighbors.restype = C.c_int
        self.lib.ref_ordered_neighbors.argtypes = [_fp, C.c_int, _fp, C.c_int, C.c_float, C.POINTER(_ip)]
        self.lib.ref_grid_subsampling.restype = C.c_int
        self.lib.ref_grid_subsampling.argtypes = [_fp, C.c_int, C.c_float, C.POINTER(_fp)]
        self.lib.ref_batch_grid_subsampling.restype = C.c_int
        self.lib.ref_batch_grid_subsampling.argtypes = [_fp, C.c_int, _ip, C.c_int, C.c_float, C.POINTER(_fp), _ip]
        self.lib.ref_free.argtypes = [C.c_void_p]

    def _nbr(self, fn, q, s, ql, sl, radius):
        q, s, ql, sl = _f32(q), _f32(s), _i32(ql), _i32(sl)
        out = _ip()
        k = fn(q.ctypes.data_as(_fp), q.shape[0], s.ctypes.data_as(_fp), s.shape[0], ql.ctypes.data_as(_ip),
               sl.ctypes.data_as(_ip), ql.shape[0], C.c_float(radius), C.byref(out))
        res = np.ctypeslib.as_array(out, shape=(q.shape[0] * k + 1,))[: q.shape[0] * k].reshape(q.shape[0], k).copy()
        self.lib.ref_free(out)
        return res

    def batch_nanoflann_neighbors(self, q, s, ql, sl, radius):
        return self._nbr(self.lib.ref_batch_nanoflann_neighbors, q, s, ql, sl, radius)

    def batch_ordered_neighbors(self, q, s, ql, sl, radius):
        return self._nbr(self.lib.ref_batch_ordered_neighbors, q, s, ql, sl, radius)

    def ordered_neighbors(self, q, s, radius):
        q, s = _f32(q), _f32(s)
        out = _ip()
        k = self.lib.ref_ordered_neighbors(q.ctypes.data_as(_fp), q.shape[0], s.ctypes.data_as(_fp), s.shape[0],
                                           C.c_float(radius), C.byref(out))
        res = np.ctypeslib.as_array(out, shape=(q.shape[0] * k + 1,))[: q.shape[0] * k].reshape(q.shape[0], k).copy()
        self.lib.ref_free(out)
        return res

    def grid_subsampling(self, points, dl):
        p = _f32(points)
        out = _fp()
        m = self.lib.ref_grid_subsampling(p.ctypes.data_as(_fp), p.shape[0], C.c_float(dl), C.byref(out))
        res = np.ctypeslib.as_array(out, shape=(m * 3 + 1,))[: m * 3].reshape(m, 3).copy()
        self.lib.ref_free(out)
        return res

    def batch_grid_subsampling(self, points, lens, dl):
        p, l = _f32(points), _i32(lens)
        out = _fp()
        ob = np.empty(l.shape[0], np.int32)
        m = self.lib.ref_batch_grid_subsampling(p.ctypes.data_as(_fp), p.shape[0], l.ctypes.data_as(_ip), l.shape[0],
                                                C.c_float(dl), C.byref(out), ob.ctypes.data_as(_ip))
        res = np.ctypeslib.as_array(out, shape=(m * 3 + 1,))[: m * 3].reshape(m, 3).copy()
        self.lib.ref_free(out)
        return res, ob


class RefWrapLib:
    """The reference's cpp_wrappers grid_subsampling core (features / classes variant)."""

    def __init__(self):
        self.lib = C.CDLL(os.path.join(_HERE, "_ref", "libd3f_ref_wrap.so"))
        self.lib.refw_grid_subsampling.restype = C.c_int
        self.lib.refw_grid_subsampling.argtypes = [_fp, C.c_int, _fp, C.c_int, _ip, C.c_int, C.c_float,
                                                   C.POINTER(_fp), C.POINTER(_fp), C.POINTER(_ip)]
        self.lib.refw_free.argtypes = [C.c_void_p]

    def grid_subsampling(self, points, dl, features=None, classes=None):
        p = _f32(points)
        n = p.shape[0]
        fdim = ldim = 0
        f = c = None
        if features is not None:
            f = _f32(features).reshape(n, -1)
            fdim = f.shape[1]
        if classes is not None:
            c = _i32(classes).reshape(n, -1)
            ldim = c.shape[1]
        op, of, oc = _fp(), _fp(), _ip()
        m = self.lib.refw_grid_subsampling(p.ctypes.data_as(_fp), n, f.ctypes.data_as(_fp) if f is not None else None,
                                           fdim, c.ctypes.data_as(_ip) if c is not None else None, ldim,
                                           C.c_float(dl), C.byref(op), C.byref(of), C.byref(oc))
        res = [np.ctypeslib.as_array(op, shape=(m * 3 + 1,))[: m * 3].reshape(m, 3).copy()]
        if features is not None:
            res.append(np.ctypeslib.as_array(of, shape=(m * fdim + 1,))[: m * fdim].reshape(m, fdim).copy())
        if classes is not None:
            res.append(np.ctypeslib.as_array(oc, shape=(m * ldim + 1,))[: m * ldim].reshape(m, ldim).copy())
        for ptr in (op, of, oc):
            self.lib.refw_free(ptr)
        return res[0] if len(res) == 1 else tuple(res)
