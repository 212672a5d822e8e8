"""torch-tensor front end of the C ABI (include/d3feat_amd.h).

PyTorch is used for three things only: device allocations (torch.empty on a CUDA/HIP device), the current
HIP stream, and torch.distributed.  Every computation below is a call into libd3feat_amd.so; nothing here
computes with torch ops and nothing falls back to the CPU.
"""
import ctypes

import os

import numpy as np
import torch

from . import _lib

_WS = {}

# Optional per-call HIP-event timing (bench.py): set PROFILE to a list to collect (name, info, start, end) records
# for the kpconv_aggregate / gemm_f32 launches issued on the current stream.
PROFILE = None


class _timed:
    def __init__(self, name, info, device):
        self.on = PROFILE is not None
        self.name, self.info, self.device = name, info, device

    def __enter__(self):
        if self.on:
            self.start = torch.cuda.Event(enable_timing=True)
            self.end = torch.cuda.Event(enable_timing=True)
            self.start.record(torch.cuda.current_stream(self.device))
        return self

    def __exit__(self, *exc):
        if self.on:
            self.end.record(torch.cuda.current_stream(self.device))
            PROFILE.append((self.name, self.info, self.start, self.end))
        return False


def _stream(device):
    return torch.cuda.current_stream(device).cuda_stream


def trace_marker(ident, device):
    """One named one-thread kernel on the current stream: cuts a rocprofv3 kernel trace at the ends of a timed region."""
    _lib.check(_lib.load().d3f_trace_marker(int(ident), _stream(device)), "trace_marker")


# ---- device-resident sizes ------------------------------------------------------------------------------------------
# A tensor whose row count is only an upper bound carries the real count as the attribute `n_dev` (int32[1] on the
# device); every op forwards it to the kernels (the `*_dev` arguments of include/d3feat_amd.h) and tags its outputs.
def _order(t):
    """Spatially coherent visiting order attached to a point / index tensor (attribute `order`, int32), or None."""
    o = getattr(t, "order", None) if t is not None else None
    return o.data_ptr() if o is not None else None


def _nd(t):
    d = getattr(t, "n_dev", None) if t is not None else None
    return d.data_ptr() if d is not None else None


def _tag(out, src):
    d = getattr(src, "n_dev", None)
    if d is not None:
        out.n_dev = d
        out.n_hint = getattr(src, "n_hint", 0)   # expected row count (host int): plans launches, never bounds them
    return out


class private_workspace:
    """Scratch buffers allocated inside this context are not shared with (or resized by) later eager calls: used while
    a HIP graph is being captured, so that the graph keeps its own scratch alive."""

    def __enter__(self):
        global _WS
        self.prev, _WS = _WS, {}
        return self

    def __exit__(self, *exc):
        global _WS
        self.kept, _WS = _WS, self.prev
        return False


def workspace(nbytes, device):
    """Stream-ordered scratch: one growable byte buffer per (device, stream)."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), _stream(device))
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf


def _req(t, dtype, name, ndim=None):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor on a GPU (got %s)" % (name, type(t)))
    if not t.is_cuda:
        raise _lib.D3FeatLibraryError("%s is on %s: d3feat_amd has no CPU path" % (name, t.device))
    if t.dtype != dtype:
        raise TypeError("%s must be %s (got %s)" % (name, dtype, t.dtype))
    if ndim is not None and t.dim() != ndim:
        raise ValueError("%s must be %d-D (got shape %s)" % (name, ndim, tuple(t.shape)))
    return t


def _rows(t, name):
    """2-D tensor whose rows are contiguous -> (tensor, leading dimension in elements)."""
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise ValueError("%s must be 2-D with contiguous rows (shape %s strides %s)" % (name, tuple(t.shape), t.stride()))
    ld = t.stride(0) if t.shape[0] > 1 else t.shape[1]
    if ld < t.shape[1]:
        raise ValueError("%s has overlapping rows" % name)
    return t, int(max(ld, 1))


def as_lens(lens, device):
    """Batch lengths as a device int32 tensor.  When the values are known on the host they ride along as the
    attribute `host_lens` (saves a device read-back wherever host logic needs them)."""
    if isinstance(lens, torch.Tensor):
        out = lens.to(device=device, dtype=torch.int32).contiguous()
        if out is not lens and hasattr(lens, "host_lens"):
            out.host_lens = lens.host_lens
        return out
    host = np.asarray(lens, dtype=np.int32).reshape(-1)
    out = torch.as_tensor(host, device=device)
    out.host_lens = [int(x) for x in host]
    return out


def host_lens(lens):
    """Python list of the batch lengths (reads the device tensor back only when no host copy rides along)."""
    if isinstance(lens, torch.Tensor):
        h = getattr(lens, "host_lens", None)
        return list(h) if h is not None else [int(x) for x in lens.tolist()]
    return [int(x) for x in np.asarray(lens).reshape(-1)]


def _raise_flags(flags, what):
    if flags:
        msgs = []
        if flags & _lib.ST_EMPTY_ELEMENT:
            msgs.append("a batch element is empty")
        if flags & _lib.ST_NEG_CELL:
            msgs.append("negative voxel index (origin above a point after fp32 rounding)")
        if flags & _lib.ST_KEY_RANGE:
            msgs.append("voxel key exceeds 2^56")
        if flags & _lib.ST_HIT_OVERFLOW:
            msgs.append("a query has more than %d in-radius supports" % _lib.NEIGHBOR_CAP)
        if flags & _lib.ST_OUT_OVERFLOW:
            msgs.append("more output rows than the capacity of the output buffer")
        if flags & _lib.ST_KEY_WIDTH:
            msgs.append("(element, voxel key) wider than the 32-bit sort key of the capacity-mode subsampling")
        raise _lib.D3FeatLibraryError("d3feat_amd.%s: %s" % (what, "; ".join(msgs)))
    return 0


def check_status(status, what):
    """status: int32[2] device tensor written by a kernel; raises on any D3F_ST_* flag. Synchronises."""
    st = status.tolist()
    _raise_flags(st[1], what)
    return st[0]


# ----------------------------------------------------------------------------------------------------------
def batch_grid_subsample(points, lens, dl, features=None, classes=None):
    """-> (sub_points f32[M,3], sub_lens i32[B] (device), sub_features | None, sub_classes | None).
    One host synchronisation (M is data dependent)."""
    lib = _lib.load()
    points = _req(points, torch.float32, "points", 2).contiguous()
    dev = points.device
    N = points.shape[0]
    lens_t = as_lens(lens, dev)
    B = lens_t.numel()
    fdim = ldim = 0
    if features is not None:
        features = _req(features, torch.float32, "features", 2).contiguous()
        fdim = features.shape[1]
    if classes is not None:
        classes = _req(classes, torch.int32, "classes", 2).contiguous()
        ldim = classes.shape[1]
    sub_p = torch.empty((max(N, 1), 3), dtype=torch.float32, device=dev)
    sub_f = torch.empty((max(N, 1), fdim), dtype=torch.float32, device=dev) if fdim else None
    sub_c = torch.empty((max(N, 1), ldim), dtype=torch.int32, device=dev) if ldim else None
    sub_l = torch.empty((B,), dtype=torch.int32, device=dev)
    status = (ctypes.c_int * (B + 2))()
    nbytes = lib.d3f_grid_subsample_workspace_bytes(N, B, fdim, ldim)
    ws = workspace(nbytes, dev)
    with _timed("grid_subsample", dict(N=N, M=None), dev) as tm:
        rc = lib.d3f_batch_grid_subsample(points.data_ptr(), N, lens_t.data_ptr(), B, float(dl),
                                          features.data_ptr() if fdim else None, fdim,
                                          classes.data_ptr() if ldim else None, ldim,
                                          sub_p.data_ptr(), sub_f.data_ptr() if fdim else None,
                                          sub_c.data_ptr() if ldim else None, sub_l.data_ptr(),
                                          ctypes.addressof(status), ws.data_ptr(), ws.numel(), _stream(dev))
        tm.info["M"] = int(status[0])
    _lib.check(rc, "batch_grid_subsample")
    M = _raise_flags(status[1], "batch_grid_subsample") or status[0]
    sub_l.host_lens = [int(status[2 + b]) for b in range(B)]
    return sub_p[:M], sub_l, (sub_f[:M] if fdim else None), (sub_c[:M] if ldim else None)


def batch_grid_subsample_async(points, lens, dl, m_cap, status=None, m_hint=0, elem_cap=0, elem_points=0, clouds=None, n_cap=0):
    """Capacity mode of batch_grid_subsample (points only): no host synchronisation.
    clouds (device int64[B]: the ADDRESS of every cloud's own f32[len, 3] array) with points=None and n_cap = bound of sum(lens):
    the clouds are read in place, nothing is stacked (d3f_batch_grid_subsample_async_inplace).
    points f32[N_cap,3] (sum(lens) rows valid) -> (sub_points f32[m_cap,3] tagged with n_dev, sub_lens i32[B] device,
    status i32[2] device = [M, flags]).  elem_cap: capacity of ONE cloud of the stack (0: m_cap); a cloud above it raises
    the overflow flag like a stack above m_cap does.  elem_points: capacity of one cloud in POINTS (0: all rows of `points`);
    at most 16384 selects the one-workgroup-per-cloud form (the coarse pyramid levels: 2 launches instead of ~25)."""
    lib = _lib.load()
    if clouds is not None:
        assert points is None and clouds.dtype == torch.int64 and clouds.is_contiguous() and int(n_cap) > 0
        dev, N = clouds.device, int(n_cap)
    else:
        points = _req(points, torch.float32, "points", 2).contiguous()
        dev = points.device
        N = points.shape[0]
    lens_t = as_lens(lens, dev)
    B = lens_t.numel()
    sub_p = torch.empty((int(m_cap), 3), dtype=torch.float32, device=dev)
    sub_l = torch.empty((B,), dtype=torch.int32, device=dev)
    if status is None:
        status = torch.empty((2,), dtype=torch.int32, device=dev)
    nbytes = lib.d3f_grid_subsample_workspace_bytes(N, B, 0, 0)
    ws = workspace(nbytes, dev)
    with _timed("grid_subsample", dict(N=N, M=int(m_cap)), dev):
        if clouds is not None:
            assert clouds.numel() == B
            rc = lib.d3f_batch_grid_subsample_async_inplace(clouds.data_ptr(), N, lens_t.data_ptr(), B, float(dl), sub_p.data_ptr(),
                                                            int(m_cap), int(elem_cap), sub_l.data_ptr(), status.data_ptr(),
                                                            ws.data_ptr(), ws.numel(), _stream(dev))
        else:
            rc = lib.d3f_batch_grid_subsample_async(points.data_ptr(), N, lens_t.data_ptr(), B, float(dl), sub_p.data_ptr(),
                                                    int(m_cap), int(elem_cap), int(elem_points), sub_l.data_ptr(), status.data_ptr(),
                                                    ws.data_ptr(), ws.numel(), _stream(dev))
    _lib.check(rc, "batch_grid_subsample_async")
    sub_p.n_dev = status[0:1]
    sub_p.n_hint = int(m_hint)
    return sub_p, sub_l, status


def pack_status(dst, blocks, clear=None):
    """dst i32[>= sum of the blocks' sizes] <- the (up to four) device int32 blocks one after the other, then the FLAG column of
    `clear` (a device int32 [k, 2] tensor of [kmax-or-size | flags] pairs, may be one of the blocks) zeroed -- the size words stay
    readable after the replay: one launch (d3f_pack_status) instead of a copy node per block and a fill."""
    lib = _lib.load()
    # contiguity is checked on the CALLER's tensors: the launch may be captured in a HIP graph with these addresses, and a silent
    # copy (reshape of a strided view) would hand the graph a temporary that is freed when this call returns
    assert 1 <= len(blocks) <= 4 and all(b.dtype == torch.int32 and b.is_contiguous() for b in blocks) and dst.dtype == torch.int32
    assert dst.is_contiguous()
    blocks = [b.view(-1) for b in blocks]
    assert dst.numel() >= sum(b.numel() for b in blocks)
    args = []
    for i in range(4):
        args += [blocks[i].data_ptr(), blocks[i].numel()] if i < len(blocks) else [None, 0]
    if clear is not None:
        assert clear.dtype == torch.int32 and clear.is_contiguous() and clear.dim() == 2 and clear.shape[1] == 2
    _lib.check(lib.d3f_pack_status(dst.data_ptr(), *args, clear.data_ptr() if clear is not None else None,
                                   clear.shape[0] if clear is not None else 0, 1, 2, _stream(dst.device)), "pack_status")
    return dst


def stack_self_pair(pts, lens=None):
    """np.concatenate([c, c]) for every cloud c of a stack whose row counts live on the device:
    pts f32[cap,3] holding B clouds (lens i32[B] on the device; default: one cloud of pts.n_dev rows)
    -> (out f32[2*cap,3] = [c_0; c_0; c_1; c_1; ...] tagged with n_dev, lens_out i32[2B])."""
    lib = _lib.load()
    pts = _req(pts, torch.float32, "pts", 2).contiguous()
    dev = pts.device
    cap = pts.shape[0]
    if lens is None:
        lens = getattr(pts, "n_dev", None)
        if lens is None:
            lens = torch.full((1,), cap, dtype=torch.int32, device=dev)
    B = lens.numel()
    out = torch.empty((2 * cap, 3), dtype=torch.float32, device=dev)
    lens_out = torch.empty((2 * B,), dtype=torch.int32, device=dev)
    total = torch.empty((1,), dtype=torch.int32, device=dev)
    rc = lib.d3f_stack_self_pair(pts.data_ptr(), cap, lens.data_ptr(), B, out.data_ptr(), lens_out.data_ptr(),
                                 total.data_ptr(), _stream(dev))
    _lib.check(rc, "stack_self_pair")
    out.n_dev = total
    out.n_hint = 2 * int(getattr(pts, "n_hint", 0) or 0)
    return out, lens_out


class NeighborGrid:
    """Cell grid over a stacked support cloud (see d3f_neighbor_grid_build); owns its device memory."""

    def __init__(self, supports, s_lens, radius):
        lib = _lib.load()
        self.supports = _tag(_req(supports, torch.float32, "supports", 2).contiguous(), supports)
        dev = self.supports.device
        self.Ns = self.supports.shape[0]
        self.s_lens = as_lens(s_lens, dev)
        self.B = self.s_lens.numel()
        self.radius = float(radius)
        self.nbytes = lib.d3f_neighbor_grid_bytes(self.Ns, self.B)
        self.mem = torch.empty((self.nbytes,), dtype=torch.uint8, device=dev)
        with _timed("nb_grid_build", dict(Ns=self.Ns), dev):
            rc = lib.d3f_neighbor_grid_build(self.supports.data_ptr(), self.Ns, self.s_lens.data_ptr(), self.B, self.radius,
                                             self.mem.data_ptr(), self.nbytes, _stream(dev))
        _lib.check(rc, "neighbor_grid_build")
        # support indices sorted by cell (a view into the grid object): a spatially coherent visiting order
        off = lib.d3f_neighbor_grid_order_offset(self.Ns, self.B)
        self.order = self.mem[off: off + 4 * max(self.Ns, 1)].view(torch.int32)
        # the level in the INTERNAL (cell-sorted) numbering: position of every support in cell order, and the supports in that order
        off = lib.d3f_neighbor_grid_inv_offset(self.Ns, self.B)
        self.inv = self.mem[off: off + 4 * max(self.Ns, 1)].view(torch.int32)
        off = lib.d3f_neighbor_grid_xyz_offset(self.Ns, self.B)
        self.xyz = self.mem[off: off + 12 * max(self.Ns, 1)].view(torch.float32).view(-1, 3)
        for a in ("n_dev", "n_hint"):
            if getattr(self.supports, a, None) is not None:
                setattr(self.xyz, a, getattr(self.supports, a))

    def search(self, queries, q_lens, width, ld=None, pad_value=None, cap=192, first_only=False, out=None, status=None,
               reset_status=True, want_kmax=True, nn_hint=0.0, query_grid=None, internal=False):
        """-> (out i32[Nq, ld], status i32[2] device tensor); no synchronisation.  reset_status=False: the caller has
        zeroed `status` (saves one launch per search).  want_kmax=False: status[0] (largest neighbour count) is not maintained
        (callers that allocate a fixed number of columns do not need it; see D3F_NB_NO_KMAX).  nn_hint (first_only): the distance
        within which the caller expects the nearest support -- a speed hint only, see include/d3feat_amd.h.
        query_grid: a NeighborGrid built over `queries` themselves (or this grid when the queries are its supports) -- its cell order
        becomes the visiting order (d3f_neighbor_grid_search_ordered); results do not depend on it.
        internal=True (needs query_grid): the INTERNAL numbering -- row j belongs to the j-th query of query_grid's cell order and
        the entries are positions in this grid's cell order (self.inv) instead of indices; pad_value is written as is."""
        lib = _lib.load()
        queries = _req(queries, torch.float32, "queries", 2).contiguous()
        dev = queries.device
        Nq = queries.shape[0]
        ql = as_lens(q_lens, dev)
        if ql.numel() != self.B:
            raise ValueError("q_batches and s_batches must have the same number of elements")
        ld = int(ld if ld is not None else width)
        if out is None:
            out = torch.empty((Nq, ld), dtype=torch.int32, device=dev)
        if status is None:
            status = torch.empty((2,), dtype=torch.int32, device=dev)
        same = 1 if (queries.data_ptr() == self.supports.data_ptr() and Nq == self.Ns) else 0
        if pad_value is None:
            # BatchOrderedNeighbors pads with the number of supports: known only on the device in capacity mode
            pad_value = _lib.PAD_NUM_SUPPORTS if getattr(self.supports, "n_dev", None) is not None else self.Ns
        if query_grid is not None:
            if query_grid.Ns != Nq or query_grid.B != self.B or query_grid.supports.data_ptr() != queries.data_ptr():
                raise ValueError("query_grid was not built over these queries")
            with _timed("nb_search", dict(Nq=Nq, Ns=self.Ns, width=int(width), first_only=int(bool(first_only))), dev):
                rc = lib.d3f_neighbor_grid_search_ordered(self.mem.data_ptr(), self.nbytes, self.Ns, queries.data_ptr(), Nq, ql.data_ptr(),
                                                          self.B, self.radius, query_grid.mem.data_ptr(), query_grid.nbytes,
                                                          out.data_ptr(), ld, int(width), int(pad_value), int(cap),
                                                          1 if first_only else 0, float(nn_hint),
                                                          (1 if reset_status else 0) | (0 if want_kmax else 2) | (4 if internal else 0),
                                                          status.data_ptr(), _stream(dev))
            _lib.check(rc, "neighbor_grid_search_ordered")
            o = getattr(queries, "order", None)
            if o is not None and not internal:
                out.order = o
            return _tag(out, queries), status
        if internal:
            raise ValueError("internal numbering needs the queries' grid")
        with _timed("nb_search", dict(Nq=Nq, Ns=self.Ns, width=int(width), first_only=int(bool(first_only))), dev):
            rc = lib.d3f_neighbor_grid_search(self.mem.data_ptr(), self.nbytes, self.Ns, queries.data_ptr(), Nq,
                                              ql.data_ptr(), self.B, self.radius, same, out.data_ptr(), ld, int(width),
                                              int(pad_value), int(cap),
                                              1 if first_only else 0, float(nn_hint),
                                              (1 if reset_status else 0) | (0 if want_kmax else 2),
                                              status.data_ptr(),
                                              _stream(dev))
        _lib.check(rc, "neighbor_grid_search")
        o = getattr(queries, "order", None)
        if o is not None:
            out.order = o         # rows of the index matrix = the queries: same visiting order
        return _tag(out, queries), status


def batch_radius_neighbors(queries, supports, q_lens, s_lens, radius, width, ld=None, out=None, pad_value=None, cap=192,
                           first_only=False):
    """Build + search; -> (out i32[Nq, ld] with `width` valid columns, status i32[2] device tensor).
    No synchronisation: status[0] = Kmax, status[1] = flags (see check_status)."""
    grid = NeighborGrid(supports, s_lens, radius)
    return grid.search(queries, q_lens, width, ld=ld, pad_value=pad_value, cap=cap, first_only=first_only, out=out)


class StackGroups:
    """Stands in for the `in_batches` matrix (datasets/common.py:453-496) on the fast path, where the head kernel works from
    the stack lengths: `group` = clouds per reference stack inside a batched stack (see d3f_detect_head)."""

    def __init__(self, group=0):
        self.group = int(group)


_KP_HOST = {}


def _kp_host(K_points):
    """Kernel points as a contiguous host float32 array (they ride in the kernel arguments).  The model keeps them on the
    host; a caller-supplied device tensor is read back once per (storage, version), not once per call."""
    if isinstance(K_points, torch.Tensor):
        key = (K_points.data_ptr(), K_points._version, tuple(K_points.shape))
        hit = _KP_HOST.get(key)
        if hit is None:
            if len(_KP_HOST) > 256:
                _KP_HOST.clear()
            # the entry keeps the tensor alive: its address cannot be handed to another tensor while the key is in use
            hit = _KP_HOST[key] = (K_points, np.ascontiguousarray(K_points.detach().cpu().numpy(), dtype=np.float32))
        return hit[1]
    return np.ascontiguousarray(K_points, dtype=np.float32)


_INFLUENCE = {"constant": 0, "linear": 1, "gaussian": 2}
_AGGREGATION = {"sum": 0, "closest": 1}


# ---- bf16-operand contractions (BASELINE configs[4]) -------------------------------------------------------------------------
# Off by default: the fp32 path is the parity path.  `with bf16_contraction():` routes every contraction issued inside it (the
# engine wraps its capture in it when built with bf16=True) through d3f_gemm_bf16.
BF16_CONTRACTION = False


BF16_FEATURES = False     # bf16 feature STORAGE (needs BF16_CONTRACTION): activations live in HBM as torch.bfloat16
F32_OUTPUT = False        # inside bf16 feature storage: the next contractions write float32 (the descriptor head's input)


class bf16_contraction:
    """BASELINE configs[4].  on: every unary / unfused KPConv contraction multiplies bf16 operands (fp32 accumulate).
    features=True additionally STORES the activations between the layers as bfloat16 (KPConv gathers, max pooling, the
    decoder gather and every contraction then move half the bytes; arithmetic inside the kernels stays fp32)."""

    def __init__(self, on=True, features=False):
        self.on, self.features = bool(on), bool(on) and bool(features)

    def __enter__(self):
        global BF16_CONTRACTION, BF16_FEATURES
        self.prev, BF16_CONTRACTION = BF16_CONTRACTION, self.on
        self.prevf, BF16_FEATURES = BF16_FEATURES, self.features
        return self

    def __exit__(self, *exc):
        global BF16_CONTRACTION, BF16_FEATURES
        BF16_CONTRACTION, BF16_FEATURES = self.prev, self.prevf
        return False


class f32_output:
    """Contractions inside this context write float32 even under bf16 feature storage (last_unary: descriptors / scores are
    computed from fp32 values)."""

    def __enter__(self):
        global F32_OUTPUT
        self.prev, F32_OUTPUT = F32_OUTPUT, True
        return self

    def __exit__(self, *exc):
        global F32_OUTPUT
        F32_OUTPUT = self.prev
        return False


def _feat(t, name, ndim=None):
    """A feature tensor: float32, or bfloat16 under bf16 feature storage."""
    if isinstance(t, torch.Tensor) and t.dtype == torch.bfloat16:
        return _req(t, torch.bfloat16, name, ndim)
    return _req(t, torch.float32, name, ndim)


def _h(t):
    return 1 if (t is not None and t.dtype == torch.bfloat16) else 0


def _out_dtype():
    return torch.bfloat16 if (BF16_FEATURES and not F32_OUTPUT) else torch.float32


def _packed_on_tensor(W, slot, make):
    """A packed copy of a weight matrix that LIVES AND DIES WITH THE WEIGHT TENSOR: the copy rides on the tensor object that owns
    the storage (the view's base), keyed by the view's offset / shape / stride and the tensor's version.  (A global cache with
    eviction would free copies that captured graphs still read: round 4 found exactly that -- an engine's replays computed with
    another model's packed weights once 512 entries had gone through the cache.)"""
    base = W._base if W._base is not None else W
    store = getattr(base, slot, None)
    if store is None:
        store = {}
        setattr(base, slot, store)
    key = (W.storage_offset(), tuple(W.shape), W.stride(0))
    hit = store.get(key)
    if hit is None:
        hit = store[key] = [base._version, make(None)]
    elif hit[0] != base._version:
        # the weight was updated in place: re-pack INTO THE SAME BUFFER.  A captured graph holds the packed copy's address; a fresh
        # allocation here would free memory that its replays still read (ADVICE r04 / VERDICT r05 item 8a) -- this way the replays
        # simply see the new values (tests/test_gpu_engine.py::test_in_place_weight_update_reaches_a_captured_engine)
        make(hit[1])
        hit[0] = base._version
    return hit[1]


def refresh_packed_weights(tensors):
    """Re-pack (in place, same addresses) every packed copy that rides on one of `tensors` and is older than its tensor.
    The packed copies are made by the EAGER ops; a caller that updates weights in place and then only REPLAYS captured graphs
    calls this once after the update (FragmentEngine.refresh_weights does)."""
    n = 0
    for t in tensors:
        if not isinstance(t, torch.Tensor):
            continue
        base = t._base if t._base is not None else t
        for slot, fn in (("_d3f_bf16t", packed_bf16_weights), ("_d3f_f32t", packed_f32t_weights), ("_d3f_x3", packed_x3_weights)):
            store = getattr(base, slot, None)
            for key, hit in list((store or {}).items()):
                if hit[0] != base._version:
                    off, shape, st0 = key
                    fn(base.as_strided(shape, (st0, 1), off))
                    n += 1
        for slot, fn in (("_d3f_packed", packed_kpconv_weights), ("_d3f_packed_x3", packed_kpconv_weights_x3),
                         ("_d3f_packed_x3m", packed_kpconv_weights_x3m)):
            hit = getattr(t, slot, None)
            if hit is not None and hit[0] != t._version:
                fn(t)
                n += 1
    return n


def packed_bf16_weights(W):
    """W f32[K,N] (contiguous rows) -> the bf16 [N][Kp] copy d3f_gemm_bf16 reads; made once per (weight tensor, view, version)."""
    def make(t):
        lib = _lib.load()
        K, N = W.shape
        Kp = (K + 31) // 32 * 32
        if t is None:
            t = torch.empty((N, Kp), dtype=torch.int16, device=W.device)
        _lib.check(lib.d3f_gemm_pack_bf16(W.data_ptr(), int(W.stride(0)), K, N, t.data_ptr(), _stream(W.device)), "gemm_pack_bf16")
        return t
    return _packed_on_tensor(W, "_d3f_bf16t", make)


# the LDS-DMA form of the fp32 contraction (d3f_gemm_f32t) is the default; D3F_GEMM_DMA=0 selects round 3's register-staged kernel
GEMM_DMA = os.environ.get("D3F_GEMM_DMA", "1") != "0"


def packed_f32t_weights(W):
    """W f32[K,N] (contiguous rows) -> the transposed, K-padded f32 [N][Kp] copy d3f_gemm_f32t reads (LDS-DMA copies 16
    contiguous bytes per lane: it cannot transpose); made once per (weight tensor, view, version) -- at a model's first eager
    use, i.e. the engine's warm-up, never inside a captured graph."""
    def make(t):
        lib = _lib.load()
        K, N = W.shape
        Kp = (K + 31) // 32 * 32
        if t is None:
            t = torch.empty((N, Kp), dtype=torch.float32, device=W.device)
        _lib.check(lib.d3f_gemm_pack_f32t(W.data_ptr(), int(W.stride(0)), K, N, t.data_ptr(), _stream(W.device)), "gemm_pack_f32t")
        return t
    return _packed_on_tensor(W, "_d3f_f32t", make)


# The operand-split form (d3f_gemm_x3: exact 3 x bf16 split of both fp32 operands, six bf16 MFMA products per fp32 product, fp32
# accumulate -- fp32 in, fp32 out, fp32-grade error; csrc/gemm_x3.h) takes every contraction it can address; D3F_GEMM_X3=0 keeps
# them all on the fp32 MFMA kernel.
GEMM_X3 = os.environ.get("D3F_GEMM_X3", "1") != "0"


def packed_x3_weights(W):
    """W f32[K,N] (contiguous rows) -> the pre-split bf16 planes d3f_gemm_x3 stages ([column group][k-tile][plane][32][40]); made
    once per (weight tensor, view, version), like the transposed fp32 copy."""
    def make(t):
        lib = _lib.load()
        K, N = W.shape
        if t is None:
            t = torch.empty((int(lib.d3f_gemm_x3_packed_bytes(K, N)) // 2,), dtype=torch.int16, device=W.device)
        _lib.check(lib.d3f_gemm_pack_x3(W.data_ptr(), int(W.stride(0)), K, N, t.data_ptr(), _stream(W.device)), "gemm_pack_x3")
        return t
    return _packed_on_tensor(W, "_d3f_x3", make)


X3R_MIN_ROWS = 65536     # d3f_gemm_x3 takes its resident-W persistent form from this many (expected) rows on (csrc/gemm_f32.hip)
X3_N32 = os.environ.get("D3F_X3_N32", "1") != "0"      # the 32-column layers on that form too (D3F_X3_N32=0: the fp32 MFMA kernel)


def _x3_ok(C1, C2, N, rows=0):
    """d3f_gemm_x3 can address the call (K and a concatenation's first part multiples of 32) and is the faster kernel for it.  The
    32-column layers (level-0 unary blocks, memory bound) are slower on its tile form than on the fp32 MFMA kernel (70.9 against
    83.6 us at M = 707592, K = 64: r04 x6) but faster on the resident-W persistent form of round 5, which the library takes for
    `rows` >= X3R_MIN_ROWS (70.6 / 105.8 us against 72-82 / 118.6 at M = 707592, 23 / 31 against 28 / 39 at M = 235864: r05 g3)."""
    wide = N > 32 or (X3_N32 and rows >= X3R_MIN_ROWS and os.environ.get("D3F_GEMM_X3R", "1") != "0")
    return GEMM_X3 and wide and (C1 + C2) % 32 == 0 and (C2 == 0 or C1 % 32 == 0)


X3_MAX_ROWS = 65535 * 128      # d3f_gemm_x3's grid holds 65535 row tiles of (at least) 128 rows; beyond: the fp32 kernel


def _f32t_ok(N, ldc, out, residual, ldr, vectors, *operands):
    """Can d3f_gemm_f32t address this call?  (every shape of the network can)"""
    if not GEMM_DMA or N % 4 or ldc % 4 or out.data_ptr() % 16:
        return False
    if residual is not None and (ldr % 4 or residual.data_ptr() % 16):
        return False
    if any(v is not None and v.data_ptr() % 16 for v in vectors):
        return False
    for t, ld, cols in operands:
        if t is not None and (cols % 4 or ld % 4 or t.data_ptr() % 16 or cols < 4):
            return False
    return True


def _gemm_f32t(A, N1, lda, C1, idx, ld_idx, skip, lds, C2, W, out, ldc, M, N, row_scale, col_scale, col_shift, residual, ldr, leaky,
               alpha, m_dev, n1_dev, hint, dev):
    lib = _lib.load()
    if _x3_ok(C1, C2, N, hint if 0 < hint < M else M) and M <= X3_MAX_ROWS:
        Wx = packed_x3_weights(W)
        ws = workspace(lib.d3f_gemm_x3_workspace_bytes(M, N, C1 + C2, hint), dev)
        # (record name only: which form the library takes -- the predicate of csrc/gemm_f32.hip d3f_gemm_x3, mirrored for the
        # per-family tables of bench.py: the resident-W persistent form is a streaming kernel, bound by HBM, not by the matrix pipe)
        ng, rows = (N + 31) // 32, (hint if 0 < hint < M else M)
        resident = (os.environ.get("D3F_GEMM_X3R", "1") != "0" and ng in (1, 2, 4) and rows >= X3R_MIN_ROWS and
                    ((C1 + C2) // 32) * ng * 7680 + 36864 + 1024 <= 160 * 1024)
        with _timed("gemm_x3r" if resident else "gemm_x3", dict(M=M, N=N, K=C1 + C2), dev):
            rc = lib.d3f_gemm_x3(A.data_ptr(), N1, lda, C1, idx.data_ptr() if idx is not None else None, ld_idx,
                                 skip.data_ptr() if skip is not None else None, lds, C2, Wx.data_ptr(), out.data_ptr(), ldc, M, N,
                                 row_scale.data_ptr() if row_scale is not None else None,
                                 col_scale.data_ptr() if col_scale is not None else None,
                                 col_shift.data_ptr() if col_shift is not None else None,
                                 residual.data_ptr() if residual is not None else None, ldr, 1 if leaky else 0, float(alpha),
                                 ws.data_ptr(), ws.numel(), m_dev, n1_dev, hint, _stream(dev))
        _lib.check(rc, "gemm_x3")
        return
    Wt = packed_f32t_weights(W)
    ws = workspace(lib.d3f_gemm_workspace_bytes(M, N, C1 + C2, hint), dev)
    with _timed("gemm_f32", dict(M=M, N=N, K=C1 + C2), dev):
        rc = lib.d3f_gemm_f32t(A.data_ptr(), N1, lda, C1, idx.data_ptr() if idx is not None else None, ld_idx,
                               skip.data_ptr() if skip is not None else None, lds, C2, Wt.data_ptr(), out.data_ptr(), ldc, M, N,
                               row_scale.data_ptr() if row_scale is not None else None,
                               col_scale.data_ptr() if col_scale is not None else None,
                               col_shift.data_ptr() if col_shift is not None else None,
                               residual.data_ptr() if residual is not None else None, ldr, 1 if leaky else 0, float(alpha),
                               ws.data_ptr(), ws.numel(), m_dev, n1_dev, hint, _stream(dev))
    _lib.check(rc, "gemm_f32t")


def _bf16_ok(*operands):
    """(tensor, leading dimension, columns) triples: float4-addressable?"""
    for t, ld, cols in operands:
        if t is None:
            continue
        if cols % 4 or ld % 4 or t.data_ptr() % (8 if t.dtype == torch.bfloat16 else 16):
            return False
    return True


def _gemm_bf16(A, N1, lda, C1, idx, ld_idx, skip, lds, C2, W, out, M, N, row_scale, col_scale, col_shift, residual, ldr, leaky,
               alpha, m_dev, n1_dev, hint, dev):
    lib = _lib.load()
    Wp = packed_bf16_weights(W)
    # (the bf16 launcher plans its K split for a 64-column tile even when N <= 32: ask for the same plan's slab)
    ws = workspace(lib.d3f_gemm_bf16_workspace_bytes(M, N, C1 + C2, hint), dev)
    if out.data_ptr() % 16 or (residual is not None and (residual.data_ptr() % 8 or ldr % 4)):
        raise ValueError("gemm (bf16): output / residual must be 16 / 8-byte aligned with a leading dimension of 4 k")
    with _timed("gemm_bf16", dict(M=M, N=N, K=C1 + C2), dev):
        rc = lib.d3f_gemm_bf16(A.data_ptr(), N1, lda, C1, idx.data_ptr() if idx is not None else None, ld_idx,
                               skip.data_ptr() if skip is not None else None, lds, C2, Wp.data_ptr(), out.data_ptr(), N, M, N,
                               row_scale.data_ptr() if row_scale is not None else None,
                               col_scale.data_ptr() if col_scale is not None else None,
                               col_shift.data_ptr() if col_shift is not None else None,
                               residual.data_ptr() if residual is not None else None, ldr, 1 if leaky else 0, float(alpha),
                               ws.data_ptr(), ws.numel(), m_dev, n1_dev, hint, _h(A), _h(out), _stream(dev))
    _lib.check(rc, "gemm_bf16")


def gemm(A, Bm, row_scale=None, col_scale=None, col_shift=None, residual=None, leaky=False, alpha=0.2, out=None):
    """out = act((A @ Bm) * row_scale[:,None] * col_scale + col_shift + residual) on the matrix cores."""
    lib = _lib.load()
    A, lda = _rows(_feat(A, "A"), "A")
    Bm, ldb = _rows(_req(Bm, torch.float32, "B"), "B")
    M, K = A.shape
    if Bm.shape[0] != K:
        raise ValueError("gemm: A is %s, B is %s" % (tuple(A.shape), tuple(Bm.shape)))
    N = Bm.shape[1]
    dev = A.device
    if out is None:
        out = torch.empty((M, N), dtype=_out_dtype() if BF16_CONTRACTION else torch.float32, device=dev)
    out, ldc = _rows(out, "out")
    ldr = 0
    if residual is not None:
        residual, ldr = _rows(_feat(residual, "residual"), "residual")
    if (_h(A) or _h(out) or _h(residual)) and not (BF16_CONTRACTION and (residual is None or _h(residual) == _h(A))):
        raise TypeError("gemm: bfloat16 feature tensors need ops.bf16_contraction(features=True)")
    for v, n, name in ((row_scale, M, "row_scale"), (col_scale, N, "col_scale"), (col_shift, N, "col_shift")):
        if v is not None:
            _req(v, torch.float32, name)
            if v.numel() != n or not v.is_contiguous():
                raise ValueError("%s must be a contiguous vector of %d" % (name, n))
    hint = int(getattr(A, "n_hint", 0) or 0)
    if BF16_CONTRACTION and K >= 4 and ldc == N and Bm.is_contiguous() and _bf16_ok((A, lda, K)):
        _gemm_bf16(A, M, lda, K, None, 0, None, 0, 0, Bm, out, M, N, row_scale, col_scale, col_shift, residual, ldr, leaky, alpha,
                   _nd(A), None, hint, dev)
        return _tag(out, A)
    if _h(A) or _h(out):
        raise TypeError("gemm: bfloat16 operands that the bf16 contraction cannot address (K %d, lda %d)" % (K, lda))
    if Bm.is_contiguous() and _f32t_ok(N, ldc, out, residual, ldr, (col_scale, col_shift), (A, lda, K)):
        _gemm_f32t(A, M, lda, K, None, 0, None, 0, 0, Bm, out, ldc, M, N, row_scale, col_scale, col_shift, residual, ldr, leaky, alpha,
                   _nd(A), None, hint, dev)
        return _tag(out, A)
    nbytes = lib.d3f_gemm_workspace_bytes(M, N, K, hint)
    ws = workspace(nbytes, dev)
    with _timed("gemm_f32", dict(M=M, N=N, K=K), dev):
        rc = lib.d3f_gemm_f32(A.data_ptr(), lda, Bm.data_ptr(), ldb, out.data_ptr(), ldc, M, N, K,
                              row_scale.data_ptr() if row_scale is not None else None,
                              col_scale.data_ptr() if col_scale is not None else None,
                              col_shift.data_ptr() if col_shift is not None else None,
                              residual.data_ptr() if residual is not None else None, ldr,
                              1 if leaky else 0, float(alpha), ws.data_ptr(), ws.numel(), _nd(A), hint, _stream(dev))
    _lib.check(rc, "gemm_f32")
    return _tag(out, A)


class UpsampleCat:
    """The not-yet-materialised result of nearest_upsample + skip concatenation (models/D3Feat.py:55-63):
    rows [ x'[inds[n,0]] | skip[n] ].  The unary block that follows consumes it through gemm_upsample_cat (one launch, no
    intermediate tensor); anything else calls .materialize()."""

    def __init__(self, x, inds, skip=None):
        self.x, self.inds, self.skip = x, inds, skip
        self.shape = (inds.shape[0], x.shape[1] + (skip.shape[1] if skip is not None else 0))
        self.device = x.device
        _tag(self, inds)

    def materialize(self):
        return closest_pool_cat(self.x, self.inds, self.skip)


def gemm_upsample_cat(u, W, col_scale=None, col_shift=None, leaky=False, alpha=0.2):
    """out = act(([ x'[inds[:,0]] | skip ] @ W) * col_scale + col_shift) without building the concatenation."""
    lib = _lib.load()
    x, ldx = _rows(_feat(u.x, "x"), "x")
    inds, ldi = _rows(_req(u.inds, torch.int32, "inds"), "inds")
    W, ldb = _rows(_req(W, torch.float32, "W"), "W")
    C1, C2, lds, skip = x.shape[1], 0, 0, None
    if u.skip is not None:
        skip, lds = _rows(_feat(u.skip, "skip"), "skip")
        C2 = skip.shape[1]
    M, N = inds.shape[0], W.shape[1]
    if W.shape[0] != C1 + C2:
        raise ValueError("gemm_upsample_cat: W has %d rows, operands %d + %d columns" % (W.shape[0], C1, C2))
    if C2 and (C1 % 4 or lds % 4 or skip.data_ptr() % 16) and not _h(x):
        return gemm(u.materialize(), W, col_scale=col_scale, col_shift=col_shift, leaky=leaky, alpha=alpha)
    dev = x.device
    out = torch.empty((M, N), dtype=_out_dtype() if BF16_CONTRACTION else torch.float32, device=dev)
    hint = int(getattr(inds, "n_hint", 0) or 0)
    if BF16_CONTRACTION and W.is_contiguous() and _bf16_ok((x, ldx, C1), (skip, lds, C2)):
        _gemm_bf16(x, x.shape[0], ldx, C1, inds, ldi, skip, lds, C2, W, out, M, N, None, col_scale, col_shift, None, 0, leaky, alpha,
                   _nd(inds), _nd(x), hint, dev)
        return _tag(out, inds)
    if _h(x) or _h(out):
        raise TypeError("gemm_upsample_cat: bfloat16 operands that the bf16 contraction cannot address")
    if W.is_contiguous() and _f32t_ok(N, N, out, None, 0, (col_scale, col_shift), (x, ldx, C1), (skip, lds, C2) if C2 else (None, 0, 0)):
        _gemm_f32t(x, x.shape[0], ldx, C1, inds, ldi, skip, lds, C2, W, out, N, M, N, None, col_scale, col_shift, None, 0, leaky, alpha,
                   _nd(inds), _nd(x), hint, dev)
        return _tag(out, inds)
    nbytes = lib.d3f_gemm_workspace_bytes(M, N, C1 + C2, hint)
    ws = workspace(nbytes, dev)
    with _timed("gemm_f32", dict(M=M, N=N, K=C1 + C2), dev):
        rc = lib.d3f_gemm_upsample_cat_f32(x.data_ptr(), x.shape[0], ldx, C1, inds.data_ptr(), ldi,
                                           skip.data_ptr() if C2 else None, lds, C2, W.data_ptr(), ldb, out.data_ptr(), N, M, N,
                                           col_scale.data_ptr() if col_scale is not None else None,
                                           col_shift.data_ptr() if col_shift is not None else None, 1 if leaky else 0,
                                           float(alpha), ws.data_ptr(), ws.numel(), _nd(inds), _nd(x), hint, _stream(dev))
    _lib.check(rc, "gemm_upsample_cat_f32")
    return _tag(out, inds)


def gemm_cat2(A1, A2, W, col_scale=None, col_shift=None, leaky=False, alpha=0.2):
    """out = act(([A1 | A2] @ W) * col_scale + col_shift) without building the concatenation (same rows in A1 and A2)."""
    lib = _lib.load()
    A1, ld1 = _rows(_feat(A1, "A1"), "A1")
    A2, ld2 = _rows(_feat(A2, "A2"), "A2")
    W, ldb = _rows(_req(W, torch.float32, "W"), "W")
    M, C1, C2, N = A1.shape[0], A1.shape[1], A2.shape[1], W.shape[1]
    if A2.shape[0] != M or W.shape[0] != C1 + C2:
        raise ValueError("gemm_cat2: operands %s | %s, W %s" % (tuple(A1.shape), tuple(A2.shape), tuple(W.shape)))
    if (C1 % 4 or ld2 % 4 or A2.data_ptr() % 16) and not _h(A1):
        return gemm(torch.cat([A1, A2], 1), W, col_scale=col_scale, col_shift=col_shift, leaky=leaky, alpha=alpha)
    dev = A1.device
    out = torch.empty((M, N), dtype=_out_dtype() if BF16_CONTRACTION else torch.float32, device=dev)
    hint = int(getattr(A1, "n_hint", 0) or 0)
    if BF16_CONTRACTION and W.is_contiguous() and _bf16_ok((A1, ld1, C1), (A2, ld2, C2)):
        _gemm_bf16(A1, M, ld1, C1, None, 0, A2, ld2, C2, W, out, M, N, None, col_scale, col_shift, None, 0, leaky, alpha,
                   _nd(A1), _nd(A1), hint, dev)
        return _tag(out, A1)
    if _h(A1) or _h(A2) or _h(out):
        raise TypeError("gemm_cat2: bfloat16 operands that the bf16 contraction cannot address")
    if W.is_contiguous() and _f32t_ok(N, N, out, None, 0, (col_scale, col_shift), (A1, ld1, C1), (A2, ld2, C2)):
        _gemm_f32t(A1, M, ld1, C1, None, 0, A2, ld2, C2, W, out, N, M, N, None, col_scale, col_shift, None, 0, leaky, alpha,
                   _nd(A1), _nd(A1), hint, dev)
        return _tag(out, A1)
    nbytes = lib.d3f_gemm_workspace_bytes(M, N, C1 + C2, hint)
    ws = workspace(nbytes, dev)
    with _timed("gemm_f32", dict(M=M, N=N, K=C1 + C2), dev):
        rc = lib.d3f_gemm_upsample_cat_f32(A1.data_ptr(), M, ld1, C1, None, 0, A2.data_ptr(), ld2, C2, W.data_ptr(), ldb,
                                           out.data_ptr(), N, M, N,
                                           col_scale.data_ptr() if col_scale is not None else None,
                                           col_shift.data_ptr() if col_shift is not None else None, 1 if leaky else 0,
                                           float(alpha), ws.data_ptr(), ws.numel(), _nd(A1), _nd(A1), hint, _stream(dev))
    _lib.check(rc, "gemm_cat2")
    return _tag(out, A1)


def kpconv_aggregate(query_points, support_points, neighbors_indices, features, K_points, KP_extent,
                     KP_influence="linear", aggregation_mode="sum"):
    """-> (wf f32[Nq, num_kp*Cin], inv_cnt f32[Nq])   (phase 1 of KPConv_ops)."""
    lib = _lib.load()
    q = _req(query_points, torch.float32, "query_points", 2).contiguous()
    s = _req(support_points, torch.float32, "support_points", 2).contiguous()
    idx, ld_idx = _rows(_req(neighbors_indices, torch.int32, "neighbors_indices"), "neighbors_indices")
    f, ldf = _rows(_feat(features, "features"), "features")
    if KP_influence not in _INFLUENCE:
        raise ValueError("Unknown influence function type (config.KP_influence)")
    if aggregation_mode not in _AGGREGATION:
        raise ValueError("Unknown convolution mode. Should be 'closest' or 'sum'")
    kp = _kp_host(K_points)
    num_kp = kp.shape[0]
    Nq, Ns, K, Cin = q.shape[0], s.shape[0], idx.shape[1], f.shape[1]
    if idx.shape[0] != Nq or f.shape[0] != Ns:
        raise ValueError("KPConv: %d queries / %d index rows, %d supports / %d feature rows" %
                         (Nq, idx.shape[0], Ns, f.shape[0]))
    dev = q.device
    wf = torch.empty((Nq, num_kp * Cin), dtype=torch.float32, device=dev)
    inv_cnt = torch.empty((Nq,), dtype=torch.float32, device=dev)
    row_pos = torch.empty((max(Ns, 1),), dtype=torch.uint8, device=dev)
    st = _stream(dev)
    nq_dev, ns_dev = _nd(query_points), _nd(support_points)
    if ns_dev is None:
        ns_dev = _nd(features)
    _lib.check(lib.d3f_row_positive(f.data_ptr(), Ns, ldf, Cin, row_pos.data_ptr(), ns_dev, _h(f), st), "row_positive")
    with _timed("kpconv_aggregate", dict(Nq=Nq, Ns=Ns, K=K, Cin=Cin), dev):
        rc = lib.d3f_kpconv_aggregate(q.data_ptr(), Nq, s.data_ptr(), Ns, idx.data_ptr(), ld_idx, K, f.data_ptr(), ldf,
                                      Cin, row_pos.data_ptr(), kp.ctypes.data, num_kp, float(KP_extent),
                                      _INFLUENCE[KP_influence], _AGGREGATION[aggregation_mode], wf.data_ptr(),
                                      inv_cnt.data_ptr(), nq_dev, ns_dev, _order(query_points), _h(f), st)
    _lib.check(rc, "kpconv_aggregate")
    return _tag(wf, query_points), _tag(inv_cnt, query_points)


def kpconv_fused32(query_points, support_points, neighbors_indices, features, K_points, K_values, KP_extent,
                   KP_influence="linear", aggregation_mode="sum", col_scale=None, col_shift=None, residual=None,
                   leaky=False, alpha=0.2):
    """Whole KPConv (+ epilogue) for Cin = Cout = 32 in one kernel (the aggregation tile is contracted from LDS)."""
    lib = _lib.load()
    q = _req(query_points, torch.float32, "query_points", 2).contiguous()
    s = _req(support_points, torch.float32, "support_points", 2).contiguous()
    idx, ld_idx = _rows(_req(neighbors_indices, torch.int32, "neighbors_indices"), "neighbors_indices")
    f, ldf = _rows(_feat(features, "features"), "features")
    kp = _kp_host(K_points)
    num_kp, cin, cout = K_values.shape
    if cin != 32 or cout != 32 or f.shape[1] != 32:
        raise ValueError("kpconv_fused32 needs Cin == Cout == 32")
    x3 = KP_X3 and (num_kp * cin) % 32 == 0
    # the aggregation on the matrix cores as well (d3f_kpconv_fused32_mfma): the shipped configuration, fp32 features
    mfma = (KP_MFMA and x3 and num_kp == 15 and KP_influence == "linear" and aggregation_mode == "sum" and f.dtype == torch.float32
            and idx.shape[1] <= 64)
    if mfma:
        W = packed_kpconv_weights_x3m(K_values)
    else:
        W = packed_kpconv_weights_x3(K_values) if x3 else _req(K_values, torch.float32, "K_values").reshape(num_kp * cin, cout).contiguous()
    Nq, Ns, K = q.shape[0], s.shape[0], idx.shape[1]
    dev = q.device
    out = torch.empty((Nq, cout), dtype=f.dtype, device=dev)
    row_pos = torch.empty((max(Ns, 1),), dtype=torch.uint8, device=dev)
    ldr = 0
    if residual is not None:
        residual, ldr = _rows(_req(residual, torch.float32, "residual"), "residual")
    st = _stream(dev)
    nq_dev, ns_dev = _nd(query_points), _nd(support_points)
    if ns_dev is None:
        ns_dev = _nd(features)
    _lib.check(lib.d3f_row_positive(f.data_ptr(), Ns, ldf, 32, row_pos.data_ptr(), ns_dev, _h(f), st), "row_positive")
    if mfma:
        with _timed("kpconv_fused32", dict(Nq=Nq, Ns=Ns, K=K, Cin=32, Cout=32), dev):
            rc = lib.d3f_kpconv_fused32_mfma(q.data_ptr(), Nq, s.data_ptr(), Ns, idx.data_ptr(), ld_idx, K, f.data_ptr(), ldf,
                                             row_pos.data_ptr(), kp.ctypes.data, num_kp, float(KP_extent), _INFLUENCE[KP_influence],
                                             _AGGREGATION[aggregation_mode], W.data_ptr(),
                                             col_scale.data_ptr() if col_scale is not None else None,
                                             col_shift.data_ptr() if col_shift is not None else None,
                                             residual.data_ptr() if residual is not None else None, ldr, 1 if leaky else 0,
                                             float(alpha), out.data_ptr(), cout, nq_dev, ns_dev, _order(query_points), st)
        _lib.check(rc, "kpconv_fused32_mfma")
        return _tag(out, query_points)
    with _timed("kpconv_fused32", dict(Nq=Nq, Ns=Ns, K=K, Cin=32, Cout=32), dev):
        rc = (lib.d3f_kpconv_fused32_x3 if x3 else lib.d3f_kpconv_fused32)(q.data_ptr(), Nq, s.data_ptr(), Ns, idx.data_ptr(), ld_idx, K, f.data_ptr(), ldf,
                                    row_pos.data_ptr(), kp.ctypes.data, num_kp, float(KP_extent), _INFLUENCE[KP_influence],
                                    _AGGREGATION[aggregation_mode], W.data_ptr(),
                                    col_scale.data_ptr() if col_scale is not None else None,
                                    col_shift.data_ptr() if col_shift is not None else None,
                                    residual.data_ptr() if residual is not None else None, ldr, 1 if leaky else 0,
                                    float(alpha), out.data_ptr(), cout, nq_dev, ns_dev, _order(query_points), _h(f), st)
    _lib.check(rc, "kpconv_fused32")
    return _tag(out, query_points)


def kpconv_fused_supported(cin, cout, num_kp, KP_influence, aggregation_mode, available=False):
    """Should KPConv_ops use the one-kernel form for this shape?  available=True: does the form exist at all (Cin = 256 exists
    but measured no faster than aggregation + contraction: see d3f_kpconv_fused_supported)."""
    r = _lib.load().d3f_kpconv_fused_supported(int(cin), int(cout), int(num_kp), _INFLUENCE.get(KP_influence, -1),
                                               _AGGREGATION.get(aggregation_mode, -1))
    return r >= 1 if available else r == 1


def packed_kpconv_weights(K_values):
    """K_values f32[num_kp, Cin, Cout] -> the k-block-packed copy d3f_kpconv_fused reads (made once per weight tensor and
    version: the result rides on the tensor object, so a model's cached device weights are packed at their first use --
    the engine's eager warm-up -- and never inside a captured graph)."""
    cached = getattr(K_values, "_d3f_packed", None)
    if cached is not None and cached[0] == K_values._version:
        return cached[1]
    lib = _lib.load()
    num_kp, cin, cout = K_values.shape
    W = _req(K_values, torch.float32, "K_values").reshape(num_kp * cin, cout).contiguous()
    # an in-place update of the weights re-packs into the SAME buffer (captured graphs hold its address: _packed_on_tensor)
    Wp = cached[1] if cached is not None else torch.empty_like(W)
    _lib.check(lib.d3f_kpconv_pack_weights(W.data_ptr(), num_kp * cin, cout, Wp.data_ptr(), _stream(W.device)), "kpconv_pack_weights")
    K_values._d3f_packed = (K_values._version, Wp)
    return Wp


# The fused KPConv kernels of levels 1 and 2 contract their LDS tile in the operand-split form (csrc/kpconv.hip, round 5: three exact
# bf16 planes per operand, six exact products per fp32 product, fp32 accumulate -- fp32 in, fp32 out); D3F_KP_X3=0 keeps the
# v_mfma_f32_16x16x4_f32 form.
KP_X3 = os.environ.get("D3F_KP_X3", "1") != "0"


# the level-0 KPConv with its aggregation on the matrix cores too (csrc/kpconv.hip: kpconv_fused32m_kernel): OFF by default -- built,
# verified (tests/test_gpu_kpconv_x3.py) and measured SLOWER than the vector form (profiles/r06_experiments.txt k1-k6: 247-298 against
# 220-235 us per level-0 launch at F = 4): these kernels are bound by the latency of a workgroup's dependent chain at three workgroups
# per CU, not by the vector pipe.  D3F_KP_MFMA=1 selects it.
KP_MFMA = os.environ.get("D3F_KP_MFMA", "0") == "1"


def packed_kpconv_weights_x3m(K_values):
    """K_values f32[15, Cin, Cout] -> the pre-split bf16 planes of W'[16 s + p][n] = K_values[p][c(s)][n] (p < 15; 0 for the 16th slot;
    c(s) = the even channels, then the odd ones):
    the channel-major k order in which the matrix-core aggregation leaves its weighted features.  Rides on the tensor like the other
    packed copies (re-packed in place when the tensor's version changes)."""
    cached = getattr(K_values, "_d3f_packed_x3m", None)
    if cached is not None and cached[0] == K_values._version:
        return cached[1]
    lib = _lib.load()
    num_kp, cin, cout = K_values.shape
    Wp = torch.zeros((cin, 16, cout), dtype=torch.float32, device=K_values.device)
    order = list(range(0, cin, 2)) + list(range(1, cin, 2))             # the even channels (chain 0 of the kernel), then the odd ones
    Wp[:, :num_kp] = _req(K_values, torch.float32, "K_values").permute(1, 0, 2)[order]
    Wp = Wp.reshape(cin * 16, cout).contiguous()
    Wx = cached[1] if cached is not None else \
        torch.empty((int(lib.d3f_kpconv_packed_x3_bytes(cin * 16, cout)) // 2,), dtype=torch.int16, device=Wp.device)
    _lib.check(lib.d3f_kpconv_pack_weights_x3(Wp.data_ptr(), cin * 16, cout, Wx.data_ptr(), _stream(Wp.device)), "kpconv_pack_weights_x3")
    K_values._d3f_packed_x3m = (K_values._version, Wx)
    return Wx


def packed_kpconv_weights_x3(K_values):
    """K_values f32[num_kp, Cin, Cout] -> the pre-split bf16 planes d3f_kpconv_fused_x3 reads; made once per weight tensor and
    version (rides on the tensor object like packed_kpconv_weights: never inside a captured graph)."""
    cached = getattr(K_values, "_d3f_packed_x3", None)
    if cached is not None and cached[0] == K_values._version:
        return cached[1]
    lib = _lib.load()
    num_kp, cin, cout = K_values.shape
    W = _req(K_values, torch.float32, "K_values").reshape(num_kp * cin, cout).contiguous()
    Wx = cached[1] if cached is not None else \
        torch.empty((int(lib.d3f_kpconv_packed_x3_bytes(num_kp * cin, cout)) // 2,), dtype=torch.int16, device=W.device)
    _lib.check(lib.d3f_kpconv_pack_weights_x3(W.data_ptr(), num_kp * cin, cout, Wx.data_ptr(), _stream(W.device)), "kpconv_pack_weights_x3")
    K_values._d3f_packed_x3 = (K_values._version, Wx)
    return Wx


def kpconv_fused(query_points, support_points, neighbors_indices, features, K_points, K_values, KP_extent,
                 KP_influence="linear", aggregation_mode="sum", col_scale=None, col_shift=None, residual=None,
                 leaky=False, alpha=0.2):
    """Whole KPConv (+ epilogue) for Cin == Cout in {64, 128} in one kernel (kpconv_fused_supported says when)."""
    lib = _lib.load()
    q = _req(query_points, torch.float32, "query_points", 2).contiguous()
    s = _req(support_points, torch.float32, "support_points", 2).contiguous()
    idx, ld_idx = _rows(_req(neighbors_indices, torch.int32, "neighbors_indices"), "neighbors_indices")
    f, ldf = _rows(_feat(features, "features"), "features")
    kp = _kp_host(K_points)
    num_kp, cin, cout = K_values.shape
    if f.shape[1] != cin:
        raise ValueError("kpconv_fused: features have %d channels, K_values expects %d" % (f.shape[1], cin))
    x3 = KP_X3 and (num_kp * cin) % 32 == 0 and cout % 16 == 0
    Wp = packed_kpconv_weights_x3(K_values) if x3 else packed_kpconv_weights(K_values)
    Nq, Ns, K = q.shape[0], s.shape[0], idx.shape[1]
    dev = q.device
    out = torch.empty((Nq, cout), dtype=f.dtype, device=dev)
    row_pos = torch.empty((max(Ns, 1),), dtype=torch.uint8, device=dev)
    ldr = 0
    if residual is not None:
        residual, ldr = _rows(_req(residual, torch.float32, "residual"), "residual")
    st = _stream(dev)
    nq_dev, ns_dev = _nd(query_points), _nd(support_points)
    if ns_dev is None:
        ns_dev = _nd(features)
    _lib.check(lib.d3f_row_positive(f.data_ptr(), Ns, ldf, cin, row_pos.data_ptr(), ns_dev, _h(f), st), "row_positive")
    with _timed("kpconv_fused", dict(Nq=Nq, Ns=Ns, K=K, Cin=cin, Cout=cout), dev):
        rc = (lib.d3f_kpconv_fused_x3 if x3 else lib.d3f_kpconv_fused)(
                                  q.data_ptr(), Nq, s.data_ptr(), Ns, idx.data_ptr(), ld_idx, K, f.data_ptr(), ldf, cin,
                                  row_pos.data_ptr(), kp.ctypes.data, num_kp, float(KP_extent), _INFLUENCE[KP_influence],
                                  _AGGREGATION[aggregation_mode], Wp.data_ptr(), cout,
                                  col_scale.data_ptr() if col_scale is not None else None,
                                  col_shift.data_ptr() if col_shift is not None else None,
                                  residual.data_ptr() if residual is not None else None, ldr, 1 if leaky else 0,
                                  float(alpha), out.data_ptr(), cout, nq_dev, ns_dev, _order(query_points), _h(f), st)
    _lib.check(rc, "kpconv_fused")
    return _tag(out, query_points)


def kpconv_fused_c1(query_points, support_points, neighbors_indices, features, K_points, K_values, KP_extent,
                    KP_influence="linear", aggregation_mode="sum", col_scale=None, col_shift=None, residual=None,
                    leaky=False, alpha=0.2):
    """Whole KPConv (+ epilogue) for Cin = 1 in one kernel.  K_values f32[num_kp, 1, Cout]."""
    lib = _lib.load()
    q = _req(query_points, torch.float32, "query_points", 2).contiguous()
    s = _req(support_points, torch.float32, "support_points", 2).contiguous()
    idx, ld_idx = _rows(_req(neighbors_indices, torch.int32, "neighbors_indices"), "neighbors_indices")
    f, ldf = _rows(_req(features, torch.float32, "features"), "features")
    kp = _kp_host(K_points)
    num_kp, cin, cout = K_values.shape
    if cin != 1 or f.shape[1] != 1:
        raise ValueError("kpconv_fused_c1 needs Cin == 1")
    W = _req(K_values, torch.float32, "K_values").reshape(num_kp, cout).contiguous()
    Nq, Ns, K = q.shape[0], s.shape[0], idx.shape[1]
    dev = q.device
    out = torch.empty((Nq, cout), dtype=torch.bfloat16 if BF16_FEATURES else torch.float32, device=dev)
    ldr = 0
    if residual is not None:
        residual, ldr = _rows(_req(residual, torch.float32, "residual"), "residual")
    with _timed("kpconv_fused_c1", dict(Nq=Nq, Ns=Ns, K=K, Cin=1, Cout=cout), dev):
        rc = lib.d3f_kpconv_fused_c1(q.data_ptr(), Nq, s.data_ptr(), Ns, idx.data_ptr(), ld_idx, K, f.data_ptr(), ldf,
                                     kp.ctypes.data, num_kp, float(KP_extent), _INFLUENCE[KP_influence],
                                     _AGGREGATION[aggregation_mode], W.data_ptr(), cout,
                                     col_scale.data_ptr() if col_scale is not None else None,
                                     col_shift.data_ptr() if col_shift is not None else None,
                                     residual.data_ptr() if residual is not None else None, ldr, 1 if leaky else 0,
                                     float(alpha), out.data_ptr(), cout, _nd(query_points), _nd(support_points),
                                     _order(query_points), _h(out), _stream(dev))
    _lib.check(rc, "kpconv_fused_c1")
    return _tag(out, query_points)


def ind_max_pool(x, inds):
    lib = _lib.load()
    x, ldx = _rows(_feat(x, "x"), "x")
    inds, ldi = _rows(_req(inds, torch.int32, "inds"), "inds")
    dev = x.device
    out = torch.empty((inds.shape[0], x.shape[1]), dtype=x.dtype, device=dev)
    with _timed("ind_max_pool", dict(N1=x.shape[0], N2=inds.shape[0], K=inds.shape[1], C=x.shape[1]), dev):
        rc = lib.d3f_ind_max_pool(x.data_ptr(), x.shape[0], ldx, x.shape[1], inds.data_ptr(), inds.shape[0], ldi,
                                  inds.shape[1], out.data_ptr(), x.shape[1], None, _nd(x), _nd(inds), _order(inds),
                                  _h(x), _stream(dev))
    _lib.check(rc, "ind_max_pool")
    return _tag(out, inds)


def closest_pool_cat(x, inds, skip=None):
    lib = _lib.load()
    x, ldx = _rows(_req(x, torch.float32, "x"), "x")
    inds, ldi = _rows(_req(inds, torch.int32, "inds"), "inds")
    dev = x.device
    C1 = x.shape[1]
    C2 = lds = 0
    if skip is not None:
        skip, lds = _rows(_req(skip, torch.float32, "skip"), "skip")
        C2 = skip.shape[1]
        if skip.shape[0] != inds.shape[0]:
            raise ValueError("closest_pool_cat: %d index rows, %d skip rows" % (inds.shape[0], skip.shape[0]))
    out = torch.empty((inds.shape[0], C1 + C2), dtype=torch.float32, device=dev)
    rc = lib.d3f_closest_pool_cat(x.data_ptr(), x.shape[0], ldx, C1, inds.data_ptr(), inds.shape[0], ldi,
                                  skip.data_ptr() if skip is not None else None, lds, C2, out.data_ptr(), C1 + C2,
                                  _nd(x), _nd(inds), _stream(dev))
    _lib.check(rc, "closest_pool_cat")
    return _tag(out, inds)


def detect_head(x, neighbors, stack_lengths_dev, include_zero_dev, stack_group=0):
    """-> (desc f32[N,C], score f32[N,1]).  stack_group: clouds per reference stack inside a batched stack (0: the whole
    stack is one reference stack), see include/d3feat_amd.h."""
    lib = _lib.load()
    x, ldx = _rows(_req(x, torch.float32, "x"), "x")
    nb, ldi = _rows(_req(neighbors, torch.int32, "neighbors"), "neighbors")
    dev = x.device
    N, Cc = x.shape
    B = stack_lengths_dev.numel()
    desc = torch.empty((N, Cc), dtype=torch.float32, device=dev)
    score = torch.empty((N, 1), dtype=torch.float32, device=dev)
    scratch = torch.empty((2 * B + 2 + (N + 3) // 4,), dtype=torch.int32, device=dev)
    with _timed("detect_head", dict(N=N, K=nb.shape[1], C=Cc), dev):
        rc = lib.d3f_detect_head(x.data_ptr(), N, ldx, Cc, nb.data_ptr(), ldi, nb.shape[1],
                                 stack_lengths_dev.data_ptr(),
                                 include_zero_dev.data_ptr() if include_zero_dev is not None else None,
                                 int(stack_group), B, desc.data_ptr(), Cc, score.data_ptr(), scratch.data_ptr(), _order(nb),
                                 _stream(dev))
    _lib.check(rc, "detect_head")
    return _tag(desc, x), _tag(score, x)


class RawRecords:
    """A cloud still in its file encoding (utils.ply.read_ply_records / utils.results.read_kitti_records): raw bytes + record
    layout.  FragmentEngine.submit takes it wherever it takes a float32 [n,3] cloud and decodes it on the GPU, straight into
    the slot's stage-0 input buffer."""

    def __init__(self, raw, layout, pin=True):
        if not isinstance(raw, torch.Tensor):
            raw = torch.from_numpy(np.ascontiguousarray(raw, dtype=np.uint8))
        if pin and not raw.is_cuda and torch.cuda.is_available():
            raw = raw.pin_memory()
        self.raw, self.layout = raw, dict(layout)
        self.shape = (int(layout["n"]), 3)
        self.is_cuda = raw.is_cuda

    def decode(self, device=None, out=None):
        return decode_xyz_records(self.raw if device is None or self.raw.is_cuda else self.raw.to(device, non_blocking=True),
                                  self.layout, out=out)


def decode_xyz_records(raw, layout, out=None):
    """Raw file records (uint8 tensor on the device, or pinned / plain host bytes that are copied there asynchronously) ->
    float32 [n,3] on the device.  layout: utils.ply.read_ply_records / utils.results.read_kitti_records."""
    lib = _lib.load()
    dev = out.device if out is not None else torch.device("cuda", torch.cuda.current_device())
    if not isinstance(raw, torch.Tensor):
        raw = torch.from_numpy(np.ascontiguousarray(raw, dtype=np.uint8))
    if not raw.is_cuda:
        raw = raw.to(dev, non_blocking=True)
    raw = raw.contiguous()
    n, stride = int(layout["n"]), int(layout["stride"])
    if raw.dtype != torch.uint8 or raw.numel() < n * stride:
        raise ValueError("decode_xyz_records: %d bytes for %d records of %d bytes" % (raw.numel(), n, stride))
    if out is None:
        out = torch.empty((n, 3), dtype=torch.float32, device=dev)
    elif out.shape[0] < n or not out.is_contiguous() or out.dtype != torch.float32:
        raise ValueError("decode_xyz_records: output buffer too small or not contiguous float32")
    ox, oy, oz = (int(v) for v in layout["offsets"])
    rc = lib.d3f_decode_xyz_records(raw.data_ptr(), n, stride, ox, oy, oz, 1 if layout["dtype"] == "f8" else 0,
                                    1 if layout.get("big_endian") else 0, out.data_ptr(), _stream(dev))
    _lib.check(rc, "decode_xyz_records")
    return out[:n]


def pack_descriptors(xyz, desc, score, lens=None, group=1, keep=0, dst=None, row_map=None):
    """-> f32[N, 3 + C + 1] records [xyz | desc | score] (one contiguous block per fragment of a stack).
    dst (device int64[fragments] of ADDRESSES, 0 = none) with lens (device i32[B], the stack's clouds), group clouds per fragment:
    the rows of the first `keep` clouds of fragment f are written to the address dst[f] instead (d3f_pack_descriptors_to).
    row_map (device i32[N]): the inputs are in an internal row order, record n is written at row row_map[n]."""
    lib = _lib.load()
    xyz = _req(xyz, torch.float32, "xyz", 2).contiguous()
    desc, ldd = _rows(_req(desc, torch.float32, "desc"), "desc")
    score = _req(score, torch.float32, "score").contiguous()
    N, Cc = desc.shape
    if xyz.shape[0] != N or xyz.shape[1] != 3 or score.numel() != N:
        raise ValueError("pack_descriptors: %s points, %s descriptors, %s scores" % (tuple(xyz.shape), tuple(desc.shape),
                                                                                     tuple(score.shape)))
    out = torch.empty((N, Cc + 4), dtype=torch.float32, device=desc.device)
    if dst is not None or row_map is not None:
        if dst is not None:
            assert dst.dtype == torch.int64 and dst.is_contiguous() and lens is not None and lens.dtype == torch.int32
            assert lens.numel() == dst.numel() * int(group)
        if row_map is not None:
            assert row_map.dtype == torch.int32 and row_map.is_contiguous() and row_map.numel() >= N
        rc = lib.d3f_pack_descriptors_to(xyz.data_ptr(), desc.data_ptr(), ldd, Cc, score.data_ptr(), N, out.data_ptr(), Cc + 4,
                                         _nd(xyz) or _nd(desc), lens.data_ptr() if lens is not None else None,
                                         lens.numel() if lens is not None else 0, int(group), int(keep),
                                         dst.data_ptr() if dst is not None else None,
                                         row_map.data_ptr() if row_map is not None else None, _stream(desc.device))
    else:
        rc = lib.d3f_pack_descriptors(xyz.data_ptr(), desc.data_ptr(), ldd, Cc, score.data_ptr(), N, out.data_ptr(), Cc + 4,
                                      _nd(xyz) or _nd(desc), _stream(desc.device))
    _lib.check(rc, "pack_descriptors")
    return _tag(out, desc)


def affine_act(x, col_scale=None, col_shift=None, residual=None, leaky=False, alpha=0.2):
    """out = act(x * col_scale + col_shift + residual): the stand-alone form of the GEMM epilogue."""
    lib = _lib.load()
    x, ldx = _rows(_req(x, torch.float32, "x"), "x")
    M, N = x.shape
    ldr = 0
    if residual is not None:
        residual, ldr = _rows(_req(residual, torch.float32, "residual"), "residual")
    out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    rc = lib.d3f_affine_act(x.data_ptr(), ldx, M, N, col_scale.data_ptr() if col_scale is not None else None,
                            col_shift.data_ptr() if col_shift is not None else None,
                            residual.data_ptr() if residual is not None else None, ldr, 1 if leaky else 0, float(alpha),
                            out.data_ptr(), N, _nd(x), _stream(x.device))
    _lib.check(rc, "affine_act")
    return _tag(out, x)
