"""Downstream matching on the MI355X: what the reference does with descriptors after the hot path.

    feature_nn(A, B)                        nearest descriptor, one direction           geometric_registration/evaluate.py:17-21
    build_correspondence(src_desc, tgt_desc) mutually closest pairs                      evaluate.py:11-27 (same name)
    ransac_feature_matching(...)            open3d.registration_ransac_based_on_feature_matching as the reference calls it
                                            (evaluate.py:93-99, demo_registration.py:184-192)

Every computation is a kernel of libd3feat_amd.so (csrc/registration.hip, csrc/radius_neighbors.hip); numpy / torch only
move data and run the host loop over batches of hypotheses.  Open3D's own random stream is unspecified, so results are
deterministic functions of `seed` here and are pinned to oracle/registration_np.py (same algorithm, same random numbers).
"""
import numpy as np
import torch

from . import _lib, ops


def _dev(device=None):
    return device if device is not None else torch.device("cuda", torch.cuda.current_device())


def _f32(a, device):
    if isinstance(a, torch.Tensor):
        return a.to(device=device, dtype=torch.float32).contiguous()
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)


def feature_nn(A, B, return_d2=False, device=None):
    """idx[i] = argmin_j ||A_i - B_j||^2 (ties to the lowest j).  A [n,C], B [m,C], C in {16,32,64} -> int32 [n] (device)."""
    lib = _lib.load()
    dev = _dev(device)
    A, B = _f32(A, dev), _f32(B, dev)
    if A.dim() != 2 or B.dim() != 2 or A.shape[1] != B.shape[1]:
        raise ValueError("feature_nn: A %s, B %s" % (tuple(A.shape), tuple(B.shape)))
    n, m, C = A.shape[0], B.shape[0], A.shape[1]
    idx = torch.empty((n,), dtype=torch.int32, device=dev)
    d2 = torch.empty((n,), dtype=torch.float32, device=dev) if return_d2 else None
    ws = ops.workspace(lib.d3f_feature_nn_workspace_bytes(n), dev)
    rc = lib.d3f_feature_nn(A.data_ptr(), n, C, B.data_ptr(), m, C, C, idx.data_ptr(), d2.data_ptr() if return_d2 else None,
                            ws.data_ptr(), ws.numel(), ops._stream(dev))
    _lib.check(rc, "feature_nn")
    return (idx, d2) if return_d2 else idx


def build_correspondence(source_desc, target_desc, device=None):
    """evaluate.py:11-27: the mutually closest pairs in feature space -> int array [k, 2] (host), ascending source index."""
    lib = _lib.load()
    dev = _dev(device)
    A, B = _f32(source_desc, dev), _f32(target_desc, dev)
    ab, ba = feature_nn(A, B, device=dev), feature_nn(B, A, device=dev)
    n = A.shape[0]
    pairs = torch.empty((max(n, 1), 2), dtype=torch.int32, device=dev)
    count = torch.zeros((1,), dtype=torch.int32, device=dev)
    ws = ops.workspace(lib.d3f_mutual_matches_workspace_bytes(n), dev)
    rc = lib.d3f_mutual_matches(ab.data_ptr(), n, ba.data_ptr(), B.shape[0], pairs.data_ptr(), count.data_ptr(), ws.data_ptr(),
                                ws.numel(), ops._stream(dev))
    _lib.check(rc, "mutual_matches")
    k = int(count.item())
    return pairs[:k].cpu().numpy().astype(np.int64)


def ransac_feature_matching(source_points, target_points, source_desc, target_desc, max_correspondence_distance, ransac_n=4,
                            edge_similarity=0.9, checker_distance=None, max_iteration=100000, max_validation=100, seed=0,
                            batch=1 << 16, device=None):
    """open3d.registration_ransac_based_on_feature_matching (Open3D 0.7 semantics):
    iterate: ransac_n random source points, each paired with its nearest target FEATURE -> edge-length checker -> rigid fit ->
    distance checker; the first `max_validation` iterations that pass (in iteration order, among at most `max_iteration`) are
    scored -- fitness = share of source points with a target POINT inside max_correspondence_distance after the transform,
    rmse over those -- and the best (fitness, then rmse) wins.  Iterations run on the GPU in batches of `batch`.
    -> dict(transformation f64[4,4], fitness, inlier_rmse, correspondence_set i64[k,2], iterations, validations)."""
    lib = _lib.load()
    dev = _dev(device)
    src, tgt = _f32(source_points, dev), _f32(target_points, dev)
    Ns, Nt = src.shape[0], tgt.shape[0]
    ident = dict(transformation=np.eye(4), fitness=0.0, inlier_rmse=0.0, correspondence_set=np.zeros((0, 2), np.int64),
                 iterations=0, validations=0)
    if Ns < ransac_n or Nt < ransac_n or max_correspondence_distance <= 0:
        return ident
    nn = feature_nn(source_desc, target_desc, device=dev)
    st = ops._stream(dev)
    chosen, it0, max_validation = [], 0, int(max_validation)
    while it0 < max_iteration and sum(c.shape[0] for c in chosen) < max_validation:
        H = int(min(batch, max_iteration - it0))
        T = torch.empty((H, 12), dtype=torch.float32, device=dev)
        valid = torch.empty((H,), dtype=torch.uint8, device=dev)
        rc = lib.d3f_ransac_hypotheses(src.data_ptr(), Ns, tgt.data_ptr(), Nt, nn.data_ptr(), int(ransac_n),
                                       float(edge_similarity or 0.0), float(checker_distance or 0.0), int(seed), int(it0), H,
                                       T.data_ptr(), valid.data_ptr(), st)
        _lib.check(rc, "ransac_hypotheses")
        keep = torch.nonzero(valid, as_tuple=False).reshape(-1)        # (plumbing: index selection of the passing rows)
        if keep.numel():
            chosen.append(T.index_select(0, keep))
        it0 += H
    if not chosen:
        ident["iterations"] = it0
        return ident
    Tv = torch.cat(chosen, 0)[:max_validation].contiguous()
    V = Tv.shape[0]
    grid = ops.NeighborGrid(tgt, ops.as_lens([Nt], dev), float(max_correspondence_distance))
    count = torch.empty((V,), dtype=torch.int32, device=dev)
    sumd2 = torch.empty((V,), dtype=torch.int64, device=dev)
    rc = lib.d3f_neighbor_grid_score(grid.mem.data_ptr(), grid.nbytes, Nt, src.data_ptr(), Ns, Tv.data_ptr(), V,
                                     float(max_correspondence_distance), count.data_ptr(), sumd2.data_ptr(), None, st)
    _lib.check(rc, "neighbor_grid_score")
    cnt = count.cpu().numpy().astype(np.int64)
    sd2 = sumd2.cpu().numpy().astype(np.float64) / 4294967296.0
    rmse = np.sqrt(sd2 / np.maximum(cnt, 1))
    # best fitness, then lowest rmse, then earliest iteration
    order = np.lexsort((np.arange(V), rmse, -cnt))
    b = int(order[0])
    Tb = Tv[b:b + 1].contiguous()
    nearest = torch.empty((Ns,), dtype=torch.int32, device=dev)
    rc = lib.d3f_neighbor_grid_score(grid.mem.data_ptr(), grid.nbytes, Nt, src.data_ptr(), Ns, Tb.data_ptr(), 1,
                                     float(max_correspondence_distance), count[:1].data_ptr(), sumd2[:1].data_ptr(),
                                     nearest.data_ptr(), st)
    _lib.check(rc, "neighbor_grid_score")
    near = nearest.cpu().numpy()
    sel = np.nonzero(near >= 0)[0]
    M = np.eye(4)
    M[:3, :4] = Tb.cpu().numpy().reshape(3, 4).astype(np.float64)
    return dict(transformation=M, fitness=float(cnt[b]) / Ns, inlier_rmse=float(rmse[b]),
                correspondence_set=np.stack([sel, near[sel]], 1).astype(np.int64), iterations=it0, validations=V)
