"""The four custom ops of the reference's tf_custom_ops/, same names and argument meaning
(datasets/common.py:34-37 loads them, :67-72 wraps them), running on the MI355X through libd3feat_amd.so.

    batch_ordered_neighbors(queries, supports, q_batches, s_batches, radius) -> int32[Nq, Kmax]
    ordered_neighbors(queries, supports, radius)                            -> int32[Nq, Kmax]   (pad = -1)
    batch_grid_subsampling(points, batches, dl)                              -> (float32[M,3], int32[B])
    grid_subsampling(points, dl)                                             -> float32[M,3]

Tensors are torch tensors on a GPU.  Output widths / lengths are data dependent, so each call synchronises once.
"""
import torch

from . import _lib, ops


def _neighbors(queries, supports, q_batches, s_batches, radius, pad_value, first_width=96):
    Nq = queries.shape[0]
    if Nq == 0:
        return torch.zeros((0, 0), dtype=torch.int32, device=queries.device)
    grid = ops.NeighborGrid(supports, s_batches, radius)
    width, cap = first_width, 192
    while True:
        out, status = grid.search(queries, q_batches, width, pad_value=pad_value, cap=cap)
        kmax, flags = status.tolist()
        if flags & _lib.ST_HIT_OVERFLOW and cap < _lib.NEIGHBOR_CAP:
            cap = _lib.NEIGHBOR_CAP          # more in-radius supports than the fast LDS budget: order them with the full one
            width = max(width, min(kmax, _lib.NEIGHBOR_CAP))
            continue
        ops.check_status(status, "batch_ordered_neighbors")
        if kmax <= width:
            return out[:, :kmax]
        width = kmax


def batch_ordered_neighbors(queries, supports, q_batches, s_batches, radius):
    """tf_custom_ops/tf_neighbors/tf_batch_neighbors.cpp:8-30 (BatchOrderedNeighbors).  Rows ascending by
    (d2, index); width = max neighbour count; pad = supports.shape[0]."""
    return _neighbors(queries, supports, q_batches, s_batches, radius, None)


def ordered_neighbors(queries, supports, radius):
    """tf_custom_ops/tf_neighbors/tf_neighbors.cpp:8-12 (OrderedNeighbors): one cloud, pad = -1
    (neighbors/neighbors.cpp:58-123)."""
    dev = queries.device
    return _neighbors(queries, supports, ops.as_lens([queries.shape[0]], dev), ops.as_lens([supports.shape[0]], dev),
                      radius, -1)


def batch_grid_subsampling(points, batches, dl):
    """tf_custom_ops/tf_subsampling/tf_batch_subsampling.cpp:8-20 (BatchGridSubsampling)."""
    sub_p, sub_l, _, _ = ops.batch_grid_subsample(points, batches, dl)
    return sub_p, sub_l


def grid_subsampling(points, dl):
    """tf_custom_ops/tf_subsampling/tf_subsampling.cpp:8-11 (GridSubsampling)."""
    sub_p, _, _, _ = ops.batch_grid_subsample(points, ops.as_lens([points.shape[0]], points.device), dl)
    return sub_p
