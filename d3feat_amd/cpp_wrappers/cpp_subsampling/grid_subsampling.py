"""`grid_subsampling.compute` -- the CPython extension of the reference (cpp_wrappers/cpp_subsampling/wrapper.cpp:58-286),
same keyword interface and return conventions, computed on the MI355X.

    compute(points, features=None, classes=None, sampleDl=0.1, method='barycenters', verbose=0)
      -> points | (points, features) | (points, classes) | (points, features, classes)

numpy in, numpy out (as the extension).  Argument handling restated from wrapper.cpp:70-276:
  * `method` must be 'barycenters' or 'voxelcenters' and is otherwise ignored (:86-90);
  * points (N,3) float32; features (N,d) float32; classes (N,) or (N,ldim) int32 (:100-172);
  * mismatching first dimensions raise RuntimeError (:175-190); classes are ALWAYS returned 2-D (:241-243);
  * an empty result raises RuntimeError("Error") (:225-229).
"""
import numpy as np
import torch

from ... import ops


def compute(points, features=None, classes=None, sampleDl=0.1, method="barycenters", verbose=0):
    if method not in ("barycenters", "voxelcenters"):
        raise RuntimeError("Error parsing method. Valid method names are \"barycenters\" and \"voxelcenters\" ")
    try:
        pts = np.ascontiguousarray(points, dtype=np.float32)
    except Exception:
        raise RuntimeError("Error converting input points to numpy arrays of type float32")
    if pts.ndim != 2 or pts.shape[1] != 3:
        raise RuntimeError("Wrong dimensions : points.shape is not (N, 3)")
    N = pts.shape[0]
    feat = cls = None
    if features is not None:
        try:
            feat = np.ascontiguousarray(features, dtype=np.float32)
        except Exception:
            raise RuntimeError("Error converting input features to numpy arrays of type float32")
        if feat.ndim != 2:
            raise RuntimeError("Wrong dimensions : features.shape is not (N, d)")
        if feat.shape[0] != N:
            raise RuntimeError("Wrong dimensions : features.shape is not (N, d)")
    if classes is not None:
        try:
            cls = np.ascontiguousarray(classes, dtype=np.int32)
        except Exception:
            raise RuntimeError("Error converting input classes to numpy arrays of type int32")
        if cls.ndim > 2:
            raise RuntimeError("Wrong dimensions : classes.shape is not (N,) or (N, d)")
        if cls.shape[0] != N:
            raise RuntimeError("Wrong dimensions : classes.shape is not (N,) or (N, d)")
        cls = cls.reshape(N, -1)
    if N == 0:
        raise RuntimeError("Error")
    dev = torch.device("cuda", torch.cuda.current_device())
    sub_p, _, sub_f, sub_c = ops.batch_grid_subsample(
        torch.from_numpy(pts).to(dev), [N], float(sampleDl),
        torch.from_numpy(feat).to(dev) if feat is not None else None,
        torch.from_numpy(cls).to(dev) if cls is not None else None)
    if sub_p.shape[0] < 1:
        raise RuntimeError("Error")
    res = [sub_p.cpu().numpy()]
    if feat is not None:
        res.append(sub_f.cpu().numpy())
    if cls is not None:
        res.append(sub_c.cpu().numpy())
    return res[0] if len(res) == 1 else tuple(res)
