"""`open3d` (0.7 API) as the reference's inference callers see it -- exactly the symbols demo_registration.py,
datasets/ThreeDMatch.py (test split) and geometric_registration/evaluate.py touch (last row), backed by
d3feat_amd.  NOT Open3D.

    read_point_cloud, voxel_down_sample        demo_registration.py:23-24, datasets/ThreeDMatch.py:348-349
    PointCloud, Vector3dVector (also utility.) demo_registration.py:228-229, evaluate.py:69-70
    registration.Feature                       :226,235
    registration_ransac_based_on_feature_matching, TransformationEstimationPointToPoint,
    CorrespondenceCheckerBasedOnEdgeLength / Distance, RANSACConvergenceCriteria      :184-192, evaluate.py:93-99
    estimate_normals, draw_geometries, geometry.create_mesh_sphere, set_verbosity_level, VerbosityLevel   (display: no-ops)

voxel_down_sample: Open3D's voxel grid (origin at min - voxel/2, hash-map order) is third-party arithmetic that nothing in
the reference pins; as BASELINE.json's north star directs, stage 0 runs the reference's OWN grid subsampler
(tf_custom_ops/tf_subsampling/grid_subsampling/grid_subsampling.cpp:5-97) on the GPU instead -- same voxel size, barycentres,
its bit-exact order.  The RANSAC entry point is d3feat_amd.registration (mutual-free feature matching + batched hypotheses
on the GPU, same checkers / criteria semantics as Open3D 0.7's registration_ransac_based_on_feature_matching).
"""
import copy as _copy

import numpy as np


class VerbosityLevel:
    Error, Warning, Info, Debug, Always = 0, 1, 2, 3, 4


def set_verbosity_level(level):
    pass


def Vector3dVector(a):
    a = np.asarray(a, dtype=np.float64)
    if a.ndim != 2 or a.shape[1] != 3:
        raise RuntimeError("Vector3dVector: expected an array of shape (n, 3), got %s" % (a.shape,))
    return np.ascontiguousarray(a)


class PointCloud:
    def __init__(self):
        self._points = np.zeros((0, 3), np.float64)
        self.colors = np.zeros((0, 3), np.float64)
        self.normals = np.zeros((0, 3), np.float64)

    @property
    def points(self):
        return self._points

    @points.setter
    def points(self, v):
        self._points = Vector3dVector(v)

    def has_points(self):
        return len(self._points) > 0

    def transform(self, T):
        T = np.asarray(T, np.float64)
        self._points = self._points @ T[:3, :3].T + T[:3, 3]
        return self

    def paint_uniform_color(self, c):
        self.colors = np.tile(np.asarray(c, np.float64), (len(self._points), 1))
        return self

    def __repr__(self):
        return "PointCloud with %d points." % len(self._points)


class TriangleMesh:
    def __init__(self, center=(0, 0, 0), radius=1.0):
        self.center, self.radius, self.color = np.asarray(center, np.float64), radius, None

    def translate(self, t):
        self.center = self.center + np.asarray(t, np.float64).reshape(-1)[:3]
        return self

    def paint_uniform_color(self, c):
        self.color = np.asarray(c, np.float64)
        return self


def read_point_cloud(filename, format="auto"):
    from d3feat_amd.utils.ply import read_ply_xyz
    pcd = PointCloud()
    pcd.points = read_ply_xyz(filename)
    return pcd


def voxel_down_sample(pcd, voxel_size):
    import torch
    from d3feat_amd import tf_custom_ops as tfo
    out = PointCloud()
    if len(pcd.points) == 0:
        return out
    dev = torch.device("cuda", torch.cuda.current_device())
    pts = torch.from_numpy(np.ascontiguousarray(pcd.points, dtype=np.float32)).to(dev)
    out.points = tfo.grid_subsampling(pts, float(voxel_size)).cpu().numpy()
    return out


def estimate_normals(pcd, search_param=None):
    return True      # display helper of the demo: nothing downstream of the descriptors reads normals


def draw_geometries(geometries, *a, **kw):
    print("[open3d compat] draw_geometries(%d geometries): no display in this environment" % len(geometries))


class _Feature:
    def __init__(self):
        self.data = np.zeros((0, 0), np.float64)

    def dimension(self):
        return self.data.shape[0]

    def num(self):
        return self.data.shape[1]


class TransformationEstimationPointToPoint:
    def __init__(self, with_scaling=False):
        self.with_scaling = bool(with_scaling)


class CorrespondenceCheckerBasedOnEdgeLength:
    def __init__(self, similarity_threshold=0.9):
        self.similarity_threshold = float(similarity_threshold)


class CorrespondenceCheckerBasedOnDistance:
    def __init__(self, distance_threshold):
        self.distance_threshold = float(distance_threshold)


class RANSACConvergenceCriteria:
    def __init__(self, max_iteration=1000, max_validation=1000):
        self.max_iteration, self.max_validation = int(max_iteration), int(max_validation)


class RegistrationResult:
    def __init__(self, transformation, fitness, inlier_rmse, correspondence_set):
        self.transformation, self.fitness, self.inlier_rmse = transformation, fitness, inlier_rmse
        self.correspondence_set = correspondence_set

    def __repr__(self):
        return ("RegistrationResult with fitness = %f, inlier_rmse = %f, and correspondence_set size of %d\n"
                "Access transformation to get result." % (self.fitness, self.inlier_rmse, len(self.correspondence_set)))


def registration_ransac_based_on_feature_matching(source, target, source_feature, target_feature, max_correspondence_distance,
                                                  estimation_method=None, ransac_n=4, checkers=(), criteria=None):
    from d3feat_amd import registration as reg
    est = estimation_method or TransformationEstimationPointToPoint(False)
    if est.with_scaling:
        raise NotImplementedError("TransformationEstimationPointToPoint(with_scaling=True) is not used by the reference")
    crit = criteria or RANSACConvergenceCriteria(100000, 100)
    edge = next((c.similarity_threshold for c in checkers if isinstance(c, CorrespondenceCheckerBasedOnEdgeLength)), None)
    dist = next((c.distance_threshold for c in checkers if isinstance(c, CorrespondenceCheckerBasedOnDistance)), None)
    r = reg.ransac_feature_matching(np.asarray(source.points), np.asarray(target.points), np.asarray(source_feature.data).T,
                                    np.asarray(target_feature.data).T, float(max_correspondence_distance), int(ransac_n),
                                    edge_similarity=edge, checker_distance=dist, max_iteration=crit.max_iteration,
                                    max_validation=crit.max_validation)
    return RegistrationResult(r["transformation"], r["fitness"], r["inlier_rmse"], r["correspondence_set"])


class _Namespace:
    pass


registration = _Namespace()
registration.Feature = _Feature
registration.registration_ransac_based_on_feature_matching = registration_ransac_based_on_feature_matching
registration.TransformationEstimationPointToPoint = TransformationEstimationPointToPoint
registration.CorrespondenceCheckerBasedOnEdgeLength = CorrespondenceCheckerBasedOnEdgeLength
registration.CorrespondenceCheckerBasedOnDistance = CorrespondenceCheckerBasedOnDistance
registration.RANSACConvergenceCriteria = RANSACConvergenceCriteria
Feature = _Feature

utility = _Namespace()
utility.Vector3dVector = Vector3dVector
utility.set_verbosity_level = set_verbosity_level
utility.VerbosityLevel = VerbosityLevel

geometry = _Namespace()
geometry.PointCloud = PointCloud
geometry.create_mesh_sphere = lambda radius=1.0, resolution=20: TriangleMesh(radius=radius)
geometry.voxel_down_sample = voxel_down_sample
geometry.estimate_normals = estimate_normals

visualization = _Namespace()
visualization.draw_geometries = draw_geometries

io = _Namespace()
io.read_point_cloud = read_point_cloud
