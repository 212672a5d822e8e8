"""Data formats on either side of the hot path: the KITTI velodyne reader the reference's test
generators use, and the per-fragment result files its testers write.  Pure numpy; no device work here.
"""
import os

import numpy as np


def read_kitti_bin(path):
    """One velodyne sweep: float32 records (x, y, z, reflectance) -> xyz f32[n,3]  (datasets/KITTI.py:131, 277-278)."""
    raw = np.fromfile(path, dtype=np.float32)
    if raw.size % 4 != 0:
        raise ValueError("%s: %d float32 values, not a multiple of 4" % (path, raw.size))
    return np.ascontiguousarray(raw.reshape(-1, 4)[:, :3])


def read_kitti_records(path):
    """One velodyne sweep as RAW bytes + record layout for ops.decode_xyz_records (16-byte records x, y, z, reflectance)."""
    raw = np.fromfile(path, dtype=np.uint8)
    if raw.size % 16 != 0:
        raise ValueError("%s: %d bytes, not a multiple of 16" % (path, raw.size))
    return raw, dict(n=raw.size // 16, stride=16, offsets=(0, 4, 8), dtype='f4', big_endian=False)


def select_first_cloud(points, features, scores, first_len):
    """The keypoint selection of utils/tester.py:208-213 / demo_registration.py:158-164 for a stacked self-pair: rows of the
    FIRST cloud (the reference indexes them through in_batches[0][:-1]), in ASCENDING score order as its np.argsort leaves
    them (ties in index order: numpy's default quicksort is not stable, the reference's order among exactly equal scores is
    therefore unspecified -- a stable sort is used here).  -> (keypts [n,3], features [n,C], scores [n,1])"""
    points, features, scores = (np.asarray(a) for a in (points, features, scores))
    n = int(first_len)
    if not 0 <= n <= scores.shape[0]:
        raise ValueError("first_len %d outside [0, %d]" % (n, scores.shape[0]))
    s = scores[:n].reshape(n, -1)
    order = np.argsort(s[:, 0], kind="stable")
    return points[:n][order].astype(np.float32), features[:n][order].astype(np.float32), s[order].astype(np.float32)


def save_3dmatch_results(root, anc_id, points, features, scores, first_len):
    """The three files utils/tester.py:215-229 writes per fragment under `root` (its <path>/<descriptors|keypoints|scores>/
    <scene>/ layout): anc_id is the generator's id string '<scene>/.../cloud_bin_<k>.ply'.  Returns the three paths."""
    if isinstance(anc_id, bytes):
        anc_id = anc_id.decode("utf-8")
    scene = anc_id.split("/")[0]
    num_frag = int(anc_id.split("_")[-1][:-4])
    kp, feat, sc = select_first_cloud(points, features, scores, first_len)
    out = []
    for sub, name, arr in (("descriptors", "cloud_bin_%d.D3Feat" % num_frag, feat), ("keypoints", "cloud_bin_%d" % num_frag, kp),
                           ("scores", "cloud_bin_%d" % num_frag, sc)):
        d = os.path.join(root, sub, scene)
        os.makedirs(d, exist_ok=True)
        p = os.path.join(d, name)
        np.save(p, arr)
        out.append(p + ".npy")
    return out
