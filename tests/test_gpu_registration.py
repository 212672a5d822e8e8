"""Downstream matching on the GPU against oracle/registration_np.py: nearest descriptors and mutual
matches (geometric_registration/evaluate.py:11-27, integer results: equal), RANSAC hypotheses (same counter-based random
numbers: validity flags equal, transforms within 1e-4), and the whole registration on a synthetic pair with a known motion."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _unit(rng, n, c=32):
    x = rng.standard_normal((n, c)).astype(np.float32)
    return x / np.linalg.norm(x, axis=1, keepdims=True)


def _pair(seed, n=400, outliers=0.3, noise=0.003):
    """target keypoints on a room surface; source = the same points moved by a known rigid motion (+ noise), a share of the
    descriptors replaced by unrelated ones."""
    from d3feat_amd.utils.synthetic import room_fragment
    rng = np.random.default_rng(seed)
    tgt = room_fragment(seed, n_raw=20000, edge=2.0)[rng.permutation(20000)[:n]].astype(np.float32)
    ang = rng.uniform(-0.6, 0.6, 3)
    cx, sx, cy, sy, cz, sz = np.cos(ang[0]), np.sin(ang[0]), np.cos(ang[1]), np.sin(ang[1]), np.cos(ang[2]), np.sin(ang[2])
    R = (np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]) @
         np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]))
    t = rng.uniform(-0.5, 0.5, 3)
    # source -> target is (R, t): source = R^T (target - t)
    src = ((tgt.astype(np.float64) - t) @ R + rng.normal(scale=noise, size=tgt.shape)).astype(np.float32)
    perm = rng.permutation(n)
    src = src[perm]
    d_t = _unit(rng, n)
    d_s = d_t[perm] + 0.05 * rng.standard_normal((n, 32)).astype(np.float32)
    bad = rng.random(n) < outliers
    d_s[bad] = _unit(rng, int(bad.sum()))
    d_s /= np.linalg.norm(d_s, axis=1, keepdims=True)
    return src, tgt, d_s.astype(np.float32), d_t, R, t


@pytest.mark.parametrize("n,m", [(250, 250), (1000, 777), (5000, 5000), (3, 70000)])
def test_feature_nn_and_mutual_matches(device, n, m):
    from d3feat_amd import registration as reg
    from oracle import registration_np as onp
    rng = np.random.default_rng(n + m)
    A, B = _unit(rng, n), _unit(rng, m)
    idx, d2 = reg.feature_nn(A, B, return_d2=True, device=device)
    idx, d2 = idx.cpu().numpy(), d2.cpu().numpy()
    want, want_d2 = onp.feature_nn(A, B)
    same = idx == want
    # a different index is acceptable only inside fp32 rounding of the minimum
    assert np.all(same | (np.abs(d2 - want_d2) <= 1e-6)), (np.nonzero(~same)[0][:5])
    assert np.allclose(d2, want_d2, atol=2e-6)
    if n * m <= 25_000_000 and n > 3:
        got = reg.build_correspondence(A, B, device=device)
        ref = onp.build_correspondence(A.astype(np.float64), B.astype(np.float64))
        assert np.array_equal(got, ref)


def test_feature_nn_ties_take_the_lowest_index(device):
    from d3feat_amd import registration as reg
    rng = np.random.default_rng(0)
    B = _unit(rng, 600)
    B[300:] = B[:300]                   # every descriptor twice: the first copy must win
    A = B[:300][rng.permutation(300)]
    idx = reg.feature_nn(A, B, device=device).cpu().numpy()
    assert np.all(idx < 300) and np.allclose(B[idx], A)


def test_ransac_hypotheses_match_the_oracle(device):
    from d3feat_amd import _lib, ops
    from oracle import registration_np as onp
    lib = _lib.load()
    src, tgt, ds, dt, R, t = _pair(3, n=300)
    nn, _ = onp.feature_nn(ds, dt)
    H, seed = 4000, 12345
    for it in (0, 1, 77, 3999):
        for d in range(4):
            assert lib.d3f_ransac_draw(seed, it, d, 300) == onp.draw(seed, it, d, 300)
    st, tt = torch.from_numpy(src).to(device), torch.from_numpy(tgt).to(device)
    nnt = torch.from_numpy(nn.astype(np.int32)).to(device)
    T = torch.empty((H, 12), dtype=torch.float32, device=device)
    valid = torch.empty((H,), dtype=torch.uint8, device=device)
    _lib.check(lib.d3f_ransac_hypotheses(st.data_ptr(), 300, tt.data_ptr(), 300, nnt.data_ptr(), 4, 0.9, 0.05, seed, 0, H,
                                         T.data_ptr(), valid.data_ptr(), ops._stream(device)), "ransac_hypotheses")
    T, valid = T.cpu().numpy(), valid.cpu().numpy().astype(bool)
    nvalid = 0
    for it in range(H):
        h = onp.hypothesis(src, tgt, nn, 4, 0.9, 0.05, seed, it)
        if h is None:
            # a sample within rounding of a checker threshold may flip: tolerate nothing here, the margins are wide
            assert not valid[it], it
            continue
        assert valid[it], it
        nvalid += 1
        M = T[it].reshape(3, 4)
        assert np.abs(M[:, :3] - h[0]).max() < 1e-4 and np.abs(M[:, 3] - h[1]).max() < 1e-4, it
    assert nvalid > 5


@pytest.mark.parametrize("n,ransac_n", [(300, 4), (250, 3)])
def test_ransac_recovers_the_motion_and_equals_the_oracle(device, n, ransac_n):
    from d3feat_amd import registration as reg
    from oracle import registration_np as onp
    src, tgt, ds, dt, R, t = _pair(7 + n, n=n)
    kw = dict(ransac_n=ransac_n, edge_similarity=0.9, checker_distance=0.05, max_iteration=50000, max_validation=200, seed=5)
    got = reg.ransac_feature_matching(src, tgt, ds, dt, 0.05, device=device, batch=8192, **kw)
    want = onp.ransac_feature_matching(src, tgt, ds, dt, 0.05, **kw)
    assert got["validations"] == want["validations"]
    assert abs(got["fitness"] - want["fitness"]) < 1e-9 + 1.0 / n          # a point on the radius may flip
    assert np.abs(got["transformation"] - want["transformation"]).max() < 1e-3
    # and it is the right motion
    M = got["transformation"]
    assert np.abs(M[:3, :3] - R).max() < 0.03 and np.abs(M[:3, 3] - t).max() < 0.03
    assert got["fitness"] > 0.9
    c = got["correspondence_set"]
    moved = src[c[:, 0]].astype(np.float64) @ M[:3, :3].T + M[:3, 3]
    assert np.all(np.linalg.norm(moved - tgt[c[:, 1]], axis=1) < 0.05 + 1e-5)


def test_open3d_compat_entry_point(device):
    """The call of geometric_registration/evaluate.py:93-99 through the compat `open3d` module."""
    import os
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "compat"))
    try:
        import open3d
        src, tgt, ds, dt, R, t = _pair(21, n=250)
        s_pcd, t_pcd = open3d.PointCloud(), open3d.PointCloud()
        s_pcd.points = open3d.utility.Vector3dVector(src)
        t_pcd.points = open3d.utility.Vector3dVector(tgt)
        s_desc, t_desc = open3d.registration.Feature(), open3d.registration.Feature()
        s_desc.data, t_desc.data = ds.T, dt.T
        result = open3d.registration_ransac_based_on_feature_matching(
            s_pcd, t_pcd, s_desc, t_desc, 0.05, open3d.TransformationEstimationPointToPoint(False), 3,
            [open3d.CorrespondenceCheckerBasedOnEdgeLength(0.9), open3d.CorrespondenceCheckerBasedOnDistance(0.05)],
            open3d.RANSACConvergenceCriteria(50000, 1000))
        assert np.abs(result.transformation[:3, :3] - R).max() < 0.03 and result.fitness > 0.9
        assert "RegistrationResult" in repr(result)
    finally:
        sys.path.remove(os.path.join(ROOT, "compat"))
        for m in [k for k in sys.modules if k == "open3d" or k.startswith("open3d.")]:
            del sys.modules[m]
