"""TEST INFRASTRUCTURE ONLY (oracle).  Never imported by d3feat_amd/.

numpy restatement of the downstream matching the reference performs on the descriptors:
  * build_correspondence            geometric_registration/evaluate.py:11-27, line by line;
  * ransac_feature_matching         the algorithm of open3d.registration_ransac_based_on_feature_matching as the reference
                                    calls it (evaluate.py:93-99, demo_registration.py:184-192).  Open3D 0.7.0 (environment.yml:106)
                                    is third-party code absent from /root/reference and its random sampling is unspecified;
                                    the published algorithm (registration/Registration.cpp of Open3D 0.7: sample ransac_n source
                                    points, nearest target feature each, checkers that need no alignment, rigid fit by
                                    Umeyama/Kabsch without scaling, checkers that need the alignment, evaluation by nearest
                                    target point within max_correspondence_distance, best fitness then rmse, validation
                                    budget) is restated here with a counter-based sampler shared with the GPU path.
PARITY STATUS: unpinned by the reference (it ships no test or golden vector for this step and Open3D cannot be installed).
"""
import numpy as np

M64 = (1 << 64) - 1


def splitmix(x):
    x = (x + 0x9E3779B97F4A7C15) & M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & M64
    return x ^ (x >> 31)


def draw(seed, it, d, n):
    r = splitmix(seed ^ splitmix((it * 64 + d) & M64))
    return int((r >> 11) % n)


def build_correspondence(source_desc, target_desc):
    """evaluate.py:11-27."""
    distance = np.sqrt(np.maximum(2 - 2 * (source_desc @ target_desc.T), 0))
    source_idx = np.argmin(distance, axis=1)
    target_idx = np.argmin(distance, axis=0)
    result = []
    for i in range(len(source_idx)):
        if target_idx[source_idx[i]] == i:
            result.append([i, source_idx[i]])
    return np.array(result).reshape(-1, 2)


def feature_nn(A, B):
    """argmin_j ||A_i - B_j||^2 in float64 (|a|^2 + |b|^2 - 2 a.b, evaluated in row blocks: no [n, m, C] temporary)."""
    A, B = A.astype(np.float64), B.astype(np.float64)
    b2 = (B * B).sum(1)
    idx, val = np.empty(len(A), np.int64), np.empty(len(A), np.float64)
    for a in range(0, len(A), 2048):
        blk = A[a:a + 2048]
        d2 = np.maximum((blk * blk).sum(1)[:, None] + b2[None, :] - 2.0 * (blk @ B.T), 0.0)
        idx[a:a + 2048] = d2.argmin(1)
        val[a:a + 2048] = d2[np.arange(len(blk)), idx[a:a + 2048]]
    return idx, val


def kabsch(s, t):
    """Rigid transform (no scaling) minimising sum |R s + tr - t|^2 (Umeyama / Kabsch by SVD)."""
    ms, mt = s.mean(0), t.mean(0)
    H = (s - ms).T @ (t - mt)
    U, _, Vt = np.linalg.svd(H)
    D = np.diag([1.0, 1.0, np.sign(np.linalg.det(Vt.T @ U.T))])
    R = Vt.T @ D @ U.T
    return R, mt - R @ ms


def hypothesis(src, tgt, nn, n, edge_sim, dist_thr, seed, it):
    si = [draw(seed, it, d, len(src)) for d in range(n)]
    if len(set(si)) < n:
        return None
    ti = [int(nn[i]) for i in si]
    s, t = src[si].astype(np.float64), tgt[ti].astype(np.float64)
    if edge_sim and edge_sim > 0:
        for a in range(n):
            for b in range(a + 1, n):
                ds, dt = np.linalg.norm(s[a] - s[b]), np.linalg.norm(t[a] - t[b])
                if ds < dt * edge_sim or dt < ds * edge_sim:
                    return None
    R, tr = kabsch(s, t)
    if dist_thr and dist_thr > 0:
        if (np.linalg.norm(s @ R.T + tr - t, axis=1) > dist_thr).any():
            return None
    return R, tr


def evaluate(src, tgt, R, tr, radius):
    p = src.astype(np.float64) @ R.T + tr
    d2 = ((p[:, None, :] - tgt[None, :, :].astype(np.float64)) ** 2).sum(-1)
    j = d2.argmin(1)
    m = d2[np.arange(len(p)), j]
    inl = m < radius * radius
    cnt = int(inl.sum())
    return cnt, (np.sqrt(m[inl].sum() / cnt) if cnt else 0.0), np.stack([np.nonzero(inl)[0], j[inl]], 1)


def ransac_feature_matching(src, tgt, src_desc, tgt_desc, radius, ransac_n=4, edge_similarity=0.9, checker_distance=None,
                            max_iteration=100000, max_validation=100, seed=0):
    nn, _ = feature_nn(src_desc, tgt_desc)
    best, vals, it = None, 0, 0
    while it < max_iteration and vals < max_validation:
        h = hypothesis(src, tgt, nn, ransac_n, edge_similarity, checker_distance, seed, it)
        if h is not None:
            vals += 1
            cnt, rmse, corr = evaluate(src, tgt, h[0], h[1], radius)
            if best is None or cnt > best[0] or (cnt == best[0] and rmse < best[1]):
                best = (cnt, rmse, h, corr, it)
        it += 1
    if best is None:
        return dict(transformation=np.eye(4), fitness=0.0, inlier_rmse=0.0, correspondence_set=np.zeros((0, 2), np.int64),
                    iterations=it, validations=0)
    M = np.eye(4)
    M[:3, :3], M[:3, 3] = best[2]
    return dict(transformation=M, fitness=best[0] / len(src), inlier_rmse=best[1], correspondence_set=best[3], iterations=it,
                validations=vals, best_iteration=best[4])
