"""Seeded synthetic inputs shaped like the reference's workloads -- host-side numpy only.

    room_fragment(seed)       config #2: a 3DMatch-like indoor fragment (surfaces, not volumes): points on the six
                              faces of a box plus three spheres standing on the floor, Gaussian jitter; ~300k raw
                              points that grid-subsample to ~30k at 0.03 m.
    lidar_sweep(seed)         config #4a: a KITTI-like 64-ring sweep over a ground plane with boxes.
"""
import numpy as np


def room_fragment(seed=0, n_raw=300000, edge=1.68, jitter=0.002):
    rng = np.random.Generator(np.random.PCG64(seed))
    ex, ey, ez = edge * (1.0 + 0.05 * rng.random()), edge * (1.0 + 0.05 * rng.random()), edge * 0.8
    n_box = int(n_raw * 0.88)
    face = rng.integers(0, 6, n_box)
    uv = rng.random((n_box, 2))
    pts = np.empty((n_box, 3))
    ax = face // 2
    side = face % 2
    ext = np.array([ex, ey, ez])
    for a in range(3):
        m = ax == a
        o = [d for d in range(3) if d != a]
        pts[m, a] = side[m] * ext[a]
        pts[m, o[0]] = uv[m, 0] * ext[o[0]]
        pts[m, o[1]] = uv[m, 1] * ext[o[1]]
    n_sph = n_raw - n_box
    # sphere centres keep 0.5 m from the walls (rooms narrower than 1 m: on the centre line); same draws for edge >= 1
    centers = np.stack([rng.uniform(min(0.5, ex / 2), max(ex - 0.5, ex / 2), 3),
                        rng.uniform(min(0.5, ey / 2), max(ey - 0.5, ey / 2), 3), np.full(3, 0.3)], 1)
    which = rng.integers(0, 3, n_sph)
    d = rng.standard_normal((n_sph, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    sph = centers[which] + 0.3 * d
    allp = np.concatenate([pts, sph], 0)
    allp += rng.normal(scale=jitter, size=allp.shape)
    allp = allp[rng.permutation(allp.shape[0])]
    return allp.astype(np.float32)


def lidar_sweep(seed=0, n_raw=120000, rings=64):
    rng = np.random.Generator(np.random.PCG64(seed))
    per = n_raw // rings
    az = rng.uniform(-np.pi, np.pi, (rings, per))
    el = np.linspace(np.deg2rad(-24.8), np.deg2rad(2.0), rings)[:, None] + np.zeros((1, per))
    dirs = np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)], -1).reshape(-1, 3)
    h = 1.73
    t_ground = np.where(dirs[:, 2] < -1e-3, -h / np.minimum(dirs[:, 2], -1e-3), 120.0)
    t = np.minimum(t_ground, 80.0)
    # axis-aligned boxes (cars / walls): slab intersection
    nb = 24
    c = np.stack([rng.uniform(-50, 50, nb), rng.uniform(-50, 50, nb), np.full(nb, -h + 0.8)], 1)
    s = np.stack([rng.uniform(1, 4, nb), rng.uniform(1, 8, nb), rng.uniform(0.7, 2.5, nb)], 1)
    for i in range(nb):
        lo, hi = c[i] - s[i], c[i] + s[i]
        with np.errstate(divide='ignore', invalid='ignore'):
            t1, t2 = lo / dirs, hi / dirs
        tn = np.max(np.minimum(t1, t2), axis=1)
        tf = np.min(np.maximum(t1, t2), axis=1)
        hit = (tn < tf) & (tn > 1.0)
        t = np.where(hit & (tn < t), tn, t)
    pts = dirs * t[:, None]
    pts += rng.normal(scale=0.01, size=pts.shape)
    return pts.astype(np.float32)
