"""Kernel point dispositions (the 15 points of each KPConv).

Role of the reference's kernels/kernel_points.py: `kernel_point_optimization_debug` (:41-181) spreads points
in the unit ball by gradient descent on a repulsion + attraction potential, `load_kernels` (:184-280) caches the
best of 100 tries as kernels/dispositions/k_015_center.ply, then scales, randomly rotates and jitters it per
layer.  At inference the points come from the checkpoint (`kernel_points` variables), so this module is only an
input provider for random-weight runs.  Differences, on purpose: the RNG is an explicit seeded
numpy Generator (the reference uses the unseeded global state, so its dispositions are not reproducible,
SURVEY.md §2 row 7), and the disposition cache lives next to this file as .npy.
"""
import os

import numpy as np

_CACHE_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'dispositions')


def kernel_point_optimization_debug(radius, num_points, num_kernels=1, dimension=3, fixed='center', ratio=1.0,
                                    verbose=0, rng=None):
    """Potential minimisation of kernel_points.py:41-181: repulsion 1/d between points, attraction 5*|x|^2 to the
    centre, clipped gradient steps with decaying step size, stop when gradient norms stabilise.
    -> (points [num_kernels, num_points, dimension] scaled to `radius`, saved gradient norms)."""
    rng = rng or np.random.default_rng(0)
    radius0, diameter0 = 1.0, 2.0
    moving_factor, decay, thresh, clip = 1e-2, 0.9995, 1e-5, 0.05 * radius0
    # rejection-sample starting points inside the ball of radius sqrt(0.5)
    need = num_kernels * num_points
    kp = np.zeros((0, dimension))
    while kp.shape[0] < need:
        cand = rng.random((need, dimension)) * diameter0 - radius0
        kp = np.vstack((kp, cand[np.sum(cand ** 2, axis=1) < 0.5 * radius0 * radius0]))
    kp = kp[:need].reshape((num_kernels, num_points, dimension))
    if fixed == 'center':
        kp[:, 0, :] = 0
    if fixed == 'verticals':
        kp[:, :3, :] = 0
        kp[:, 1, -1] += 2 * radius0 / 3
        kp[:, 2, -1] -= 2 * radius0 / 3
    saved = np.zeros((10000, num_kernels))
    old = np.zeros((num_kernels, num_points))
    first_free = {'center': 1, 'verticals': 3}.get(fixed, 0)
    for it in range(10000):
        diff = kp[:, :, None, :] - kp[:, None, :, :]
        d2 = np.sum(diff ** 2, axis=-1)
        # diff[k,i,j] = p_i - p_j; summing over i gives d/dp_j of sum 1/|p_i - p_j|  (kernel_points.py:109-113)
        inter = np.sum(diff / (np.power(d2[..., None], 1.5) + 1e-6), axis=1)
        grads = inter + 10 * kp
        if fixed == 'verticals':
            grads[:, 1:3, :-1] = 0
        gn = np.sqrt(np.sum(grads ** 2, axis=-1) + 1e-12)
        saved[it, :] = np.max(gn, axis=1)
        if np.max(np.abs(old[:, first_free:] - gn[:, first_free:])) < thresh:
            break
        old = gn
        step = np.minimum(moving_factor * gn, clip)
        if fixed in ('center', 'verticals'):
            step[:, 0] = 0
        kp -= step[..., None] * grads / (gn + 1e-6)[..., None]
        moving_factor *= decay
    r = np.sqrt(np.sum(kp ** 2, axis=-1) + 1e-12)
    kp *= ratio / np.mean(r[:, 1:])
    return kp * radius, saved


def load_kernels(radius, num_kpoints, num_kernels, dimension, fixed, rng=None, num_tries=100):
    """kernel_points.py:184-280: best-of-`num_tries` unit disposition (cached), then per kernel: scale by `radius`,
    random rotation, N(0, 0.01*radius) jitter.  -> [num_kernels, num_kpoints, dimension]."""
    rng = rng or np.random.default_rng(0)
    if dimension not in (2, 3):
        raise ValueError('Unsupported dimpension of kernel : ' + str(dimension))
    name = 'k_{:03d}_{:s}{:s}.npy'.format(num_kpoints, fixed, '_2D' if dimension == 2 else '')
    path = os.path.join(_CACHE_DIR, name)
    if os.path.exists(path):
        original = np.load(path)
    else:
        pts, gnorms = kernel_point_optimization_debug(1.0, num_kpoints, num_kernels=num_tries, dimension=dimension,
                                                      fixed=fixed, rng=np.random.default_rng(2018))
        last = np.max(np.where(gnorms.sum(1) > 0)[0])
        original = pts[np.argmin(gnorms[last, :])]
        try:
            os.makedirs(_CACHE_DIR, exist_ok=True)
            np.save(path, original)
        except OSError:
            pass
    if dimension == 2:
        return original
    if fixed == 'verticals':
        th = rng.random(num_kernels) * 2 * np.pi
        c, s = np.cos(th), np.sin(th)
        R = np.zeros((num_kernels, 3, 3))
        R[:, 0, 0], R[:, 1, 1], R[:, 2, 2], R[:, 0, 1], R[:, 1, 0] = c, c, 1, s, -s
        return np.matmul(radius * original[None], R)
    u = np.ones((num_kernels, 3))
    v = np.ones((num_kernels, 3))
    wrong = np.abs(np.sum(u * v, axis=1)) > 0.99
    while np.any(wrong):
        nu = rng.random((num_kernels, 3)) * 2 - 1
        nu /= (np.linalg.norm(nu, axis=1) + 1e-9)[:, None]
        nv = rng.random((num_kernels, 3)) * 2 - 1
        nv /= (np.linalg.norm(nv, axis=1) + 1e-9)[:, None]
        u[wrong], v[wrong] = nu[wrong], nv[wrong]
        wrong = np.abs(np.sum(u * v, axis=1)) > 0.99
    v -= np.sum(u * v, axis=1)[:, None] * u
    v /= (np.linalg.norm(v, axis=1) + 1e-9)[:, None]
    R = np.stack((u, v, np.cross(u, v)), axis=-1)
    kernels = np.matmul(radius * original[None], R)
    return kernels + rng.normal(scale=radius * 0.01, size=kernels.shape)


def create_kernel_points(radius, num_kpoints, num_kernels, dimension, fixed, rng=None):
    """kernels/convolution_ops.py:26-33 -> load_kernels."""
    return load_kernels(radius, num_kpoints, num_kernels, dimension, fixed, rng=rng)
