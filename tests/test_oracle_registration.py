"""CPU checks of the downstream-matching oracle (oracle/registration_np.py) and of the host half of the C ABI for it."""
import numpy as np


def test_build_correspondence_is_the_mutual_argmin():
    from oracle import registration_np as onp
    a = np.eye(4)[[0, 1, 2]]                      # three unit descriptors
    b = np.eye(4)[[2, 0, 3, 1]]
    assert onp.build_correspondence(a, b).tolist() == [[0, 1], [1, 3], [2, 0]]
    # b[2] is nobody's nearest; a one-sided nearest neighbour is dropped
    a2 = np.array([[1.0, 0, 0, 0], [0.99, 0.14, 0, 0]])
    a2 /= np.linalg.norm(a2, axis=1, keepdims=True)
    assert onp.build_correspondence(a2, np.eye(4)[[0]]).tolist() == [[0, 0]]


def test_kabsch_recovers_a_rigid_motion_and_never_reflects():
    from oracle import registration_np as onp
    rng = np.random.default_rng(0)
    s = rng.standard_normal((4, 3))
    q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    R = q * np.sign(np.linalg.det(q))
    t = np.array([0.3, -1.0, 2.0])
    Rg, tg = onp.kabsch(s, s @ R.T + t)
    assert np.allclose(Rg, R, atol=1e-10) and np.allclose(tg, t, atol=1e-10)
    flat = np.concatenate([rng.standard_normal((3, 2)), np.zeros((3, 1))], 1)        # planar sample: det must stay +1
    Rg, _ = onp.kabsch(flat, flat * np.array([1, 1, -1]))
    assert np.isclose(np.linalg.det(Rg), 1.0)


def test_sampler_is_shared_with_the_library():
    from d3feat_amd import _lib
    from oracle import registration_np as onp
    lib = _lib.load()
    for seed in (0, 5, 2 ** 40 + 17):
        for it in (0, 1, 255, 10 ** 6 + 3):
            for d in range(4):
                assert lib.d3f_ransac_draw(seed, it, d, 250) == onp.draw(seed, it, d, 250)
    assert lib.d3f_ransac_draw(1, 1, 0, 0) == -1


def test_oracle_ransac_finds_a_planted_motion():
    from oracle import registration_np as onp
    rng = np.random.default_rng(1)
    tgt = rng.random((120, 3))
    R, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    R *= np.sign(np.linalg.det(R))
    t = np.array([0.1, 0.2, -0.3])
    src = (tgt - t) @ R
    d = rng.standard_normal((120, 32))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    r = onp.ransac_feature_matching(src, tgt, d, d, 0.02, ransac_n=3, max_iteration=2000, max_validation=50, seed=3)
    assert r["fitness"] == 1.0 and np.allclose(r["transformation"][:3, :3], R, atol=1e-6)
