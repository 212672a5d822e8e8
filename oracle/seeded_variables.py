"""TEST INFRASTRUCTURE ONLY (oracle).  Never imported by d3feat_amd/.

Name-keyed seeded variable values for the golden network fixtures (tools/make_golden_network.py).

The network has 14.1 M parameters (56 MB): too large to commit next to the outputs the reference's code produced from
them.  The large `weights` tensors are therefore *derived from their checkpoint name and a seed* -- same distribution
as the reference's initialiser (models/network_blocks.py:37-41: truncated normal, stddev sqrt(2 / shape[-1]), rounded
to three decimals) -- so that the fixture generator (which feeds them to the reference's code through the variable hook
of oracle/tf_eager) and the tests (which feed them to oracle/network_np.py and to the HIP path) obtain the same tensors;
the fixture carries a sha256 per regenerated tensor, checked by `resolve()`.  Small variables (batch-norm statistics,
kernel points) are stored in the fixture itself.
"""
import hashlib
import zlib

import numpy as np


def _rng(name, seed):
    return np.random.Generator(np.random.PCG64([zlib.crc32(name.encode()) & 0xFFFFFFFF, int(seed)]))


def seeded_weights(name, shape, seed):
    rng = _rng(name, seed)
    out = rng.standard_normal(tuple(shape))
    bad = np.abs(out) > 2.0
    while np.any(bad):
        out[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(out) > 2.0
    w = (out * np.sqrt(2.0 / shape[-1])).astype(np.float32)
    return (np.round(w * np.float32(1000)) / np.float32(1000)).astype(np.float32)


def seeded_bn(name, shape, seed):
    """Non-trivial inference batch-norm statistics (identity statistics hide bugs)."""
    rng = _rng(name, seed)
    c = int(shape[0])
    leaf = name.rsplit("/", 1)[-1]
    if leaf == "gamma":
        return (1.0 + 0.2 * rng.standard_normal(c)).astype(np.float32)
    if leaf == "beta":
        return (0.1 * rng.standard_normal(c)).astype(np.float32)
    if leaf == "moving_mean":
        return (0.1 * rng.standard_normal(c)).astype(np.float32)
    if leaf == "moving_variance":
        return (0.5 + rng.random(c)).astype(np.float32)
    raise KeyError(name)


def digest(a):
    a = np.ascontiguousarray(a, np.float32)
    return hashlib.sha256(a.tobytes()).hexdigest()[:16]


def resolve(fixture, extra_sources=()):
    """name -> float32 array for every variable the fixture's run used.

    fixture: the loaded npz.  Keys 'var/<name>' hold stored values; 'varspec' is a JSON list of
    [name, shape, kind, digest] with kind in {'stored', 'seeded', 'external'}; 'external' tensors come from
    `extra_sources` (npz-like objects keyed by name with '/' -> '__'), e.g. tests/golden/kitti_epoch61_weights.npz."""
    import json
    seed = int(fixture["seed"])
    W = {}
    for name, shape, kind, dg in json.loads(str(fixture["varspec"])):
        if kind == "stored":
            v = np.ascontiguousarray(fixture["var/" + name], np.float32)
        elif kind == "seeded":
            v = seeded_weights(name, shape, seed)
        elif kind == "external":
            key = name.replace("/", "__")
            v = None
            for src in extra_sources:
                if key in src.files:
                    v = np.ascontiguousarray(src[key], np.float32)
                    break
            if v is None:
                raise KeyError("external variable %s not found in the supplied sources" % name)
        else:
            raise ValueError(kind)
        if list(v.shape) != list(shape) or digest(v) != dg:
            raise ValueError("variable %s (%s): regenerated tensor differs from the one the fixture was made with" % (name, kind))
        W[name] = v
    return W
