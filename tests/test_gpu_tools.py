"""The descriptor-extraction tool (tools/extract_descriptors.py = the reference demo's RegTester.generate_descriptor,
demo_registration.py:150-170) end to end on a slice of the reference's demo cloud: output files, shapes, ordering."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu


def test_extract_descriptors_tool(device, tmp_path, coracle):
    from d3feat_amd.utils.ply import write_ply
    head = np.load(os.path.join(GOLDEN, "demo_bin0_head.npy"))
    clouds = []
    for i, sl in enumerate((slice(0, 20000), slice(3000, 18000))):
        fn = str(tmp_path / ("cloud_%d.ply" % i))
        assert write_ply(fn, [head[sl]], ["x", "y", "z"])
        clouds.append(fn)
    out = tmp_path / "out"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "extract_descriptors.py"), *clouds, "--out", str(out)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    for i, sl in enumerate((slice(0, 20000), slice(3000, 18000))):
        z = np.load(str(out / ("cloud_%d.npz" % i)))
        want = coracle.grid_subsampling(head[sl], 0.03)
        assert z["keypts"].shape == want.shape and z["features"].shape == (len(want), 32) and z["scores"].shape == (len(want), 1)
        assert np.all(np.diff(z["scores"][:, 0]) >= 0)                         # ascending score order, as np.argsort leaves it
        assert np.abs(np.linalg.norm(z["features"], axis=1) - 1).max() < 1e-5
        # the keypoints are the subsampled cloud, permuted
        a = np.sort(z["keypts"].view([("x", "f4"), ("y", "f4"), ("z", "f4")]).ravel())
        b = np.sort(want.view([("x", "f4"), ("y", "f4"), ("z", "f4")]).ravel())
        assert np.array_equal(a, b)
