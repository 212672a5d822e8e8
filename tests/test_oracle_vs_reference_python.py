"""oracle/network_np.py against the reference's own Python at a LARGER size than the committed fixtures hold, live (only where the
reference checkout exists: this container, not the GPU box).  tools/make_golden_network.py --live-check N imports /root/reference's
kernels/convolution_ops.py, models/network_blocks.py, models/D3Feat.py, datasets/common.py unmodified under oracle/tf_eager, runs them on an
N-point crop of the demo cloud (self-pair), runs the restatement on the same inputs and weights, and prints the differences.  A subprocess:
the generator binds the module names `tensorflow`, `utils`, `datasets`, `models`, `kernels` to the stand-in / the reference."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.timeout(600)
def test_restatement_equals_the_reference_python_on_a_4000_point_pair():
    from oracle import clib
    if not os.path.isdir("/root/reference") or not clib.ref_available():
        pytest.skip("needs the reference checkout and oracle/_ref")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_golden_network.py"), "--live-check", "4000"],
                       capture_output=True, text=True, cwd=ROOT, timeout=580)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert res["rows"] == 8000 and res["blocks"] == 19
    assert res["block_rel_max"] <= 4e-6, res
    assert res["desc_max_abs"] <= 1e-5 and res["score_max_abs"] <= 1e-5, res
