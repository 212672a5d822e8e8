#!/usr/bin/env python3
"""Wall time of the REFERENCE'S OWN PYTHON on one benchmark fragment -- a clearly labelled second CPU row.

The reference's model code (kernels/convolution_ops.py, models/network_blocks.py, models/D3Feat.py, datasets/common.py; imported
UNMODIFIED from /root/reference exactly as tools/make_golden_network.py does) is executed under oracle/tf_eager -- a numpy float32
eager stand-in for TensorFlow 1.12, since TensorFlow itself is not installable here -- on bench.py's synthetic 3DMatch fragment
(seed 0, 300 k raw points, self-pair), with the custom ops served by the reference's C++ (oracle/_ref).  This is NOT TensorFlow's CPU
runtime (no graph optimiser, no Eigen thread pool: numpy + its BLAS) and it runs in the BUILD CONTAINER (the GPU box has no reference
checkout), so it is a different host than bench.py's cpu_baseline: read it as "the reference's Python, line by line, on numpy".

    python tools/time_reference_python.py > profiles/r05_reference_python_timing.json
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_golden_network as mg  # noqa: E402


def main():
    tf = mg.setup_imports()
    import tempfile
    os.chdir(tempfile.mkdtemp(prefix="d3f_refpy_"))
    import kernels.convolution_ops as conv_ops
    import models.network_blocks as network_blocks
    import models.D3Feat as d3feat
    import datasets.common as common
    from utils.config import Config
    from oracle.clib import COracle
    sys.path.insert(0, ROOT)
    from d3feat_amd.utils.synthetic import room_fragment
    mods = (conv_ops, network_blocks, d3feat, common)
    cfg = Config()
    cfg.load(os.path.join(mg.REF, "results", "Log_contraloss"))
    raw = room_fragment(0, n_raw=300000, edge=1.68)
    co = COracle()
    t0 = time.perf_counter()
    sub = co.grid_subsampling(raw, 0.03)
    t_sub = time.perf_counter() - t0
    limits = [42, 42, 46, 51, 49]                    # the calibrated limits of bench.py's pool
    mg.ROWS = 8
    times = []
    for rep in range(2):
        t0 = time.perf_counter()
        out = mg.run_case(tf, mods, cfg, [sub, sub], limits, tag="timing%d" % rep)
        times.append(time.perf_counter() - t0)
    import multiprocessing
    print(json.dumps({"what": "the reference's own Python (unmodified) under oracle/tf_eager (numpy float32 eager stand-in for TF 1.12) on "
                              "bench.py's fragment seed 0, self-pair, limits 42/42/46/51/49: tf_descriptor_input (custom ops = the "
                              "reference's C++) + assemble_FCNN_blocks; build container, NOT the GPU box's host",
                      "points_per_cloud": int(len(sub)), "stage0_subsample_s": round(t_sub, 3),
                      "pyramid_plus_network_s": [round(t, 2) for t in times], "fragments_per_s": round(1.0 / (min(times) + t_sub), 4),
                      "host_cores": multiprocessing.cpu_count(), "numpy": np.__version__,
                      "descriptor_rows": int(out["descriptors"].shape[0])}))


if __name__ == "__main__":
    main()
