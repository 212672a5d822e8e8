#!/usr/bin/env python3
"""tests/golden/kitti_epoch61_weights.npz: the REAL trained tensors the reference ships for its KITTI model --
results_kitti/Log_11011605/kernel_points/epoch61/*.npy (34 weight tensors, written by utils/trainer.py:503-557) -- as
one fixture, so that the GPU box (where /root/reference does not exist) can push trained weights of realistic scale
through the oracle and the HIP path (tests/test_gpu_real_weights.py).  The 10 trained kernel-point dispositions (*.ply)
are already in kitti_kernel_points.npz (tools/make_golden.py).  Batch-norm statistics and the deepest block's remaining
weights are not part of the reference's dump; the tests fill them with seeded values.

    python tools/make_golden_weights.py          # needs /root/reference; rewrites the fixture + its MANIFEST entry
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def main():
    from d3feat_amd.utils import tf_checkpoint
    dumps = tf_checkpoint.load_weight_dumps(os.path.join(REF, "results_kitti", "Log_11011605", "kernel_points", "epoch61"))
    w = {k.replace("/", "__"): v for k, v in dumps.items() if k.endswith("/weights")}
    path = os.path.join(OUT, "kitti_epoch61_weights.npz")
    np.savez_compressed(path, **w)
    man_p = os.path.join(OUT, "MANIFEST.json")
    man = json.load(open(man_p))
    man["files"]["kitti_epoch61_weights.npz"] = hashlib.sha256(open(path, "rb").read()).hexdigest()
    man.setdefault("generated_by_also", {})["kitti_epoch61_weights.npz"] = "tools/make_golden_weights.py"
    json.dump(man, open(man_p, "w"), indent=1)
    print("%d tensors, %.1f MB on disk" % (len(w), os.path.getsize(path) / 1e6))


if __name__ == "__main__":
    main()
