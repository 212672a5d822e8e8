#!/usr/bin/env python3
"""PLY fragments -> D3Feat keypoints / descriptors / scores on an MI355X: the descriptor-extraction half of the reference's
demo_registration.py (its RegTester.generate_descriptor, :150-170) and of utils/tester.py:196-229, with the reference's
output files (one .npz per cloud: keypts [n,3], features [n,32], scores [n,1], rows in ASCENDING score order as the
reference's np.argsort leaves them).  The RANSAC / visualisation half of the demo (open3d) is out of scope.

    python tools/extract_descriptors.py cloud_bin_0.ply cloud_bin_1.ply [--snapshot results/Log_contraloss/snapshots/snap-54]
                                        [--config results/Log_contraloss] [--out DIR]

Without --snapshot the network runs with seeded random-init weights (the released blobs are not in the public checkout).
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from d3feat_amd import tf_custom_ops as tfo  # noqa: E402
from d3feat_amd.datasets.common import FragmentDataset  # noqa: E402
from d3feat_amd.engine import FragmentEngine  # noqa: E402
from d3feat_amd.models.variables import build_variables  # noqa: E402
from d3feat_amd.utils.config import Config, threedmatch_config  # noqa: E402
from d3feat_amd.utils.ply import read_ply_xyz  # noqa: E402
from d3feat_amd.utils.tf_checkpoint import load_checkpoint  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("clouds", nargs="+")
    ap.add_argument("--snapshot", default=None, help="checkpoint prefix (snap-N without extension)")
    ap.add_argument("--config", default=None, help="folder holding the model's parameters.txt")
    ap.add_argument("--out", default=".")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    if args.config:
        cfg = Config()
        cfg.load(args.config)
    else:
        cfg = threedmatch_config()
    dl0 = cfg.first_subsampling_dl
    weights = load_checkpoint(args.snapshot) if args.snapshot else build_variables(cfg, seed=42).values
    raws = [torch.from_numpy(read_ply_xyz(p)).to(dev) for p in args.clouds]
    # the reference voxelises with open3d before the pipeline (demo_registration.py:23-24); here its own grid subsampler does
    subs = [tfo.grid_subsampling(r, dl0).cpu().numpy() for r in raws]
    # neighbourhood limits as init_test_input_pipeline calibrates them (datasets/common.py:776-857)
    ds = FragmentDataset(subs, ids=[os.path.basename(p) for p in args.clouds])
    ds.init_test_input_pipeline(cfg)
    engine = FragmentEngine(cfg, weights, ds.neighborhood_limits, raw_cap=int(max(len(r) for r in raws) * 1.05) + 1024,
                            n0_cap=int(max(len(s) for s in subs) * 1.3) + 1024, slots=min(4, len(raws)), device=dev,
                            mirror_self_pair=True)
    os.makedirs(args.out, exist_ok=True)
    S = len(engine.slots)
    pending = [None] * S

    def save(i, out):
        pts, desc, score = (t.cpu().numpy() for t in out)
        n = pts.shape[0] // 2                                  # first cloud of the stacked self-pair
        order = np.argsort(score[:n], axis=0).squeeze()        # demo_registration.py:161-162
        name = os.path.join(args.out, os.path.basename(args.clouds[i]).replace(".ply", ""))
        np.savez_compressed(name, keypts=pts[:n][order], features=desc[:n][order], scores=score[:n][order])
        print("%s: %d keypoints -> %s.npz" % (args.clouds[i], n, name))

    for i, raw in enumerate(raws):
        k = i % S
        if pending[k] is not None:
            save(pending[k], engine.fetch(k))
        engine.submit(k, raw)
        pending[k] = i
    for k in range(S):
        if pending[k] is not None:
            save(pending[k], engine.fetch(k))
    if engine.fallbacks:
        print("(%d fragment(s) took the op-by-op path)" % engine.fallbacks)


if __name__ == "__main__":
    main()
