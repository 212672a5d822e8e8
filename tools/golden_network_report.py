#!/usr/bin/env python3
"""Measured differences between the HIP path and the reference's own Python (tests/golden/network_*.npz), on the GPU of this box:
what tests/test_gpu_golden_network.py asserts, as numbers.    python tools/golden_network_report.py > profiles/rNN_golden_network_report.txt"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))


def main():
    import test_gpu_golden_network as T
    from d3feat_amd.engine import FragmentEngine
    from d3feat_amd.models.KPFCNN_model import KernelPointFCNN
    from oracle.golden_network import GoldenNetwork
    dev = torch.device("cuda", 0)
    for name in ("3dmatch", "kitti"):
        g = GoldenNetwork(name)
        got, restore = T._record_blocks()
        try:
            m = KernelPointFCNN(T._flat(g, dev), g.config(), weights=dict(g.W))
        finally:
            restore()
        d, s = m.out_features.cpu().numpy(), m.out_scores.cpu().numpy()
        print("%-8s HIP model on the reference's inputs: descriptors max |diff| %.3e   scores %.3e   (bar 1e-4)"
              % (name, np.abs(d - g.descriptors).max(), np.abs(s - g.scores).max()))
        for scope in g.block_order:
            if scope in got:
                rows, want = g.block(scope)
                have = got[scope].cpu().numpy()[rows]
                print("   %-34s max |diff| %.3e   (largest |value| %.3f)" % (scope, np.abs(have - want).max(), np.abs(want).max()))
    g = GoldenNetwork("3dmatch")
    eng = FragmentEngine(g.config(), dict(g.W), g.limits, n0_cap=1536, level_ratio=0.45, slots=1, device=dev, batch=1, stage0=False)
    p, d, s = (t.cpu().numpy() for t in eng.run(torch.from_numpy(g.clouds()[0]).to(dev)))
    print("3dmatch  FragmentEngine replay (HIP graph): descriptors max |diff| %.3e   scores %.3e   fallbacks %d"
          % (np.abs(d - g.descriptors).max(), np.abs(s - g.scores).max(), eng.fallbacks))


if __name__ == "__main__":
    main()
