"""Run as a subprocess with D3F_GS_SORT_MIN=1 (tests/test_gpu_preprocess.py): the capacity-mode subsampler then takes its
SORT form for every size, and must return bit for bit what the synchronous call (always the hash form) returns."""
import sys

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from d3feat_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
rng = np.random.default_rng(3)
cases = []
for B, n, dl, spread in [(1, 5000, 0.05, 1.0), (4, 20000, 0.03, 2.2), (8, 3000, 0.1, 1.0), (3, 40000, 0.02, 0.5), (2, 1, 0.05, 1.0),
                         (5, 700, 0.5, 1.0)]:
    lens = [max(1, int(n * f)) for f in rng.uniform(0.5, 1.0, B)]
    pts = [((rng.random((l, 3)) * spread) + rng.uniform(-3, 3, 3)).astype(np.float32) for l in lens]
    pts[0][: min(10, lens[0])] = pts[0][0]            # duplicates: a crowded voxel
    cases.append((np.concatenate(pts), lens, dl))
for pts, lens, dl in cases:
    P = torch.from_numpy(pts).to(dev)
    want_p, want_l, _, _ = ops.batch_grid_subsample(P, lens, dl)                       # hash form, one host sync
    cap = P.shape[0] + 1000                                                            # capacity > real size: exercises the tail
    Pc = torch.zeros((cap, 3), dtype=torch.float32, device=dev)
    Pc[: P.shape[0]] = P
    lens_dev = torch.tensor(lens, dtype=torch.int32, device=dev)
    got_p, got_l, st = ops.batch_grid_subsample_async(Pc, lens_dev, dl, cap, elem_cap=max(lens))
    torch.cuda.synchronize()
    stl = st.tolist()
    assert stl[1] == 0, stl
    m = stl[0]
    assert m == want_p.shape[0], (m, want_p.shape)
    assert torch.equal(got_l.cpu(), want_l.cpu())
    assert torch.equal(got_p[:m].cpu(), want_p.cpu()), "sort form differs from the hash form"
print("SORT-PATH-OK", len(cases))
