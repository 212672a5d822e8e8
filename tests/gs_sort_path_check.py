"""Run as a subprocess (tests/test_gpu_preprocess.py): the capacity-mode subsampler's SORT form (csrc/radix_sort.h + the
gs_sortkey / gs_heads / gs_runs / gs_emit kernels) and its ONE-WORKGROUP-PER-CLOUD form (csrc/gs_small.h, clouds of at most
16384 points: elem_points) must return bit for bit what the ORACLE (oracle/d3f_oracle.c, pinned to the reference's C++) and the
synchronous call (hash form) return: ragged stacks, duplicates, one-point clouds, sizes around the 4096-item sort tile, 100 clouds per stack, a grid that
needs 4 digit passes, a capacity tail -- and a grid too wide for the 32-bit sort key is REPORTED (D3F_ST_KEY_WIDTH), empty."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from d3feat_amd import _lib, ops  # noqa: E402
from oracle.clib import COracle  # noqa: E402

dev = torch.device("cuda", 0)
rng = np.random.default_rng(3)
co = COracle()
cases = []
#        B   points/cloud  dl    spread
for B, n, dl, spread in [(1, 5000, 0.05, 1.0), (4, 20000, 0.03, 2.2), (8, 3000, 0.1, 1.0), (3, 40000, 0.02, 0.5), (2, 1, 0.05, 1.0),
                         (5, 700, 0.5, 1.0), (1, 4096, 0.05, 1.0), (1, 4097, 0.05, 1.0), (2, 8192, 0.04, 1.5), (100, 300, 0.1, 1.0),
                         (2, 60000, 0.3, 150.0), (4, 300000, 0.03, 1.68), (1, 1, 0.03, 1.0), (8, 11000, 0.06, 1.68), (8, 16384, 0.1, 2.0),
                         (16, 2500, 0.12, 1.68), (8, 800, 0.24, 1.68), (3, 2048, 0.01, 1.0), (2, 5000, 0.3, 150.0)]:
    lens = [max(1, int(n * f)) for f in rng.uniform(0.5, 1.0, B)] if n > 1 else [1] * B
    pts = [((rng.random((l, 3)) * spread) + rng.uniform(-3, 3, 3)).astype(np.float32) for l in lens]
    pts[0][: min(10, lens[0])] = pts[0][0]            # duplicates: a crowded voxel
    cases.append((np.concatenate(pts), lens, dl))
nforms = 0
for pts, lens, dl in cases:
    P = torch.from_numpy(pts).to(dev)
    lens_np = np.asarray(lens, np.int32)
    ora_p, ora_l = co.batch_grid_subsampling(pts, lens_np, dl)                         # the oracle (plain C restatement)
    want_p, want_l, _, _ = ops.batch_grid_subsample(P, lens, dl)                       # hash form, one host sync
    cap = P.shape[0] + 1000                                                            # capacity > real size: exercises the tail
    Pc = torch.full((cap, 3), 1e30, dtype=torch.float32, device=dev)                   # (garbage beyond the real points)
    Pc[: P.shape[0]] = P
    lens_dev = torch.tensor(lens, dtype=torch.int32, device=dev)
    # the one-workgroup form is chosen by the caller's capacities: <= 16384 points and <= 5087 voxels per cloud
    mv = int(ora_l.max())
    small = max(lens) <= 16384 and mv <= 5087
    forms = 0
    for elem_points, elem_cap in ([(0, max(lens)), (max(lens), mv)] if small else [(0, max(lens))]):
        for rep in range(2):                                                           # the second call reuses the workspace as it was left
            got_p, got_l, st = ops.batch_grid_subsample_async(Pc, lens_dev, dl, cap, elem_cap=elem_cap, elem_points=elem_points)
            torch.cuda.synchronize()
            stl = st.tolist()
            assert stl[1] == 0, (stl, len(lens), len(pts), elem_points)
            m = stl[0]
            assert m == ora_p.shape[0] == want_p.shape[0], (m, ora_p.shape, want_p.shape)
            assert np.array_equal(got_l.cpu().numpy(), ora_l), (got_l.cpu().numpy(), ora_l)
            g = got_p[:m].cpu().numpy()
            assert np.array_equal(g.view(np.uint32), ora_p.view(np.uint32)), \
                "capacity-mode form differs from the oracle (B=%d n=%d elem_points=%d)" % (len(lens), len(pts), elem_points)
            assert torch.equal(got_p[:m].cpu(), want_p.cpu()), "capacity-mode form differs from the hash form"
            forms += 1
    if small:   # one voxel too few: reported, empty
        _, _, st = ops.batch_grid_subsample_async(Pc, lens_dev, dl, cap, elem_cap=mv - 1, elem_points=max(lens))
        torch.cuda.synchronize()
        assert mv == 1 or st.tolist() == [0, _lib.ST_OUT_OVERFLOW], st.tolist()
    nforms += forms
# a cloud above its point capacity is REPORTED by the one-workgroup form
_, _, st = ops.batch_grid_subsample_async(torch.rand((9000, 3), device=dev), torch.tensor([6000, 3000], dtype=torch.int32, device=dev),
                                          0.05, 9000, elem_points=4096)
torch.cuda.synchronize()
assert st.tolist() == [0, _lib.ST_OUT_OVERFLOW], st.tolist()
# a grid whose (element, voxel key) does not fit 32 bits: flagged, empty -- and the synchronous (hash) call still answers
wide = (rng.random((5000, 3)) * 2000.0).astype(np.float32)
Pw = torch.from_numpy(wide).to(dev)
_, _, st = ops.batch_grid_subsample_async(Pw, torch.tensor([5000], dtype=torch.int32, device=dev), 0.05, 5000)
torch.cuda.synchronize()
assert st.tolist() == [0, _lib.ST_KEY_WIDTH], st.tolist()
wp, wl, _, _ = ops.batch_grid_subsample(Pw, [5000], 0.05)
op, ol = co.batch_grid_subsampling(wide, np.asarray([5000], np.int32), 0.05)
assert np.array_equal(wp.cpu().numpy().view(np.uint32), op.view(np.uint32))
print("SORT-PATH-OK", len(cases), nforms)
