import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from d3feat_amd import ops, tf_custom_ops as tfo
from d3feat_amd.utils.synthetic import room_fragment
dev = torch.device("cuda", 0)
subs = [tfo.grid_subsampling(torch.from_numpy(room_fragment(s, n_raw=300000, edge=1.68)).to(dev), 0.03) for s in range(4)]
pts = torch.cat([x for s in subs for x in (s, s)], 0)
lens = [int(s.shape[0]) for s in subs for _ in (0, 1)]
levels = [(pts, lens)]
dl = 0.06
for l in range(4):
    p, pl, _, _ = ops.batch_grid_subsample(levels[-1][0], levels[-1][1], dl)
    levels.append((p, [int(x) for x in pl.tolist()])); dl *= 2
r = 0.075
limits = [37, 35, 36, 38, 38]
for l, (p, pl) in enumerate(levels):
    g = ops.NeighborGrid(p, pl, r)
    for dbg in ("32", "35"):
        os.environ["D3F_NBC_DBG"] = dbg
        st = torch.zeros((2,), dtype=torch.int32, device=dev)
        out, _ = g.search(p, pl, limits[l], cap=192, status=st, reset_status=False, want_kmax=False)
        torch.cuda.synchronize()
        print("L%d conv dbg %s: rows %d exact-path queries %d max m %d" % (l, dbg, p.shape[0], int(st[0]), int(st[1])))
    r *= 2
