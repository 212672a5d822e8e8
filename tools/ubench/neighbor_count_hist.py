import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from d3feat_amd import ops, tf_custom_ops as tfo
from d3feat_amd.utils.synthetic import room_fragment
dev = torch.device("cuda", 0)
subs = [tfo.grid_subsampling(torch.from_numpy(room_fragment(s, n_raw=300000, edge=1.68)).to(dev), 0.03) for s in range(2)]
pts = torch.cat([x for s in subs for x in (s, s)], 0)
lens = [int(s.shape[0]) for s in subs for _ in (0, 1)]
levels = [(pts, lens)]
dl = 0.06
for l in range(4):
    p, pl, _, _ = ops.batch_grid_subsample(levels[-1][0], levels[-1][1], dl)
    levels.append((p, [int(x) for x in pl.tolist()])); dl *= 2
r = 0.075
for l, (p, pl) in enumerate(levels):
    g = ops.NeighborGrid(p, pl, r)
    for tag, (q, ql) in (("conv", (p, pl)),) + ((("pool", levels[l + 1]),) if l + 1 < len(levels) else ()):
        out, st = g.search(q, ql, 256, cap=256)
        n = (out < p.shape[0]).sum(1).cpu().numpy()
        print("L%d %s: rows %d mean %.1f p50 %d p80 %d p95 %d max %d  >32: %.3f >64: %.3f >128: %.4f" % (l, tag, len(n), n.mean(), np.percentile(n, 50), np.percentile(n, 80), np.percentile(n, 95), n.max(), (n > 32).mean(), (n > 64).mean(), (n > 128).mean()))
    r *= 2
