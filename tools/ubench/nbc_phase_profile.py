"""Per-phase shader-clock cycles of nb_cell_search_kernel (D3F_NBC_PROF: s_memtime sums / maxima over the wavefronts).
Needs a measurement build of the library:  make -C d3feat_amd/csrc -B EXTRA=-DD3F_NBC_MEASURE  (the production kernel ignores the
measurement arguments: as run-time tests they cost it ~8 % -- profiles/r06_experiments.txt n11)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
names = ["prologue", "stencil", "tests", "order", "exact+pad", "write-out"]
for l, (p, pl) in enumerate(levels[:3]):
    g = ops.NeighborGrid(p, pl, r)
    for q in sys.argv[1:] or ["16", "4", "1"]:
        os.environ["D3F_NBC_Q"] = q
        prof = torch.zeros((8 * (p.shape[0] + 64),), dtype=torch.int64, device=dev)
        st = torch.zeros((2,), dtype=torch.int32, device=dev)
        out = torch.empty((p.shape[0], limits[l]), dtype=torch.int32, device=dev)
        for rep in range(4):
            prof.zero_()
            if rep == 3:
                os.environ["D3F_NBC_PROF"] = hex(prof.data_ptr())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.search(p, pl, limits[l], cap=192, status=st, reset_status=False, want_kmax=False, out=out)
            e1.record()
            torch.cuda.synchronize()
            if rep == 2:
                t_plain = e0.elapsed_time(e1) * 1e3
        t_prof = e0.elapsed_time(e1) * 1e3
        os.environ.pop("D3F_NBC_PROF")
        print("   kernel %.1f us plain, %.1f us with the timers" % (t_plain, t_prof))
        v = prof.cpu().numpy().reshape(-1, 8)
        v = v[v[:, 7] > 0]
        t0, t1 = v[:, 6].min(), v[:, 7].max()
        life = v[:, 7] - v[:, 6]
        print("L%d rows %d Q=%s: waves %d, launch span %d cycles; wave life mean %d p50 %d p95 %d max %d | per wave mean (p95, max): " % (
            l, p.shape[0], q, len(v), t1 - t0, life.mean(), np.percentile(life, 50), np.percentile(life, 95), life.max()) +
              "  ".join("%s %d (%d, %d)" % (names[k], v[:, k].mean(), np.percentile(v[:, k], 95), v[:, k].max()) for k in range(6)))
        # how many wavefronts are alive over time (20 samples)
        ts = np.linspace(t0, t1, 22)[1:-1]
        print("   wavefronts alive per SIMD over the launch: " + " ".join("%.1f" % (((v[:, 6] <= t) & (v[:, 7] > t)).sum() / 1024.0) for t in ts))
    r *= 2
