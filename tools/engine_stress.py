import sys; sys.path.insert(0,'.')
import faulthandler; faulthandler.dump_traceback_later(100, exit=True)
import numpy as np, torch
from d3feat_amd import engine as E
from d3feat_amd.models.variables import build_variables
from d3feat_amd.utils.config import threedmatch_config
from d3feat_amd.utils.synthetic import room_fragment
cfg = threedmatch_config()
dev = torch.device('cuda',0)
nslots = int(sys.argv[1]); iters = int(sys.argv[2])
W = build_variables(cfg, seed=42, randomize_bn=True).values
limits = np.asarray([37, 35, 36, 38, 38], np.int32)
eng = E.FragmentEngine(cfg, W, limits, raw_cap=45000, n0_cap=14000, slots=nslots, device=dev)
raws = [torch.from_numpy(room_fragment(20 + i, n_raw=n, edge=1.0)).to(dev) for i, n in enumerate((30000, 40000, 25000, 35000))]
refs = []
for r in raws:
    p, d, s = eng.run_eager(r); refs.append((p.clone(), d.clone(), s.clone()))
torch.cuda.synchronize(); print("refs ok", flush=True)
bad = 0; inflight = [None] * nslots
def check(i, out):
    global bad
    p, d, s = out
    rp, rd, rs = refs[i]
    ok = p.shape == rp.shape and torch.equal(p, rp) and (d - rd).abs().max().item() < 1e-5 and (s - rs).abs().max().item() < 1e-5
    if not ok:
        bad += 1
        print("MISMATCH frag", i, p.shape, rp.shape, flush=True)
for it in range(iters):
    sl = it % nslots
    if inflight[sl] is not None:
        check(inflight[sl], eng.fetch(sl))
    eng.submit(sl, raws[it % len(raws)]); inflight[sl] = it % len(raws)
for sl in range(nslots):
    if inflight[sl] is not None: check(inflight[sl], eng.fetch(sl))
print("slots", nslots, "iters", iters, "bad", bad, "fallbacks", eng.fallbacks, flush=True)
