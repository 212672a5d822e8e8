import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from d3feat_amd import ops
dev = torch.device("cuda", 0)
mode = sys.argv[1]
rng = np.random.default_rng(1)
def cloud(n, spread=2.2):
    return (rng.random((n, 3)) * spread).astype(np.float32)
if mode == "big_eager":
    lens = [300000, 299000, 301000, 300500]
    P = torch.from_numpy(np.concatenate([cloud(l) for l in lens])).to(dev)
    want_p, want_l, _, _ = ops.batch_grid_subsample(P, lens, 0.03)
    print("hash M", want_p.shape[0], flush=True)
    cap = 1264096
    Pc = torch.zeros((cap, 3), dtype=torch.float32, device=dev); Pc[: P.shape[0]] = P
    ld = torch.tensor(lens, dtype=torch.int32, device=dev)
    got_p, got_l, st = ops.batch_grid_subsample_async(Pc, ld, 0.03, 160000, elem_cap=40000)
    torch.cuda.synchronize(); print("async done", st.tolist(), flush=True)
    m = st.tolist()[0]
    print("equal", m == want_p.shape[0] and torch.equal(got_p[:m].cpu(), want_p.cpu()), flush=True)
elif mode == "room":
    from d3feat_amd.utils.synthetic import room_fragment
    raws = [room_fragment(sd, n_raw=300000, edge=1.68) for sd in range(4)]
    lens = [r.shape[0] for r in raws]
    P = torch.from_numpy(np.concatenate(raws)).to(dev)
    want_p, want_l, _, _ = ops.batch_grid_subsample(P, lens, 0.03)
    print("hash M", want_p.shape[0], want_l.tolist(), flush=True)
    cap = 1264096
    Pc = torch.zeros((cap, 3), dtype=torch.float32, device=dev); Pc[: P.shape[0]] = P
    ld = torch.tensor(lens, dtype=torch.int32, device=dev)
    got_p, got_l, st = ops.batch_grid_subsample_async(Pc, ld, 0.03, 135168, elem_cap=33792)
    torch.cuda.synchronize(); print("async done", st.tolist(), got_l.tolist(), flush=True)
    m = st.tolist()[0]
    print("equal", m == want_p.shape[0] and torch.equal(got_p[:m].cpu(), want_p.cpu()), flush=True)
    if len(sys.argv) > 2 and sys.argv[2] == "twice":
        for rep in range(3):
            print("eager call", rep, flush=True)
            got_p, got_l, st = ops.batch_grid_subsample_async(Pc, ld, 0.03, 135168, elem_cap=33792)
            torch.cuda.synchronize(); print("  ->", st.tolist(), flush=True)
            m = st.tolist()[0]
            print("  equal", m == want_p.shape[0] and torch.equal(got_p[:m].cpu(), want_p.cpu()), flush=True)
    elif len(sys.argv) > 2:
        s = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(s):
            with ops.private_workspace():
                ops.batch_grid_subsample_async(Pc, ld, 0.03, 135168, elem_cap=33792)
        s.synchronize(); print("warm ok", flush=True)
        g = torch.cuda.CUDAGraph()
        with ops.private_workspace() as pw:
            with torch.cuda.graph(g, stream=s):
                got_p, got_l, st = ops.batch_grid_subsample_async(Pc, ld, 0.03, 135168, elem_cap=33792)
        print("captured", flush=True)
        for rep in range(3):
            g.replay(); torch.cuda.synchronize(); print("replayed", st.tolist(), flush=True)
        m = st.tolist()[0]
        print("equal (graph)", m == want_p.shape[0] and torch.equal(got_p[:m].cpu(), want_p.cpu()), flush=True)
elif mode == "capture":
    lens = [30000, 29000, 31000, 30500]
    P = torch.from_numpy(np.concatenate([cloud(l) for l in lens])).to(dev)
    want_p, want_l, _, _ = ops.batch_grid_subsample(P, lens, 0.03)
    cap = int(sys.argv[2]) if len(sys.argv) > 2 else 126464
    Pc = torch.zeros((cap, 3), dtype=torch.float32, device=dev); Pc[: P.shape[0]] = P
    ld = torch.tensor(lens, dtype=torch.int32, device=dev)
    s = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s):
        with ops.private_workspace():
            ops.batch_grid_subsample_async(Pc, ld, 0.03, 160000, elem_cap=40000)
    s.synchronize(); print("warm ok", flush=True)
    g = torch.cuda.CUDAGraph()
    with ops.private_workspace() as pw:
        with torch.cuda.graph(g, stream=s):
            got_p, got_l, st = ops.batch_grid_subsample_async(Pc, ld, 0.03, 160000, elem_cap=40000)
    print("captured", flush=True)
    g.replay(); torch.cuda.synchronize(); print("replayed", st.tolist(), flush=True)
    m = st.tolist()[0]
    print("equal", m == want_p.shape[0] and torch.equal(got_p[:m].cpu(), want_p.cpu()), flush=True)
