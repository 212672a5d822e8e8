"""kpconv_fused32 A/B on level-0 shapes of F stacked fragments, HIP-graph timed (10 launches per replay): feature rows one chunk
ahead (D3F_KP_AHEAD; experiment k7 -- the kernel variant was measured and reverted, the shipped library ignores the switch, so
the two lines time the same kernel) against the round-5 loop; KP_MFMA_AB=1 adds the matrix-core aggregation (D3F_KP_MFMA) against the vector form."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from d3feat_amd import ops, tf_custom_ops as tfo
from d3feat_amd.kernels.kernel_points import create_kernel_points
from d3feat_amd.utils.synthetic import room_fragment
F = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda", 0)
subs = [tfo.grid_subsampling(torch.from_numpy(room_fragment(i, 300000, 1.68)).to(dev), 0.03) for i in range(F)]
pts = torch.cat([x for s in subs for x in (s, s)], 0).contiguous()
lens = ops.as_lens([int(s.shape[0]) for s in subs for _ in (0, 1)], dev)
grid = ops.NeighborGrid(pts, lens, 0.075)
nb, _ = grid.search(pts, lens, 37, query_grid=grid, internal=True)
P = grid.xyz.contiguous()
g = torch.Generator(device="cpu").manual_seed(0)
KP = create_kernel_points(0.045, 15, 1, 3, "center", rng=np.random.default_rng(1)).reshape(15, 3).astype(np.float32)
f = torch.randn((P.shape[0], 32), generator=g).to(dev)
W = (torch.randn((15, 32, 32), generator=g) * 0.05).to(dev)
def timed(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph(); st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        with torch.cuda.graph(gr, stream=st):
            for _ in range(iters): fn()
        gr.replay(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); gr.replay(); e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
ref = None
for rep in range(2):
    for ahead in ("0", "1"):
        os.environ["D3F_KP_AHEAD"] = ahead
        ops.KP_MFMA = False
        out = ops.kpconv_fused32(P, P, nb, f, KP, W, 0.03, leaky=True)
        if ref is None:
            ref = out.clone()
        same = bool(torch.equal(out, ref))
        t = timed(lambda: ops.kpconv_fused32(P, P, nb, f, KP, W, 0.03, leaky=True))
        print("F=%d rows %d: kpconv_fused32 rows-ahead %s: %.1f us (row_positive included), bit-equal to the first form: %s" % (F, P.shape[0], ahead, t, same))
if os.environ.get("KP_MFMA_AB"):
    for flag in (True, False):
        ops.KP_MFMA = flag
        t = timed(lambda: ops.kpconv_fused32(P, P, nb, f, KP, W, 0.03, leaky=True))
        print("F=%d rows %d: kpconv_fused32 %s: %.1f us" % (F, P.shape[0], "matrix-core aggregation" if flag else "vector aggregation     ", t))
