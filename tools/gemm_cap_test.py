import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from d3feat_amd import ops
dev = torch.device('cuda', 0)
def t(A, B, reps=30):
    for _ in range(3): ops.gemm(A, B, leaky=True)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(reps): ops.gemm(A, B, leaky=True)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
for (m, cap, K, N) in ((197, 2045, 7680, 512), (197, 2045, 512, 2048), (905, 5112, 3840, 256), (3632, 12780, 1920, 128)):
    A = torch.randn(m, K, device=dev); Bm = torch.randn(K, N, device=dev)
    Acap = torch.zeros(cap, K, device=dev); Acap[:m] = A
    nd = torch.tensor([m], dtype=torch.int32, device=dev)
    base = t(A, Bm)
    Acap.n_dev = nd; Acap.n_hint = int(m * 1.3)
    capt = t(Acap, Bm)
    Acap.n_hint = 0
    nohint = t(Acap, Bm)
    print("M=%d cap=%d K=%d N=%d: exact %.1f us, capacity+hint %.1f us, capacity no hint %.1f us" % (m, cap, K, N, base, capt, nohint))
