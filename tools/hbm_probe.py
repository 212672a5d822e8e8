#!/usr/bin/env python3
"""What this box's HBM actually sustains for the access mixes of the memory-bound kernels (torch fill / copy / reduce over
buffers that do not fit the 256 MB Infinity Cache, and over the 60-180 MB sizes of the fine-level GEMM operands)."""
import torch

dev = torch.device("cuda", 0)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3


for mb in (60, 120, 240, 1024):
    n = mb * (1 << 20) // 4
    a = torch.empty(n, device=dev).normal_()
    b = torch.empty(n, device=dev)
    t_fill = timed(lambda: b.fill_(1.0))
    t_copy = timed(lambda: b.copy_(a))
    t_read = timed(lambda: a.sum())
    t_axpy = timed(lambda: torch.add(a, b, out=b))
    print("%5d MB: write %.2f TB/s   copy %.2f TB/s (r+w)   read %.2f TB/s   a+b->b %.2f TB/s (2r+w)" %
          (mb, mb / 1048576 / t_fill * 1.048576, 2 * mb / 1048576 / t_copy * 1.048576, mb / 1048576 / t_read * 1.048576,
           3 * mb / 1048576 / t_axpy * 1.048576))
