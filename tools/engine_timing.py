#!/usr/bin/env python3
"""Where does a fragment's wall time go?  Host time of each engine call vs GPU time of one replay (tuning aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from d3feat_amd.engine import FragmentEngine
from d3feat_amd.models.variables import build_variables
from d3feat_amd.utils.config import threedmatch_config
from d3feat_amd.utils.synthetic import room_fragment
cfg = threedmatch_config()
dev = torch.device('cuda', 0)
W = build_variables(cfg, seed=42).values
limits = np.asarray([42, 42, 46, 51, 49], np.int32)
raw = torch.from_numpy(room_fragment(0)).to(dev)
eng = FragmentEngine(cfg, W, limits, raw_cap=320000, n0_cap=40000, slots=3, device=dev)
sl = eng.slots[0]
# GPU time of one replay alone
for _ in range(3): eng.run(raw, 0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(sl.stream):
        e0.record(); sl.graph.replay(); e1.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    ts.append((t1 - t0, e0.elapsed_time(e1)))
print("replay(): host %.3f ms, GPU %.3f ms (medians)" % (np.median([a for a, b in ts]) * 1e3, np.median([b for a, b in ts])))
# host time of submit / fetch in the pipelined loop
hs, hf = [], []
busy = [False] * 3
t_all = time.perf_counter()
for i in range(30):
    k = i % 3
    if busy[k]:
        t0 = time.perf_counter(); eng.fetch(k); hf.append(time.perf_counter() - t0)
    t0 = time.perf_counter(); eng.submit(k, raw); hs.append(time.perf_counter() - t0); busy[k] = True
for k in range(3):
    if busy[k]: eng.fetch(k)
t_all = time.perf_counter() - t_all
print("pipelined: %.3f ms/fragment; submit host %.3f ms, fetch host(wait) %.3f ms" % (t_all / 30 * 1e3, np.median(hs) * 1e3, np.median(hf) * 1e3))
