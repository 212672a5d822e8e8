#!/usr/bin/env python3
"""Concurrency / launch-gap summary of a rocprofv3 kernel trace (rocpd sqlite output): how busy the GPU is while several
graph replays are in flight, and what the dependent-launch gaps inside one stream cost.  SURVEY.md §8(d) asks for the
kernel count and the launch gaps next to the roofline figures.  Usage:
    python tools/timeline_summary.py <results.db> [lo_frac hi_frac]     (window: that slice of the trace; default: the steady
                                                                         state, found from the streams' activity)
"""
import json
import sqlite3
import sys

import numpy as np


def main(path, lo=-1.0, hi=0.9):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
    lane = next((c for c in ("stream_id", "queue_id", "stream", "queue") if c in cols), None)
    q = "select start, end, name%s from kernels order by start" % ((", " + lane) if lane else "")
    rows = db.execute(q).fetchall()
    st = np.array([r[0] for r in rows], np.int64)
    en = np.array([r[1] for r in rows], np.int64)
    names = [r[2].split("(")[0].replace("void ", "") for r in rows]
    ln = np.array([r[3] if lane else 0 for r in rows])
    t0, t1 = st.min(), en.max()
    if lane and lo < 0:
        # steady state = the longest run of 2 ms bins in which at least 3 streams launch kernels (several replays in flight);
        # 10 % trimmed at both ends
        nb = int((t1 - t0) // 2000000) + 1
        seen = [set() for _ in range(nb)]
        for s_, l_ in zip(st, ln):
            seen[int((s_ - t0) // 2000000)].add(l_)
        ok = [len(x) >= 3 for x in seen]
        best, cur0 = (0, 0), None
        for i, f in enumerate(ok + [False]):
            if f and cur0 is None:
                cur0 = i
            if not f and cur0 is not None:
                if i - cur0 > best[1] - best[0]:
                    best = (cur0, i)
                cur0 = None
        span = (best[1] - best[0]) * 2000000
        w0, w1 = t0 + best[0] * 2000000 + span // 10, t0 + best[1] * 2000000 - span // 10
    else:
        w0, w1 = t0 + int((t1 - t0) * max(lo, 0.0)), t0 + int((t1 - t0) * hi)
    sel = (st >= w0) & (en <= w1)
    st, en, ln = st[sel], en[sel], ln[sel]
    names = [n for n, s in zip(names, sel) if s]
    win = float(w1 - w0)
    # union of the busy intervals, time by concurrency level
    ev = np.concatenate([np.stack([st, np.ones_like(st)], 1), np.stack([en, -np.ones_like(en)], 1)])
    ev = ev[np.lexsort((-ev[:, 1], ev[:, 0]))]
    level_time = {}
    cur, last = 0, w0
    for t, d in ev:
        level_time[cur] = level_time.get(cur, 0) + (t - last)
        cur += int(d)
        last = t
    level_time[cur] = level_time.get(cur, 0) + (w1 - last)
    busy = 1.0 - level_time.get(0, 0) / win
    out = {"lane_column": lane, "window_start_ms": round((w0 - t0) / 1e6, 2), "window_ms": round(win / 1e6, 3), "kernels_in_window": int(len(st)),
           "busy_fraction": round(busy, 4), "mean_concurrency": round(float((en - st).sum()) / win, 3),
           "time_share_by_concurrent_kernels": {str(k): round(v / win, 4) for k, v in sorted(level_time.items())}}
    # per stream: kernels, busy share, gaps between consecutive kernels
    per = {}
    for l in np.unique(ln):
        m = ln == l
        s_, e_ = st[m], en[m]
        o = np.argsort(s_)
        s_, e_ = s_[o], e_[o]
        gaps = (s_[1:] - e_[:-1]) / 1e3
        gaps = gaps[gaps > 0]
        per[str(l)] = {"kernels": int(m.sum()), "busy_share": round(float((e_ - s_).sum()) / win, 4),
                       "gap_us_median": round(float(np.median(gaps)), 2) if len(gaps) else None,
                       "gap_us_p90": round(float(np.percentile(gaps, 90)), 2) if len(gaps) else None,
                       "gap_share": round(float(np.minimum(gaps, 200.0).sum()) * 1e3 / win, 4) if len(gaps) else None}
    out["per_stream"] = per
    # which kernels run while NOTHING else does (serial latency exposed) -- top by exposed time
    order = np.argsort(st)
    st_o, en_o = st[order], en[order]
    nm_o = [names[i] for i in order]
    alone = {}
    # a kernel is "alone" for the part of its interval not covered by any other kernel: approximate with the level-1 segments
    ev2 = sorted([(int(s), 1, i) for i, s in enumerate(st_o)] + [(int(e), -1, i) for i, e in enumerate(en_o)])
    active = set()
    last = w0
    for t, d, i in ev2:
        if len(active) == 1:
            k = nm_o[next(iter(active))]
            alone[k] = alone.get(k, 0) + (t - last)
        if d == 1:
            active.add(i)
        else:
            active.discard(i)
        last = t
    top = sorted(alone.items(), key=lambda kv: -kv[1])[:12]
    out["running_alone_share_top"] = {k: round(v / win, 4) for k, v in top}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    a = sys.argv
    main(a[1], *(float(x) for x in a[2:4]))
