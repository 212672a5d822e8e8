#!/usr/bin/env python3
"""Timeline of one stage-0 grid-subsampling call inside the timed region of a rocprofv3 kernel trace (rocpd sqlite):
kernel by kernel, start offset and duration -- where the preprocessing chain of a replay spends its time.
    python tools/gs_timeline.py <results.db> [which stage-0 call, default 2] [kernels to print, default 48]"""
import glob
import sqlite3
import sys


def main(path, which=2, count=48):
    db = sqlite3.connect(path)
    marks = db.execute("select start,end from kernels where name like '%trace_marker%' order by start").fetchall()
    lo, hi = marks[0][1], marks[1][0]
    rows = db.execute("select name, grid_x, start, end-start, stream_id from kernels where start>=? and end<=? order by start", (lo, hi)).fetchall()
    big = max(r[1] for r in rows if "gs_sortkey" in r[0])
    idx = [i for i, r in enumerate(rows) if "gs_sortkey" in r[0] and r[1] == big][which]
    sid, t0, k = rows[idx][4], rows[idx][2], 0
    for r in rows[max(idx - 3, 0):]:
        if r[4] != sid:
            continue
        print("%-52s grid %8d  +%8.1f us  dur %7.1f" % (r[0].split("(")[0].replace("void ", "")[:52], r[1], (r[2] - t0) / 1e3, r[3] / 1e3))
        k += 1
        if k > count:
            break


if __name__ == "__main__":
    a = sys.argv[1:]
    main(glob.glob(a[0])[0], int(a[1]) if len(a) > 1 else 2, int(a[2]) if len(a) > 2 else 48)
