#!/usr/bin/env python3
"""Launches of the kernels whose name contains SUBSTR inside bench.py's timed region, grouped by (name, grid size):
    python tools/kernel_by_grid.py <results.db> SUBSTR
-- which launches of a many-launch kernel (searches, contractions) a change helped and which it hurt."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
sub = sys.argv[2]
marks = db.execute("select start, end from kernels where name like '%d3f_trace_marker_kernel%' order by start").fetchall()
lo, hi = (marks[0][1], marks[1][0]) if len(marks) >= 2 else (0, 1 << 62)
cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
gcol = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else ("grid_size" if "grid_size" in cols else None))
if gcol is None:
    sys.exit("no grid column in %s" % cols)
rows = db.execute("select name, %s, count(*), avg(end-start), min(end-start), max(end-start) from kernels where start >= ? and end <= ? "
                  "and name like ? group by name, %s order by name, %s" % (gcol, gcol, gcol), (lo, hi, "%" + sub + "%")).fetchall()
for n, g, c, a, mn, mx in rows:
    print("%-52s grid %9d  x%4d  avg %8.2f us  min %8.2f  max %8.2f" % (n.split("(")[0].replace("void ", "")[:52], g, c, a / 1e3, mn / 1e3, mx / 1e3))
