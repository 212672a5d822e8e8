#!/usr/bin/env python3
"""Turn a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace into the per-kernel stats table that
`--stats` prints: name, calls, total/avg/min/max duration (us), share.  Usage:
    python tools/rocpd_summary.py gpurun_out/prof/*/*_results.db > profiles/<name>.csv
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                      "from kernels group by name order by sum(end-start) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print("kernel,calls,total_us,avg_us,min_us,max_us,percent")
    for n, c, s, a, mn, mx in rows:
        short = n.split("(")[0].replace("void ", "")
        print('"%s",%d,%.1f,%.2f,%.2f,%.2f,%.2f' % (short, c, s / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot))


if __name__ == "__main__":
    main(sys.argv[1])
