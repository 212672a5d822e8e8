#!/usr/bin/env python3
"""Turn a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace into the per-kernel stats table that
`--stats` prints: name, calls, total/avg/min/max duration (us), share.  Usage:
    python tools/rocpd_summary.py gpurun_out/prof/*/*_results.db > profiles/<name>.csv
    python tools/rocpd_summary.py --timed-region [--fragments K] <results.db>
        only the launches between bench.py's two `d3f_trace_marker_kernel` launches, i.e. the K full-size fragments of the
        timed region replayed from the captured graphs: capture warm-ups, the calibration prologue and every untimed leg
        of the run are cut away.  With --fragments K two more columns give calls / total_us per fragment.
"""
import sqlite3
import sys


def main(argv):
    timed = "--timed-region" in argv
    frags = 0
    if "--fragments" in argv:
        frags = int(argv[argv.index("--fragments") + 1])
        argv = [a for i, a in enumerate(argv) if a != "--fragments" and argv[i - 1] != "--fragments"]
    path = [a for a in argv if not a.startswith("--")][0]
    db = sqlite3.connect(path)
    where = ""
    note = ""
    if timed:
        marks = db.execute("select start, end from kernels where name like '%d3f_trace_marker_kernel%' order by start").fetchall()
        if len(marks) < 2:
            sys.exit("no two d3f_trace_marker_kernel launches in %s" % path)
        lo, hi = marks[0][1], marks[1][0]
        where = " where start >= %d and end <= %d and name not like '%%d3f_trace_marker_kernel%%'" % (lo, hi)
        note = "# timed region only: %.3f ms between the markers" % ((hi - lo) / 1e6)
        if frags:
            note += ", %d fragments -> %.4f ms per fragment wall (profiler attached: streams are serialised)" % (frags, (hi - lo) / 1e6 / frags)
    rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                      "from kernels%s group by name order by sum(end-start) desc" % where).fetchall()
    tot = sum(r[2] for r in rows) or 1
    if note:
        print(note + "; summed kernel time %.3f ms%s" % (tot / 1e6, (" = %.4f ms per fragment" % (tot / 1e6 / frags)) if frags else ""))
    print("kernel,calls,total_us,avg_us,min_us,max_us,percent" + (",calls_per_fragment,us_per_fragment" if frags else ""))
    for n, c, s, a, mn, mx in rows:
        short = n.split("(")[0].replace("void ", "")
        line = '"%s",%d,%.1f,%.2f,%.2f,%.2f,%.2f' % (short, c, s / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot)
        if frags:
            line += ",%.2f,%.2f" % (c / frags, s / 1e3 / frags)
        print(line)


if __name__ == "__main__":
    main(sys.argv[1:])
