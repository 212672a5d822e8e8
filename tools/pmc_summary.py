#!/usr/bin/env python3
"""Summarise rocprofv3 counter-collection CSVs (one --pmc pass per counter group) into per-kernel HBM traffic.

    python tools/pmc_summary.py <dir with *counter_collection.csv> [...more dirs] > profiles/<tag>_hbm_traffic.json

Per kernel (template arguments kept, argument list dropped): launches, mean FETCH_SIZE / WRITE_SIZE as reported
(KiB, the unit rocprofv3 uses for both) and the HBM bytes per launch after the two corrections of
MI355X_MICROARCH.md §HBM: counters are in KiB (x1024), and on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes,
i.e. reports half of a wide coalesced stream (x2 on the read side; WRITE_SIZE is taken as reported):
    traffic_bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024
"""
import csv
import glob
import json
import os
import sys


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def main(dirs):
    acc = {}
    region = {"files": 0, "with_markers": 0}
    for d in dirs:
        for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(fn, newline="") as f:
                rows = list(csv.DictReader(f))
            # only the launches of bench.py's timed region (between its first and last d3f_trace_marker_kernel): warm-up captures,
            # the parity pass and the F = 1 latency replays that follow would otherwise dilute the per-launch means
            marks = [int(r["Dispatch_Id"]) for r in rows if "d3f_trace_marker_kernel" in (r.get("Kernel_Name") or "")]
            lo, hi = (min(marks), max(marks)) if len(set(marks)) >= 2 else (None, None)
            region["files"] += 1
            region["with_markers"] += 1 if lo is not None else 0
            if True:
                for row in rows:
                    if lo is not None and not (lo < int(row["Dispatch_Id"]) < hi):
                        continue
                    k = short(row.get("Kernel_Name") or row.get("kernel_name") or "")
                    c = row.get("Counter_Name") or row.get("counter_name")
                    v = float(row.get("Counter_Value") or row.get("counter_value") or 0.0)
                    a = acc.setdefault(k, {})
                    s = a.setdefault(c, [0.0, 0])
                    s[0] += v
                    s[1] += 1
    out = {}
    for k, a in sorted(acc.items()):
        e = {"launches": max(s[1] for s in a.values())}
        for c, s in a.items():
            e[c + "_KiB_mean"] = round(s[0] / max(s[1], 1), 3)
        if "FETCH_SIZE" in a or "WRITE_SIZE" in a:
            f = a.get("FETCH_SIZE", [0.0, 1])
            w = a.get("WRITE_SIZE", [0.0, 1])
            e["traffic_bytes_per_launch"] = int(2 * 1024 * f[0] / max(f[1], 1) + 1024 * w[0] / max(w[1], 1))
        out[k] = e
    # tie the counters to the kernel sources they measured (bench.py marks `traffic_stale` when the hash differs)
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for fn in sorted(glob.glob(os.path.join(root, "d3feat_amd", "csrc", "*.h*"))):
        h.update(os.path.basename(fn).encode())
        h.update(open(fn, "rb").read())
    out["__source_hash__"] = h.hexdigest()[:16]
    # fragments per replay of the counter run (tools/gpu_visit.sh pmc: --batch 4): per-launch traffic scales with it, bench.py
    # rescales to the fragments per launch of its own instrumented pass
    out["__timed_region_only__"] = region["files"] > 0 and region["with_markers"] == region["files"]
    out["__fragments_per_launch__"] = int(os.environ.get("D3F_PMC_FRAGMENTS_PER_LAUNCH", "4"))
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(sys.argv[1:] or ["."])
