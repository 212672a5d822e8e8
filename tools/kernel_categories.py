#!/usr/bin/env python3
"""Group a kernel_stats.csv (tools/rocpd_summary.py) into pipeline stages; prints us per fragment.
    python tools/kernel_categories.py gpurun_out/<tag>/kernel_stats.csv <fragments>"""
import csv, sys
CATS = [("gemm", ("gemm_",)), ("neighbour search", ("nb_search",)), ("neighbour grid build", ("nb_", "bbox_kernel<NbPrepEpi")),
        ("kpconv aggregate", ("kpconv_", "kp_rowpos")), ("grid subsample", ("gs_", "bbox_kernel<GsPrepEpi", "scan_fold_kernel<GsMarkIn")),
        ("scan / reset helpers", ("scan_", "begin_kernel", "fill_u32", "copy_i32", "offsets_", "bbox_")),
        ("pools / upsample", ("colmin", "maxpool", "upsample")), ("head", ("head_",)), ("torch / runtime", ("at::", "__amd", "void at"))]
rows = list(csv.DictReader(open(sys.argv[1])))
nfrag = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = {c: [0.0, 0] for c, _ in CATS}
tot["other"] = [0.0, 0]
for r in rows:
    name = r["kernel"]
    for c, pre in CATS:
        if any(name.startswith(p) or ("<" in p and p in name) for p in pre):
            break
    else:
        c = "other"
    # more specific: nb_search before nb_, handled by order
    tot[c][0] += float(r["total_us"]); tot[c][1] += int(r["calls"])
allus = sum(v[0] for v in tot.values())
for c, (us, n) in sorted(tot.items(), key=lambda kv: -kv[1][0]):
    print("%-22s %8.1f us/frag  %6.1f launches/frag  %5.1f%%" % (c, us / nfrag, n / nfrag, 100 * us / allus))
print("%-22s %8.1f us/frag  %6.1f launches/frag" % ("TOTAL", allus / nfrag, sum(v[1] for v in tot.values()) / nfrag))
