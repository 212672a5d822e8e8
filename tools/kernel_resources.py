#!/usr/bin/env python3
"""Registers / LDS / occupancy of every kernel in d3feat_amd/csrc (hipcc -Rpass-analysis=kernel-resource-usage), one line each.
Static LDS only: kernels launched with dynamic LDS show 0 here (their size is in the launch code)."""
import os, re, subprocess, sys
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "d3feat_amd", "csrc")
files = sys.argv[1:] or sorted(f for f in os.listdir(root) if f.endswith(".hip"))
print("%-58s %5s %5s %7s %4s %7s" % ("kernel", "VGPR", "AGPR", "scratch", "occ", "LDS"))
for f in files:
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-I../../include",
                        "-Rpass-analysis=kernel-resource-usage", "-c", f, "-o", "/dev/null"], cwd=root, capture_output=True, text=True)
    cur = {}
    for line in r.stderr.splitlines():
        m = re.search(r"remark: +(.*?): +(.*?) \[-Rpass", line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2).strip()
        if k == "Function Name":
            cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip().split("(")[0]}
        cur[k] = v
        if k.startswith("LDS Size"):
            print("%-58s %5s %5s %7s %4s %7s" % (cur["name"][-58:], cur.get("VGPRs"), cur.get("AGPRs"), cur.get("ScratchSize [bytes/lane]"),
                                                 cur.get("Occupancy [waves/SIMD]"), v))
