#!/usr/bin/env python3
"""Matrix-pipe utilisation of the MFMA kernels from ONE `rocprofv3 --pmc` pass (tools/gpu_visit.sh mfma).

    python tools/pmc_mfma.py <dir with *counter_collection.csv> [--trace <dir with *kernel_trace.csv>] > profiles/<tag>_mfma_counters.json

Counters (MI355X_MICROARCH.md, "Per-instruction cycle constants" / "rocprofv3 PMC slots"):
  SQ_VALU_MFMA_BUSY_CYCLES   cycles a SIMD's matrix pipe is busy, summed over the chip's 1024 SIMDs (= 32 x N for
                             v_mfma_f32_32x32x16_bf16, 16 x N for v_mfma_f32_16x16x32_bf16 at the 8- and 4-pass rates)
  SQ_BUSY_CYCLES             cycles the SQ of a shader engine has work (summed over the instances rocprofv3 reports)
  GRBM_GUI_ACTIVE            cycles the GPU is active, summed over the 8 XCDs: / 8 = the kernel's duration in shader cycles
  SQ_INSTS_VALU_MFMA_MOPS_BF16 / _F32   matrix operations in units of 512 flops
Per kernel (template arguments kept): mean per dispatch of every counter and
  mfma_busy      = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)     fraction of all SIMD-cycles the matrix pipe is busy
  issued_tflops  = 512 x MOPS_BF16 / duration, with duration = (GRBM_GUI_ACTIVE / 8) cycles at the clock the pass ran at
                   (rocprofv3's own start/end stamps of the same dispatch when the trace CSV is there, else 2.4 GHz)
  issued_frac    = issued_tflops / 2516.6 (dense bf16 peak): must agree with mfma_busy when every MFMA runs at the full rate
"""
import csv
import glob
import hashlib
import json
import os
import sys

PEAK_BF16_TF = 2516.6
SIMDS = 1024


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def main(argv):
    dirs, trace_dirs = [], []
    it = iter(argv)
    for a in it:
        if a == "--trace":
            trace_dirs.append(next(it))
        else:
            dirs.append(a)
    acc, dur = {}, {}
    region = [0, 0]
    for d in dirs:
        for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(fn, newline="") as f:
                rows = list(csv.DictReader(f))
            # only the launches of bench.py's timed region (see tools/pmc_summary.py)
            marks = [int(r["Dispatch_Id"]) for r in rows if "d3f_trace_marker_kernel" in (r.get("Kernel_Name") or "")]
            lo, hi = (min(marks), max(marks)) if len(set(marks)) >= 2 else (None, None)
            region[0] += 1
            region[1] += 1 if lo is not None else 0
            if True:
                for row in rows:
                    if lo is not None and not (lo < int(row["Dispatch_Id"]) < hi):
                        continue
                    k = short(row.get("Kernel_Name") or "")
                    c = row.get("Counter_Name")
                    v = float(row.get("Counter_Value") or 0.0)
                    s = acc.setdefault(k, {}).setdefault(c, [0.0, 0])
                    s[0] += v
                    s[1] += 1
                    if row.get("Start_Timestamp") and row.get("End_Timestamp") and c == "GRBM_GUI_ACTIVE":
                        t = dur.setdefault(k, [0.0, 0])
                        t[0] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
                        t[1] += 1
    for d in trace_dirs + dirs:
        for fn in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            with open(fn, newline="") as f:
                for row in csv.DictReader(f):
                    k = short(row.get("Kernel_Name") or "")
                    if k in dur and dur[k][1] >= acc.get(k, {}).get("GRBM_GUI_ACTIVE", [0, 0])[1]:
                        continue
                    t = dur.setdefault(k, [0.0, 0])
                    t[0] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
                    t[1] += 1
    out = {}
    for k, a in sorted(acc.items()):
        mean = {c: s[0] / max(s[1], 1) for c, s in a.items()}
        mfma = mean.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        if mfma <= 0:
            continue
        e = {"launches": max(s[1] for s in a.values())}
        for c, v in sorted(mean.items()):
            e[c] = round(v, 1)
        gui = mean.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        if gui > 0:
            e["duration_cycles"] = round(gui, 1)
            e["mfma_busy"] = round(mfma / (SIMDS * gui), 4)
            ns = dur[k][0] / dur[k][1] if k in dur and dur[k][1] else None
            ghz = gui / ns if ns else 2.4
            if ns:
                e["duration_us_in_counter_pass"] = round(ns / 1e3, 2)
                e["clock_ghz_in_counter_pass"] = round(ghz, 3)
            mops = mean.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0)
            if mops > 0:
                tf = 512.0 * mops / (gui / ghz) / 1e3      # flops / ns = GF/s -> TF/s
                e["issued_tflops_bf16"] = round(tf, 1)
                e["issued_frac_of_2516.6"] = round(tf / PEAK_BF16_TF, 4)
                # the same ratio at the peak's own clock: MOPS x 512 flops / (1024 SIMDs x 1024 flops per cycle x cycles)
                e["issued_frac_clock_free"] = round(512.0 * mops / (SIMDS * 1024.0 * gui), 4)
        out[k] = e
    # families (template instances of one kernel together): time-weighted utilisation = sum of busy cycles / sum of SIMD cycles
    fams = {}
    for k, e in out.items():
        f = k.split("<")[0]
        if "duration_cycles" not in e:
            continue
        a = fams.setdefault(f, [0.0, 0.0, 0.0, 0])
        a[0] += e["SQ_VALU_MFMA_BUSY_CYCLES"] * e["launches"]
        a[1] += SIMDS * e["duration_cycles"] * e["launches"]
        a[2] += 512.0 * e.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0) * e["launches"]
        a[3] += e["launches"]
    out["__families__"] = {f: {"launches": a[3], "mfma_busy": round(a[0] / a[1], 4),
                               "issued_frac_clock_free": round(a[2] / (1024.0 * a[1]), 4) if a[2] else None}
                           for f, a in sorted(fams.items())}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for fn in sorted(glob.glob(os.path.join(root, "d3feat_amd", "csrc", "*.h*"))):
        h.update(os.path.basename(fn).encode())
        h.update(open(fn, "rb").read())
    out["__source_hash__"] = h.hexdigest()[:16]
    out["__timed_region_only__"] = region[0] > 0 and region[1] == region[0]
    out["__fragments_per_launch__"] = int(os.environ.get("D3F_PMC_FRAGMENTS_PER_LAUNCH", "4"))
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(sys.argv[1:] or ["."])
