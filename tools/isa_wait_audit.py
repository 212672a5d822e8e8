#!/usr/bin/env python3
"""ISA audit: for every kernel of d3feat_amd/csrc, how many global stores / loads sit in a basic block that first drains the
memory counter (s_waitcnt vmcnt(0)) -- the signature of the compiler re-waiting conservatively inside predicated blocks
(one HBM round trip per store / load instead of pipelined accesses).  python tools/isa_wait_audit.py [file.hip ...]"""
import os, re, subprocess, sys
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "d3feat_amd", "csrc")
files = sys.argv[1:] or sorted(f for f in os.listdir(root) if f.endswith(".hip"))
print("%-52s %6s %8s %6s %8s %6s" % ("kernel", "stores", "st@wait0", "loads", "ld@wait0", "mfma"))
for f in files:
    asm = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-I../../include", "-S",
                          "--cuda-device-only", f, "-o", "-"], cwd=root, capture_output=True, text=True).stdout
    name, stats, waited = None, None, False
    def flush():
        if name and stats and (stats[0] or stats[2]):
            d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
            print("%-52s %6d %8d %6d %8d %6d" % (d[-52:], *stats))
    for line in asm.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            flush()
            name, stats, waited = m.group(1), [0, 0, 0, 0, 0], False
            continue
        if stats is None:
            continue
        t = line.strip()
        if t.startswith(".LBB") or t.startswith("s_cbranch") or t.startswith("s_branch") or t.startswith("; %bb"):
            waited = False
        elif t.startswith("s_waitcnt") and "vmcnt(0)" in t:
            waited = True
        elif t.startswith("global_store") or t.startswith("buffer_store"):
            stats[0] += 1; stats[1] += waited; waited = False
        elif t.startswith("global_load") or t.startswith("buffer_load"):
            stats[2] += 1; stats[3] += waited; waited = False
        elif t.startswith("v_mfma"):
            stats[4] += 1
        elif t.startswith("s_endpgm"):
            flush(); name = None; stats = None
