#!/usr/bin/env python3
"""Build-time disassembly gate for the hand-counted memory pipeline of csrc/gemm_x3.h.

The operand-split contraction kernels request their operands through inline asm (`global_load_dwordx4`, LDS-DMA) and wait for
them with hand-counted `s_waitcnt vmcnt(N)` statements: the compiler neither sees those loads nor pads for them, so the scheme
is only correct / fast while the generated code keeps three properties.  This script disassembles gemm_f32.hip for gfx950 and
checks them for every `gemm_x3*` kernel; `__graft_entry__.build()` runs it and fails the build when one is violated:

  1. no scratch traffic at all (`scratch_load` / `scratch_store`, ScratchSize 0): a spilled register is reloaded by a load the
     compiler waits for with vmcnt(0) -- round 5 found one that silently drained the prefetch pipeline once per tile;
  2. no hand-counted `s_waitcnt vmcnt(N > 0)` is reached with a compiler-issued vector LOAD possibly in flight (one issued since
     the last full drain): loads retire in order, a foreign load issued after ours would be counted in our place and the wait
     could return early.  (Stores may be in flight: they only make the wait conservative.  A compiler load that is followed by
     its own vmcnt(0) -- the optional row-index gather -- is safe, and costs a drain.)
  3. the number of compiler-issued `s_waitcnt vmcnt(0)` inside the innermost loops does not GROW against the recorded build
     (tools/isa_gate_baseline.json; `--update` rewrites it): each one drains every operand request in flight.  The tile kernels
     have none; the resident-W kernels carry the ones of their optional paths (row gather, residual) and of the join after them.

    python tools/isa_gate.py            # exit code 0: all kernels pass; 1: a violation (printed)
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "d3feat_amd", "csrc")
VMEM_LOAD = ("global_load", "buffer_load", "scratch_load", "flat_load")


def disassemble(src="gemm_f32.hip"):
    return subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S",
                           "--cuda-device-only", src, "-o", "-"], cwd=CSRC, capture_output=True, text=True, check=True).stdout


def kernels(asm):
    name, body = None, []
    for line in asm.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, body = m.group(1), []
            continue
        if name is None:
            continue
        body.append(line)
        if line.strip().startswith("s_endpgm"):
            yield name, body
            name = None


def check(name, body):
    """-> (list of violations, summary dict)"""
    bad = []
    in_asm = False
    inner, inner_id = False, None     # inside an innermost loop (and its header's label)
    foreign = 0              # compiler-issued vector loads since the last full drain (vmcnt(0) of either origin)
    stats = dict(asm_loads=0, asm_waits=0, foreign_loads_at_counted_waits=0, compiler_vmcnt0_in_inner_loops=0, scratch=0)
    for ln, line in enumerate(body):
        t = line.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            if "Inner Loop Header" in line:
                inner_id, inner = m.group(1)[1:], True          # "LBBx_y" (the annotations drop the leading dot)
            else:
                mm = re.search(r"in Loop: Header=(BB\d+_\d+)", line)
                inner = bool(mm) and inner_id is not None and ("L" + mm.group(1)) == inner_id
            continue
        if t.startswith("scratch_"):
            stats["scratch"] += 1
            bad.append("%s: scratch access `%s`" % (name, t))
        if in_asm:
            if t.startswith("global_load") or t.startswith("buffer_load"):
                stats["asm_loads"] += 1
            elif t.startswith("s_waitcnt") and "vmcnt" in t:
                stats["asm_waits"] += 1
                if "vmcnt(0)" in t:
                    foreign = 0
                elif foreign > 0:
                    # a hand-counted wait with compiler loads possibly in flight: they are counted in place of our requests
                    stats["foreign_loads_at_counted_waits"] += 1
                    bad.append("%s: hand-counted `%s` with %d compiler-issued load(s) possibly in flight (line %d)" % (name, t, foreign, ln))
            continue
        if t.startswith(VMEM_LOAD):
            foreign += 1
        if t.startswith("s_waitcnt") and "vmcnt(0)" in t:
            foreign = 0
            if inner:
                stats["compiler_vmcnt0_in_inner_loops"] += 1
    return bad, stats


def main():
    """--update: rewrite the baseline of rule 3 from the current build."""
    import json
    base_fn = os.path.join(ROOT, "tools", "isa_gate_baseline.json")
    baseline = json.load(open(base_fn)) if os.path.exists(base_fn) else {}
    asm = disassemble()
    failures, seen, now = [], 0, {}
    for name, body in kernels(asm):
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void ", "")
        if "gemm_x3" not in dem:
            continue
        seen += 1
        bad, st = check(dem, body)
        now[dem] = st["compiler_vmcnt0_in_inner_loops"]
        allowed = baseline.get(dem)
        print("%-28s asm loads %3d  asm waits %3d  scratch %d  counted waits with foreign loads in flight %d  compiler vmcnt(0) in "
              "innermost loops %d (baseline %s)" % (dem[-28:], st["asm_loads"], st["asm_waits"], st["scratch"],
                                                    st["foreign_loads_at_counted_waits"], st["compiler_vmcnt0_in_inner_loops"], allowed))
        failures += bad
        if allowed is not None and st["compiler_vmcnt0_in_inner_loops"] > allowed:
            failures.append("%s: %d compiler-issued `s_waitcnt vmcnt(0)` inside innermost loops, the recorded build has %d: a new one "
                            "drains the operand requests in flight (look for a spill or a new compiler-visible load in the walk)"
                            % (dem, st["compiler_vmcnt0_in_inner_loops"], allowed))
    if "--update" in sys.argv:
        json.dump(now, open(base_fn, "w"), indent=1, sort_keys=True)
        print("baseline written:", base_fn)
    if seen == 0:
        failures.append("no gemm_x3 kernel found in the disassembly")
    for b in failures:
        print("GATE:", b)
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
