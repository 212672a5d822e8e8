"""The disassembly gate of csrc/gemm_x3.h (tools/isa_gate.py) as a CPU test: hipcc cross-compiles gfx950 without a GPU.  The
gate's own logic is checked on synthetic listings: it must flag scratch traffic, a hand-counted wait reached with a compiler-issued
load in flight, and leave the legitimate shapes alone."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("isa_gate", os.path.join(ROOT, "tools", "isa_gate.py"))
gate = importlib.util.module_from_spec(spec)
spec.loader.exec_module(gate)


def _body(text):
    return [l for l in text.strip("\n").splitlines()]


def test_gate_flags_the_three_failure_shapes():
    ok = _body("""
.LBB0_1:                                ; =>This Inner Loop Header: Depth=1
	;;#ASMSTART
	global_load_dwordx4 v[0:3], v[8:9], off
	;;#ASMEND
	v_add_u32_e32 v4, 1, v4
	;;#ASMSTART
	s_waitcnt vmcnt(6)
	;;#ASMEND
	global_store_dwordx4 v[10:11], v[0:3], off
	s_cbranch_scc1 .LBB0_1
	s_endpgm
""")
    bad, st = gate.check("k", ok)
    assert bad == [] and st["asm_loads"] == 1 and st["asm_waits"] == 1 and st["compiler_vmcnt0_in_inner_loops"] == 0
    spill = ok[:5] + ["\tscratch_load_dword v5, off, off offset:4"] + ok[5:]
    bad, st = gate.check("k", spill)
    assert st["scratch"] == 1 and any("scratch" in b for b in bad)
    foreign = ok[:5] + ["\tglobal_load_dword v7, v[12:13], off"] + ok[5:]
    bad, st = gate.check("k", foreign)
    assert st["foreign_loads_at_counted_waits"] == 1 and any("hand-counted" in b for b in bad)
    # the same foreign load followed by its own full drain is safe (and costs a drain: counted for rule 3)
    drained = ok[:5] + ["\tglobal_load_dword v7, v[12:13], off", "\ts_waitcnt vmcnt(0)"] + ok[5:]
    bad, st = gate.check("k", drained)
    assert bad == [] and st["compiler_vmcnt0_in_inner_loops"] == 1


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_current_sources_pass_the_gate():
    assert gate.main() == 0
