"""tools/pmc_summary.py and tools/pmc_mfma.py average per kernel over the launches of bench.py's TIMED REGION only (between the first
and the last d3f_trace_marker_kernel of a rocprofv3 counter CSV): the warm-up captures, the parity pass and the F = 1 latency replays of
the same process are other shapes of the same kernels and diluted the per-launch means of rounds 3-5."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COLS = ["Correlation_Id", "Dispatch_Id", "Agent_Id", "Queue_Id", "Process_Id", "Thread_Id", "Grid_Size", "Kernel_Id", "Kernel_Name",
        "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Counter_Name",
        "Counter_Value", "Start_Timestamp", "End_Timestamp"]


def write_csv(path, launches, counters):
    """launches: [(kernel name, {counter: value}, duration ns)] in dispatch order."""
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(COLS)
        t = 1000
        for i, (name, vals, dur) in enumerate(launches, 1):
            for c in counters:
                w.writerow([i, i, "Agent 2", 1, 1, 1, 256, 7, name, 256, 0, 0, 8, 0, 32, c, float(vals.get(c, 0.0)), t, t + dur])
            t += dur + 10


MARK = "d3f_trace_marker_kernel(int)"
GEMM = "void gemm_x3_kernel<4, 8>(float const*, int)"


def run(tool, d, env=None):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), d], capture_output=True, text=True,
                       env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stderr
    return json.loads(r.stdout)


def test_hbm_traffic_is_the_mean_over_the_timed_region(tmp_path):
    small, big = {"FETCH_SIZE": 10.0}, {"FETCH_SIZE": 1000.0}
    seq = [(GEMM, small, 5)] * 3 + [(MARK, {}, 1)] + [(GEMM, big, 50)] * 4 + [(MARK, {}, 1)] + [(GEMM, small, 5)] * 9
    write_csv(str(tmp_path / "pmc_fetch" / "run" / "1_counter_collection.csv"), seq, ["FETCH_SIZE"])
    smallw, bigw = {"WRITE_SIZE": 1.0}, {"WRITE_SIZE": 100.0}
    seqw = [(GEMM, smallw, 5)] * 3 + [(MARK, {}, 1)] + [(GEMM, bigw, 50)] * 4 + [(MARK, {}, 1)] + [(GEMM, smallw, 5)] * 9
    write_csv(str(tmp_path / "pmc_write" / "run" / "2_counter_collection.csv"), seqw, ["WRITE_SIZE"])
    out = run("pmc_summary.py", str(tmp_path))
    e = out["gemm_x3_kernel<4, 8>"]
    assert e["launches"] == 4
    assert e["traffic_bytes_per_launch"] == 2 * 1024 * 1000 + 1024 * 100      # FETCH_SIZE x 2 on gfx950, KiB
    assert out["__timed_region_only__"] is True
    assert "d3f_trace_marker_kernel" not in out


def test_without_markers_every_launch_counts_and_the_file_says_so(tmp_path):
    write_csv(str(tmp_path / "p" / "1_counter_collection.csv"), [(GEMM, {"FETCH_SIZE": 10.0}, 5), (GEMM, {"FETCH_SIZE": 30.0}, 5)],
              ["FETCH_SIZE"])
    out = run("pmc_summary.py", str(tmp_path))
    assert out["gemm_x3_kernel<4, 8>"]["launches"] == 2 and out["__timed_region_only__"] is False


def test_matrix_pipe_utilisation_of_the_timed_region(tmp_path):
    # 1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs: 8 x 1000 cycles; busy 256000 SIMD-cycles -> 0.25
    hot = {"SQ_VALU_MFMA_BUSY_CYCLES": 256000.0, "GRBM_GUI_ACTIVE": 8000.0, "SQ_INSTS_VALU_MFMA_MOPS_BF16": 512000.0}
    cold = {"SQ_VALU_MFMA_BUSY_CYCLES": 100.0, "GRBM_GUI_ACTIVE": 8000.0, "SQ_INSTS_VALU_MFMA_MOPS_BF16": 200.0}
    seq = [(GEMM, cold, 400)] * 5 + [(MARK, {}, 1)] + [(GEMM, hot, 400)] * 2 + [(MARK, {}, 1)] + [(GEMM, cold, 400)] * 20
    write_csv(str(tmp_path / "pmc_mfma" / "1_counter_collection.csv"), seq, sorted(hot))
    out = run("pmc_mfma.py", str(tmp_path), {"D3F_PMC_FRAGMENTS_PER_LAUNCH": "12"})
    e = out["gemm_x3_kernel<4, 8>"]
    assert e["launches"] == 2 and abs(e["mfma_busy"] - 0.25) < 1e-4
    assert abs(e["issued_frac_clock_free"] - 512.0 * 512000.0 / (1024 * 1024.0 * 1000.0)) < 1e-4     # = 0.25: the two counters agree
    assert out["__families__"]["gemm_x3_kernel"]["launches"] == 2
    assert out["__timed_region_only__"] is True and out["__fragments_per_launch__"] == 12
