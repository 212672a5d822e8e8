"""bench.py's driver-facing stdout line (a 20 KB line was cut by the driver's 8 KB tail, BENCH_r04.parsed = null).

The compact line is built from the full result object by `bench.compact_line`; here it is rebuilt from RECORDED full objects of round 4
(profiles/r04_*_bench*.json: the 20 KB lines themselves) and from a mocked 8-rank job, and must stay strict JSON below 4 KB with every
key the contract and the judge read.  No GPU, no torch."""
import copy
import importlib.util
import json
import os

import pytest

from conftest import ROOT


def _bench():
    spec = importlib.util.spec_from_file_location("bench_for_line_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


RECORDED = ["profiles/r04_v67_bench_k20.json", "profiles/r04_v67_bench.json", "profiles/r04_v62_bench_demo.json",
            "profiles/r04_v62_bench_config4.json", "profiles/r04_v62_bench_bf16_features.json"]
CONTRACT = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline", "cpu_baseline", "parity", "latency_ms"]


def _strict(line):
    def bad(x):
        raise ValueError("non-finite constant %s in the line" % x)
    return json.loads(line, parse_constant=bad)


@pytest.mark.parametrize("path", [p for p in RECORDED if os.path.exists(os.path.join(ROOT, p))])
def test_compact_line_of_recorded_runs(path):
    b = _bench()
    full = json.load(open(os.path.join(ROOT, path)))
    assert len(json.dumps(full)) > 8192          # the recorded object really is one of the lines the driver could not keep
    line = b.compact_line(full, "gpurun_out/bench_detail.json")
    assert "\n" not in line and len(line.encode()) < 4096 and len(line.encode()) <= b.COMPACT_MAX_BYTES
    obj = _strict(line)
    for k in CONTRACT:
        assert k in obj, k
    assert obj["value"] == full["value"] and obj["ms_per_step"] == full["ms_per_step"] and obj["n_gpus"] == 1
    assert obj["dtype"] in ("f32", "bf16") and obj["config"]["workload"] and "model" not in obj["config"]
    r = obj["roofline"]
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "alg_flops_per_launch", "avg_launch_us"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = obj["cpu_baseline"]
    assert c["value"] > 0 and c["cores"] >= 1 and c["kind"] in ("port", "reference") and c["sample"]
    assert obj["parity"]["ok"] is True and obj["parity"]["desc_max_abs"] <= obj["parity"]["tolerance"]
    assert obj["latency_ms"]["median"] > 0 and obj["detail"] == "gpurun_out/bench_detail.json"
    assert abs(obj["vs_cpu_baseline"] - obj["value"] / c["value"]) <= 1e-2 * obj["vs_cpu_baseline"]


def test_compact_line_of_a_mocked_eight_rank_job_and_of_nan():
    b = _bench()
    full = json.load(open(os.path.join(ROOT, "profiles/r04_v67_bench_k20.json")))
    full = copy.deepcopy(full)
    full["n_gpus"] = 8
    full["value"] *= 8
    full["cpu_baseline"] = full["parity"] = None              # N > 1: no CPU leg
    full.pop("vs_cpu_baseline", None)
    full["config"]["rccl"] = {"backend": "nccl", "world_size": 8}
    full["config"]["parallelism"] = "fragment-dp8"
    rows = [20 * 29123 + i for i in range(8)]
    full["config"]["final_gather"] = {"ranks": 8, "to": "rank 0", "received_on_rank0": [True] * 8, "fragments_per_rank": [20] * 8,
                                      "rows_per_rank": rows, "bytes_per_rank": [r * 144 for r in rows], "what": "x" * 400}
    full["roofline"]["traffic"] = float("nan")                # a counter file gone wrong must not break the line
    full["timing"]["window_ms"] = [13.5] * 9
    line = b.compact_line(full, None)
    assert len(line.encode()) < 4096
    obj = _strict(line)
    fg = obj["config"]["final_gather"]
    assert obj["n_gpus"] == 8 and fg["received_on_rank0"] == [True] * 8 and fg["fragments_per_rank"] == [20] * 8
    assert fg["rows_total"] == sum(rows) and fg["bytes_total"] == 144 * sum(rows) and obj["config"]["rccl"]["world_size"] == 8
    assert obj["roofline"]["traffic"] is None and obj["cpu_baseline"] is None and obj["parity"] is None


def test_detail_file_round_trip(tmp_path):
    b = _bench()
    full = json.load(open(os.path.join(ROOT, "profiles/r04_v67_bench_k20.json")))
    p = b.write_detail(full, str(tmp_path / "sub" / "bench_detail.json"))
    assert p is not None
    back = json.load(open(tmp_path / "sub" / "bench_detail.json"))
    assert back["rooflines"] == full["rooflines"] and back["value"] == full["value"]
    assert b.write_detail(full, "/proc/definitely/not/writable/x.json") is None
