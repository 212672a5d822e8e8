"""RCCL device-to-device transport, self-validating the first time >= 2 GPUs are visible (VERDICT r04 missing 3 / next 7).

On the pool's 1-GPU boxes this SKIPS (reason printed); on a node it runs `python bench.py --gpus 2` (bench.py starts its own ranks:
d3feat_amd/launch.py), checks what rank 0 received over RCCL, and compares every rank's gathered shard with a SINGLE-process run that
regenerates that rank's fragments (--seed-rank r) under the job's neighbourhood limits (--limits): same row counts, same float64
checksum of the [xyz | desc | score] records.  Fragment independence: /root/reference utils/tester.py:196-229."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
LEAN = ["--no-cpu-baseline", "--no-instrument", "--no-mirror-extra", "--no-pcie-extra", "--no-latency", "--no-marginal", "--windows", "2"]


def _bench(args, detail, timeout=900):
    """-> (compact stdout line, full detail object)"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "bench.py"] + args + LEAN + ["--detail-out", str(detail)], capture_output=True, text=True, cwd=ROOT,
                       env=env, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1]), json.load(open(detail))


@pytest.mark.timeout(2400)
def test_two_gpus_gather_over_rccl_equals_single_process_runs(device, tmp_path):
    import torch
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip("RCCL device-to-device gather needs >= 2 GPUs; this box exposes %d (the gloo world-2 tests cover the control flow)" % n)
    two, _ = _bench(["--gpus", "2", "--steps", "16", "--warmup", "4"], tmp_path / "two.json")
    assert two["n_gpus"] == 2 and two["config"]["rccl"] == {"backend": "nccl", "world_size": 2}
    fg = two["config"]["final_gather"]
    assert fg["received_on_rank0"] == [True, True] and fg["fragments_per_rank"] == [16, 16] and fg["to"] == "rank 0"
    assert two["config"]["engine_fallbacks"] == 0 and all(r > 16 * 20000 for r in fg["rows_per_rank"])
    limits = ",".join(str(x) for x in two["config"]["neighborhood_limits"])
    for r in range(2):
        _, det = _bench(["--gpus", "1", "--steps", "16", "--warmup", "4", "--seed-rank", str(r), "--limits", limits], tmp_path / ("one%d.json" % r))
        g1 = det["config"]["final_gather"]
        assert g1["rows_per_rank"][0] == fg["rows_per_rank"][r], (r, g1["rows_per_rank"], fg["rows_per_rank"])
        a, b = g1["checksum_per_rank"][0], fg["checksum_per_rank"][r]
        assert abs(a - b) <= 1e-9 * max(abs(a), 1.0), (r, a, b)        # the same kernels on the same inputs: equal sums


@pytest.mark.timeout(1200)
def test_seed_rank_and_limits_reproduce_a_rank_in_one_process(device, tmp_path):
    """The comparison tool of the test above on ONE GPU: --seed-rank 1 generates rank 1's pool, --limits pins the limits; two
    runs give the same rows and checksum (deterministic kernels), another rank's pool gives different ones."""
    flags = ["--steps", "8", "--warmup", "2", "--limits", "40,40,44,48,46"]
    la, da = _bench(flags + ["--seed-rank", "1"], tmp_path / "a.json")
    lb, db = _bench(flags + ["--seed-rank", "1"], tmp_path / "b.json")
    lc, dc = _bench(flags + ["--seed-rank", "0"], tmp_path / "c.json")
    assert la["config"]["neighborhood_limits"] == [40, 40, 44, 48, 46] and la["detail"]
    ga, gb, gc = (d["config"]["final_gather"] for d in (da, db, dc))
    assert ga["rows_per_rank"] == gb["rows_per_rank"] and ga["fragments_per_rank"] == [8]
    assert abs(ga["checksum_per_rank"][0] - gb["checksum_per_rank"][0]) <= 1e-9 * abs(ga["checksum_per_rank"][0])
    assert gc["rows_per_rank"] != ga["rows_per_rank"] or gc["checksum_per_rank"] != ga["checksum_per_rank"]
