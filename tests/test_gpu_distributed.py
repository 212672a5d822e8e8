"""The N > 1 launch path on hardware as far as ONE GPU allows: bench.py and tools/run_sharded.py started by
torch.distributed.run with one process and backend "nccl" (= RCCL): process-group initialisation, the histogram
all-reduce, the barrier + max-over-ranks timing and the final shard gather all execute their real code (the gather
degenerates to this rank's shard).  The multi-rank control flow itself is covered by the gloo tests on CPU."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _torchrun(args, timeout=900):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_port())] + args
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    return subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=env, timeout=timeout)


@pytest.mark.timeout(1000)
def test_bench_under_torchrun_nccl(device):
    r = _torchrun(["bench.py", "--gpus", "1", "--steps", "16", "--warmup", "4", "--no-cpu-baseline", "--no-instrument",
                   "--no-mirror-extra", "--no-pcie-extra"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["scaling"] == "weak"
    g = line["config"]["final_gather"]
    assert g["ranks"] == 1 and g["fragments_per_rank"] == [16] and g["rows_per_rank"][0] > 16 * 25000


@pytest.mark.timeout(1000)
def test_sharded_runner_under_torchrun_nccl(device, tmp_path):
    out = str(tmp_path / "out")
    r = _torchrun(["tools/run_sharded.py", "--synthetic", "5", "--raw-points", "60000", "--out", out, "--slots", "2", "--batch", "2"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["fragments"] == 5 and line["fragments_per_rank"] == [5] and len(line["limits"]) == 5
    assert len(os.listdir(os.path.join(out, "descriptors", "synthetic"))) == 5
