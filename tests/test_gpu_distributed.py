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


def _torchrun(args, timeout=900, nproc=1):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
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


@pytest.mark.timeout(1000)
@pytest.mark.parametrize("overlap", [0, 2])
def test_two_ranks_share_the_gpu_through_gloo(device, tmp_path, overlap):
    """Runner + FragmentEngine + shard exchange with MORE THAN ONE rank on hardware: two processes, both computing on GPU 0,
    collectives through gloo (RCCL refuses two ranks on one device, so the device-to-device transport itself stays untested on
    a 1-GPU box).  Seven fragments of different sizes (LPT shards of 4 and 3): the limits, the per-fragment files and the
    gathered shards equal the single-process run byte for byte / row for row -- also with the shards exchanged in asynchronous
    chunks of two fragments while they are produced."""
    import numpy as np
    common = ["tools/run_sharded.py", "--synthetic", "7", "--raw-points", "50000", "--slots", "2", "--batch", "2"]
    one, two = str(tmp_path / "one"), str(tmp_path / "two")
    r1 = _torchrun(common + ["--out", one])
    assert r1.returncode == 0, r1.stdout[-2000:] + r1.stderr[-3000:]
    r2 = _torchrun(common + ["--out", two, "--backend", "gloo", "--one-device", "--overlap-chunk", str(overlap)], nproc=2)
    assert r2.returncode == 0, r2.stdout[-2000:] + r2.stderr[-3000:]
    l1 = json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][-1])
    l2 = json.loads([l for l in r2.stdout.splitlines() if l.startswith("{")][-1])
    assert l2["world"] == 2 and sorted(l2["fragments_per_rank"]) == [3, 4] and l2["fallbacks"] == 0 and l1["fallbacks"] == 0
    assert l2["limits"] == l1["limits"]
    assert sum(l2["gathered_rows"]) == sum(l1["gathered_rows"])
    assert abs(sum(l2["gathered_checksum"]) - sum(l1["gathered_checksum"])) <= 1e-6 * abs(sum(l1["gathered_checksum"]))
    for sub in ("descriptors", "keypoints", "scores"):
        files = sorted(os.listdir(os.path.join(one, sub, "synthetic")))
        assert len(files) == 7 and files == sorted(os.listdir(os.path.join(two, sub, "synthetic")))
        for f in files:
            a, b = np.load(os.path.join(one, sub, "synthetic", f)), np.load(os.path.join(two, sub, "synthetic", f))
            assert a.shape == b.shape and np.array_equal(a, b), (sub, f)
