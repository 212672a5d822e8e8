"""`python bench.py --gpus N` must start N ranks itself (the driver invokes it without torchrun) and must refuse when the box has
fewer than N devices.  The launch path is driven here on CPU: world 2 over gloo through the same d3feat_amd.launch code."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def _run(args, env=None, timeout=300):
    e = dict(os.environ)
    e.pop("RANK", None)
    e.pop("WORLD_SIZE", None)
    e.update(env or {})
    return subprocess.run([sys.executable] + args, capture_output=True, text=True, cwd=ROOT, env=e, timeout=timeout)


@pytest.mark.timeout(300)
def test_self_launch_starts_two_ranks_over_gloo():
    r = _run(["tests/launch_probe.py", "--gpus", "2"], env={"D3F_LAUNCH_DEVICES": "2"})
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line == {"n_gpus": 2, "world_size": 2}


def test_self_launch_refuses_more_ranks_than_devices():
    r = _run(["tests/launch_probe.py", "--gpus", "2"], env={"D3F_LAUNCH_DEVICES": "1"})
    assert r.returncode != 0 and "refusing to run" in r.stderr and "--gpus 2" in r.stderr


def test_bench_gpus_2_without_devices_exits_non_zero():
    """The real bench.py on this GPU-less container: --gpus 2 -> clear message, non-zero exit, no JSON line."""
    r = _run(["bench.py", "--gpus", "2", "--steps", "4", "--warmup", "1"])
    assert r.returncode != 0 and "refusing to run" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_world_mismatch_is_refused():
    r = _run(["tests/launch_probe.py", "--gpus", "3"], env={"RANK": "0", "WORLD_SIZE": "1"})
    assert r.returncode != 0 and "refusing to run" in r.stderr
