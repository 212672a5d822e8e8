"""Self-launch of the one-process-per-GPU job: `python bench.py --gpus N` (no torchrun around it) re-executes itself under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>` so that N ranks
really start -- and refuses, loudly, when the box has fewer than N devices ("discover torch.cuda.device_count()
at run time ... never extrapolate").  Under torchrun (RANK set) nothing is launched; `check_world` then verifies that the world
the launcher made is the one the command line asked for."""
import os
import socket
import subprocess
import sys


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def device_count():
    """Devices this process may use.  D3F_LAUNCH_DEVICES overrides the probe (CPU tests of the launch path: gloo ranks need no GPU)."""
    env = os.environ.get("D3F_LAUNCH_DEVICES")
    if env is not None:
        return int(env)
    import torch
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def needs_launch(n):
    return int(n) > 1 and "RANK" not in os.environ


def relaunch(n, script, argv, what="GPU"):
    """Run `script argv` as n ranks on this node; -> the launcher's exit code (non-zero with a clear message when the box has
    fewer than n devices)."""
    have = device_count()
    if have < n:
        sys.stderr.write("%s: --gpus %d requested but this box has %d %s device(s): refusing to run (a %d-rank result on fewer "
                         "devices would not be a %d-GPU measurement)\n" % (os.path.basename(script), n, have, what, n, n))
        return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(n)), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), script] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
    return subprocess.call(cmd, env=env)


def check_world(n, world, script="bench.py"):
    """Under a launcher: the number of ranks must be what the command line says (n_gpus in the result = what RCCL saw)."""
    if int(n) != int(world):
        sys.stderr.write("%s: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE): refusing to run\n"
                         % (os.path.basename(script), n, world))
        return False
    return True
