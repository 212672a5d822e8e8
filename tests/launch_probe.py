"""Helper of tests/test_launch.py (not a test): a miniature of bench.py's start-up -- self-launch when --gpus N > 1 and no
launcher is around, then one gloo all_reduce over the ranks that were really started; rank 0 prints one JSON line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from d3feat_amd import launch   # noqa: E402


def main():
    n = int(sys.argv[sys.argv.index("--gpus") + 1])
    if launch.needs_launch(n):
        sys.exit(launch.relaunch(n, os.path.abspath(__file__), sys.argv[1:], what="(test) "))
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not launch.check_world(n, world, __file__):
        sys.exit(2)
    if "RANK" in os.environ:
        dist.init_process_group("gloo")
    t = torch.ones(1)
    if dist.is_initialized():
        dist.all_reduce(t)
    if int(os.environ.get("RANK", "0")) == 0:
        print(json.dumps({"n_gpus": int(t.item()), "world_size": dist.get_world_size() if dist.is_initialized() else 1}))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
