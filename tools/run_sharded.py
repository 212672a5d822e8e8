#!/usr/bin/env python3
"""Descriptor extraction for a folder of fragments, sharded over the GPUs of one node (d3feat_amd/runner.py):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/run_sharded.py \
        --fragments data/3DMatch/fragments --out geometric_registration/D3Feat_run --snapshot results/Log_x/snapshots/snap-54

--fragments DIR: <DIR>/<scene>/cloud_bin_<k>.ply (the layout of datasets/ThreeDMatch.py:326-366); --synthetic N generates N
room fragments instead (no data set on this machine).  Every rank writes the reference's per-fragment files for its share
(utils/tester.py:215-229 layout under --out); the final all_gather of the shards is done unless --no-gather.
Works with one process too (plain `python tools/run_sharded.py ...`).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fragments", default=None)
    ap.add_argument("--synthetic", type=int, default=0)
    ap.add_argument("--raw-points", type=int, default=300000)
    ap.add_argument("--out", default="sharded_out")
    ap.add_argument("--snapshot", default=None)
    ap.add_argument("--config", default=None)
    ap.add_argument("--slots", type=int, default=4)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--gather-to", default="0", help="rank that receives the shards (default 0: gather at the end), or 'all'")
    ap.add_argument("--overlap-chunk", type=int, default=0,
                    help="exchange the shards in asynchronous chunks of this many fragments while they are produced (0: one "
                         "all_gather at the end)")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                    help="gloo: collectives through host memory -- with --one-device, several ranks share GPU 0 (RCCL refuses "
                         "two ranks on one device): the way to run runner + engine + gather with > 1 rank on a 1-GPU box")
    ap.add_argument("--one-device", action="store_true", help="every rank computes on cuda:0")
    a = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if a.one_device else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if "RANK" in os.environ:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "gloo":
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)
    from d3feat_amd import runner
    from d3feat_amd.models.variables import build_variables
    from d3feat_amd.utils.config import Config, threedmatch_config
    from d3feat_amd.utils.ply import read_ply_xyz, ply_vertex_count
    from d3feat_amd.utils.synthetic import room_fragment
    from d3feat_amd.utils.tf_checkpoint import load_checkpoint
    cfg = threedmatch_config()
    if a.config:
        cfg = Config()
        cfg.load(a.config)
    W = load_checkpoint(a.snapshot) if a.snapshot else build_variables(cfg, seed=42).values
    if a.synthetic:
        rng = np.random.default_rng(0)
        ids = ["synthetic/cloud_bin_%d.ply" % i for i in range(a.synthetic)]
        sizes = [int(x) for x in rng.integers(a.raw_points // 2, a.raw_points * 3 // 2, a.synthetic)]

        def load(i):
            return room_fragment(i, n_raw=sizes[i], edge=1.68 * float(np.sqrt(sizes[i] / 300000.0)))
    else:
        ids = []
        for scene in sorted(os.listdir(a.fragments)):
            d = os.path.join(a.fragments, scene)
            if os.path.isdir(d):
                for f in sorted((f for f in os.listdir(d) if f.endswith(".ply")), key=lambda x: int(x[:-4].split("_")[-1])):
                    ids.append(scene + "/" + f)
        sizes = [ply_vertex_count(os.path.join(a.fragments, i)) for i in ids]

        def load(i):
            return read_ply_xyz(os.path.join(a.fragments, ids[i]))
    t0 = time.perf_counter()
    res = runner.run_sharded(ids, sizes, load, cfg, W, runner.gpu_engine_factory(a.slots, a.batch), runner.gpu_calibrate(cfg), dev,
                             gather=not a.no_gather, save=runner.save_records_3dmatch(a.out), overlap_chunk=a.overlap_chunk,
                             dst=None if a.gather_to == "all" else int(a.gather_to))
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    dst_rank = 0 if a.gather_to == "all" else int(a.gather_to)
    if rank == dst_rank:                 # the rank that holds the shards reports (--gather-to r: the others received None)
        print(json.dumps({"fragments": len(ids), "world": world, "seconds": round(dt, 3), "limits": [int(x) for x in res["limits"]],
                          "fragments_per_rank": [len(o) for o in res["order"]], "fallbacks": res["fallbacks"],
                          "gathered_rows": [int(s[0].shape[0]) for s in res["shards"] if s[0] is not None] if res["shards"] else None,
                          "gathered_checksum": ([round(float(s[0].double().sum().item()), 3) for s in res["shards"] if s[0] is not None]
                                                if res["shards"] else None)}))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
