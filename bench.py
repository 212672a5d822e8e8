#!/usr/bin/env python3
"""D3Feat hot-path benchmark (BASELINE.json: fragments/sec on 30k-pt clouds + ms/KPConv-layer).

One "step" = one synthetic 3DMatch-shaped fragment end to end on the GPU, exactly the work the reference does per
sess.run plus the stage-0 voxelisation:
    raw cloud (300k pts, resident in HBM) -> grid subsample @0.03 m (~30k pts) -> stacked with itself (the
    reference's test generators feed every fragment as a self-pair, datasets/ThreeDMatch.py:190-192) ->
    5-level pyramid (13 radius searches + 4 grid subsamplings) -> KPFCNN forward (10 KPConv + 28 unary) ->
    32-d descriptors + detection scores in HBM, kept as one [xyz | desc | score] record block per fragment.
Multi-GPU: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...; fragments are sharded across
ranks (weak scaling: K fragments per rank); every rank keeps its WHOLE shard's records in HBM and the shards are
all-gathered over RCCL once at the end of the timed region (the path's only data collective).

Prints ONE JSON line (rank 0).  Extra objects:
  parity        the engine's output (same F, slots, graph path as the timed region) for the fragments the CPU leg pushes
                through the oracle: points / level-0 neighbour indices bit-equal, descriptors / scores max |diff|;
                the process exits non-zero (after printing the line) when a bound is exceeded;
  roofline      the dominant kernel family (largest share of GPU time among the timed launches), HIP-event timed on the
                launch stream in a separate instrumented pass over the same fragments;
  rooflines     the same for every timed kernel family (HBM GB/s of the gather kernels, TFLOP/s of the contractions);
  kpconv_layers ms per KPConv layer (aggregation + contraction kernels);
  cpu_baseline  the same step on the host: reference C++ (oracle/_ref, 1 thread) when available, else the C
                restatement, for the geometry; torch-CPU restatement of the TF graph for the network (N=1 only).
"""
import argparse
import glob
import hashlib
import json
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# The engine keeps several fragments in flight on separate HIP streams; by default the ROCm runtime multiplexes all
# streams of a process onto 4 hardware queues.  Must be set before the HIP runtime initialises (i.e. before torch).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# multi-process GPU work on this driver needs dmabuf IPC (RCCL / shared device tensors fail with `hipIpcGetMemHandle: invalid
# argument` otherwise): set here too, not only in launch.relaunch, so a user's own `torchrun bench.py` gets the same environment
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md)
MFMA_F32_PEAK_TF = 157.3   # v_mfma_f32_32x32x2_f32 dense peak
MFMA_BF16_PEAK_TF = 2516.6  # v_mfma_f32_32x32x16_bf16 dense peak (16 x the fp32 matrix rate; MI355X_MICROARCH.md: "~2.5 PF dense")
# the operand-split contraction (csrc/gemm_x3.h) issues SIX bf16 products per fp32 product: the rate at which its instruction
# stream could at best deliver fp32 products is a sixth of the bf16 peak -- 2.67 x the fp32 matrix pipe's
X3_PRODUCTS = 6
MFMA_X3_PEAK_TF = MFMA_BF16_PEAK_TF / X3_PRODUCTS
PARITY_TOL = 1e-4          # BASELINE.json north_star: descriptors and scores within 1e-4 (absolute), indices bit-exact
BF16_TOL = 1.5e-2          # documented tolerance of the bf16 configurations vs the fp32 oracle (tests/test_gpu_bf16.py; measured on this
                           # sample: descriptors 6.2e-3, scores 1.27e-2)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=512)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU leg (and with it the parity check)")
    ap.add_argument("--cpu-fragments", type=int, default=5, help="fragments of the CPU / parity sample")
    ap.add_argument("--no-cpu-1thread", action="store_true", help="skip the 1-thread network row of the CPU baseline")
    ap.add_argument("--pool", type=int, default=4, help="distinct synthetic fragments per rank (cycled)")
    ap.add_argument("--slots", type=int, default=0, help="replays in flight per GPU (HIP-graph slots on separate streams); 0 = 4, or "
                                                         "3 for a job of fewer than 64 steps (one round of replays: r04 x9)")
    ap.add_argument("--batch", type=int, default=0,
                    help="fragments stacked into one graph replay (FragmentEngine(batch=F)); a step is still ONE fragment.  "
                         "0 = by job size: 12 from 384 fragments on (1591 / 1586 against 1558 / 1578 fragments/s at 8 and 1569 / 1570 at 16, "
                         "profiles/r03_experiments.txt x22), 8 from 64 (96 fragments: 1540 against 1475 at 4), or for a job shorter "
                         "than 64 fragments ceil(steps / slots) (at most 8) so that the "
                         "whole job is one round of replays, all in flight together, instead of a round plus a lone straggler")
    ap.add_argument("--cap-factor", type=float, default=1.1,
                    help="voxel capacity of a slot = this x the largest fragment of the pool.  Launches are sized by capacity, so slack "
                         "costs idle workgroups (1.3 / 0.4 instead of 1.1 / 0.32: -2 %% fragments/s, profiles/r02_experiments.txt); a "
                         "fragment beyond a capacity is flagged on the device and recomputed eagerly (engine_fallbacks in the line)")
    ap.add_argument("--level-ratio", type=float, default=0.32,
                    help="row capacity of pyramid level l+1 / level l (surface clouds keep 0.27-0.30 per level)")
    ap.add_argument("--eager", action="store_true", help="op-by-op eager path instead of the graph engine")
    ap.add_argument("--mirror", action="store_true",
                    help="headline run with the self-pair computed once and mirrored (default: the full stacked pair)")
    ap.add_argument("--no-mirror-extra", action="store_true", help="skip the secondary mirrored measurement")
    ap.add_argument("--no-pcie-extra", action="store_true", help="skip the secondary PCIe-inclusive measurement")
    ap.add_argument("--no-instrument", action="store_true", help="skip the per-launch HIP-event pass (clean rocprof runs)")
    ap.add_argument("--ablate", default="", help="MEASUREMENT TOOL, results invalid: comma list of op families whose library calls "
                    "are skipped (gemm, kpconv, maxpool, head, rowpos): the replay keeps its shape (sizes are device-resident and "
                    "do not depend on feature values), so the throughput difference is that family's cost in the concurrent regime")
    ap.add_argument("--bf16", action="store_true",
                    help="BASELINE configs[4] (a SEPARATE configuration, never the fp32 headline): unary / unfused KPConv "
                         "contractions with bf16 operands and fp32 accumulation; use with --batch 8 --slots 2")
    ap.add_argument("--bf16-features", action="store_true",
                    help="BASELINE configs[4] in full: bf16 contraction AND the activations between the layers stored as bfloat16 "
                         "(implies --bf16; still a separate configuration)")
    ap.add_argument("--no-marginal", action="store_true", help="skip the marginal-cost measurements (extra engines with one op "
                    "family skipped each)")
    ap.add_argument("--config3", action="store_true",
                    help="SURVEY §8d config #3's workload shape instead of config #2's uniform fragments: a pool of 16 fragments per "
                         "rank whose sizes are drawn from U(15 k, 45 k) points after the 0.03 m subsample (raw points and room edge "
                         "scaled to keep the sampling density); capacities are those of the largest fragment, so the smaller ones "
                         "run with capacity slack.  A SEPARATE configuration (never the config #2 headline)")
    ap.add_argument("--config4", action="store_true",
                    help="BASELINE configs[3] (a SEPARATE configuration): KITTI-like stacks of two DIFFERENT synthetic LiDAR sweeps "
                         "(~120k raw points each, datasets/KITTI.py:94-106,268-337), dl = 0.3 m, the reference's real trained KITTI tensors "
                         "(tests/golden/kitti_epoch61_weights.npz) where its dump has them; a step is one stack = 2 frames")
    ap.add_argument("--demo", action="store_true",
                    help="BASELINE configs[0] (a SEPARATE configuration): the reference's demo pair (demo_data/cloud_bin_0/1.ply after ITS "
                         "0.03 m grid subsample = tests/golden/demo_bin{0,1}_sub003.npy), each cloud fed as a self-pair like "
                         "demo_registration.py:30-95 does; no stage-0 voxelisation inside the step (the script subsamples before the "
                         "dataset sees the cloud, demo_registration.py:24)")
    ap.add_argument("--windows", type=int, default=9,
                    help="the K-step job is timed this many times back to back (each window bracketed by barrier + synchronize); "
                         "`value` is the MEDIAN window (a 20-fragment window lasts 14 ms: single windows spread 1290..1430 fragments/s "
                         "across boxes, profiles/r03_experiments.txt x30/x34), all samples, p10 / p90 are in the line")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-fragment latency measurement (F = 1, one slot)")
    ap.add_argument("--gather-to", default="0", help="N > 1: the rank that receives the shards at the end (default 0: north_star's "
                    "\"gather of descriptors only at the end\"), or 'all' (all_gather: every rank receives every shard)")
    ap.add_argument("--gather-chunk", type=int, default=8,
                    help="N > 1: fragments per asynchronous shard-exchange chunk (parallel.ShardCollector overlapped mode: the "
                         "shards cross xGMI while the next fragments are computed); 0 = one all_gather of the whole shard at the end")
    ap.add_argument("--schedule", default="", help="comma list of replay sizes for ONE window (sum = --steps), submitted round-robin over "
                    "the slots: a short job has no steady state, so an uneven split (a small first replay reaches the network phase "
                    "while the others are still in the latency-bound geometry chain) changes its wall time; engine batch = the largest")
    ap.add_argument("--seed-rank", type=int, default=-1, help="generate the synthetic pool of THIS rank of a larger job (default: the "
                    "process's own rank): a single process reproduces what rank r of an N-rank run computed (tests/test_gpu_multi.py)")
    ap.add_argument("--limits", default="", help="comma list: neighbourhood limits to use instead of calibrating (to repeat a multi-rank "
                    "run's limits, which are calibrated over all ranks' pools, in one process)")
    ap.add_argument("--detail-out", default=os.path.join(ROOT, "gpurun_out", "bench_detail.json"),
                    help="file that receives the FULL result object (every family's roofline, per-launch tables, marginal costs, "
                         "secondary measurements); stdout carries the compact line only")
    ap.add_argument("--detail-stdout", action="store_true", help="also print the full object as an EARLIER stdout line")
    ap.add_argument("--raw-points", type=int, default=300000, help="raw points per synthetic fragment (config #2: 300k)")
    ap.add_argument("--kitti-points", type=int, default=120000, help="raw points per synthetic LiDAR sweep (config #4: 120k)")
    ap.add_argument("--edge", type=float, default=1.68, help="room edge in metres (config #2: 1.68 -> ~30k pts at 0.03 m)")
    return ap.parse_args()


class Step:
    """The hot path for one stack, op by op (instrumented pass, --eager)."""

    def __init__(self, cfg, model, limits, device, bf16=False, bf16_features=False, two=False, stage0=True):
        from d3feat_amd.datasets.common import FragmentDataset
        self.cfg, self.model, self.device, self.bf16 = cfg, model, device, bool(bf16)
        self.bf16_features = bool(bf16_features)
        self.two, self.stage0 = bool(two), bool(stage0)
        self.ds = FragmentDataset([], fast=True)
        self.ds.neighborhood_limits = limits
        self.ds.stack_group = 2
        self.map = self.ds.get_tf_mapping(cfg)

    def __call__(self, raw_dev):
        """raw_dev: one item, or a list of F items stacked like FragmentEngine(batch=F): [c_1; c_1; c_2; c_2; ...] for self-pairs,
        [a_1; b_1; a_2; b_2; ...] when an item is a pair of different clouds (two=True)."""
        import torch
        from d3feat_amd import ops
        from d3feat_amd import tf_custom_ops as tfo
        items = raw_dev if isinstance(raw_dev, list) else [raw_dev]
        clouds = [c for it in items for c in (it if self.two else (it,))]
        subs = [tfo.grid_subsampling(r, self.cfg.first_subsampling_dl) if self.stage0 else r for r in clouds]   # stage 0
        stack = subs if self.two else [x for s in subs for x in (s, s)]                      # self-pairs (device copies)
        pts = torch.cat(stack, 0)
        lens = ops.as_lens([int(s.shape[0]) for s in stack], self.device)
        flat = self.map(pts, None, None, None, lens, ("a", "a"), pts)
        with ops.bf16_contraction(self.bf16, features=self.bf16_features):
            desc, score = self.model.run(flat)
        return ops.pack_descriptors(pts, desc, score)


def kpconv_alg_bytes(Nq, Ns, K, Cin, Cout):
    """algorithmic bytes of one KPConv layer."""
    return 4 * (3 * Nq + 3 * Ns + Nq * K + Ns * Cin + 45 + 15 * Cin * Cout + Nq * Cout)


def kpconv_flops(Nq, K, Cin, Cout):
    """flops_gemm + flops_agg of one KPConv layer."""
    return 2.0 * Nq * 15 * Cin * Cout, 2.0 * Nq * 15 * K * Cin + 11.0 * Nq * K * 15


def contraction_description():
    """What the contractions of this run are, from the switches that route them (ops.GEMM_X3 / X3_N32 / KP_X3 / KP_MFMA)."""
    from d3feat_amd import ops
    split = ("EXACT operand splitting (each f32 operand = 3 bf16 planes, 6 exact bf16 products per f32 product, f32 accumulate on "
             "v_mfma_f32_32x32x16_bf16 / 16x16x32_bf16: csrc/gemm_x3.h; error vs float64 below the f32 MFMA kernel's, "
             "tests/test_gpu_gemm_x3.py)")
    parts = []
    if ops.GEMM_X3:
        parts.append("unary contractions%s by %s" % ("" if getattr(ops, "X3_N32", True) else " wider than 32 columns", split))
    else:
        parts.append("unary contractions f32 on v_mfma_f32_32x32x2_f32 (D3F_GEMM_X3=0)")
    parts.append("the fused KPConv kernels (Cin 32 / 64 / 128) contract their LDS tile %s" %
                 ("in the same operand-split form" if ops.KP_X3 else "on v_mfma_f32_16x16x4_f32 / 32x32x2_f32 (D3F_KP_X3=0)"))
    if getattr(ops, "KP_MFMA", False):
        parts.append("level-0 KPConv aggregation on v_mfma_f32_16x16x1_4b_f32 (D3F_KP_MFMA=1)")
    return "f32 in / f32 out; " + "; ".join(parts)


def source_hash():
    """sha256 over the kernel sources: ties a committed counter file (tools/pmc_summary.py stamps it) to the code it measured."""
    h = hashlib.sha256()
    for fn in sorted(glob.glob(os.path.join(ROOT, "d3feat_amd", "csrc", "*.h*"))):
        h.update(os.path.basename(fn).encode())
        h.update(open(fn, "rb").read())
    return h.hexdigest()[:16]


def kitti_weights(cfg, seed=11):
    """The reference's REAL trained KITTI tensors (results_kitti/Log_11011605/kernel_points/epoch61: 34 weight arrays + 10
    kernel-point sets, committed as tests/golden/kitti_*.npz) over seeded values for what its dump lacks (batch-norm statistics,
    the deepest block) -- the same weight set tests/test_gpu_real_weights.py uses."""
    from d3feat_amd.models.variables import build_variables
    W = build_variables(cfg, seed=seed, randomize_bn=True).values
    n = 0
    for fn in ("kitti_epoch61_weights.npz", "kitti_kernel_points.npz"):
        src = np.load(os.path.join(ROOT, "tests", "golden", fn))
        for k in src.files:
            name = k.replace("__", "/")
            if name not in W or W[name].shape != src[k].shape:
                raise ValueError("trained tensor %s %s does not fit the KITTI architecture" % (name, src[k].shape))
            W[name] = np.ascontiguousarray(src[k], np.float32)
            n += 1
    return W, n


class Workload:
    """What a "step" is made of, per BASELINE configuration.  items: what one step feeds the engine -- a raw cloud (config #2 /
    #3), a pair of raw clouds (config #4), an already subsampled cloud (config #1)."""

    def __init__(self, args, rank):
        from d3feat_amd.models.variables import build_variables
        from d3feat_amd.utils.config import kitti_config, threedmatch_config
        from d3feat_amd.utils.synthetic import lidar_sweep, room_fragment
        self.two, self.stage0, self.frames = False, True, 1
        self.level_ratio, self.cap_factor = args.level_ratio, args.cap_factor
        self.weights = "random-init weights (reference initialiser, seed 42), 14.1M params"
        nfr = max(args.pool, args.cpu_fragments)
        seeds = [rank * 1000 + i for i in range(nfr)]
        if args.config4:
            self.name = "config4"
            self.cfg = kitti_config()
            self.W, nreal = kitti_weights(self.cfg)
            self.weights = "the reference's trained KITTI tensors (epoch61: %d arrays) + seeded batch-norm statistics / deepest block" % nreal
            self.two, self.frames = True, 2
            self.level_ratio = max(args.level_ratio, 0.6)       # LiDAR sweeps thin out more slowly than room surfaces (test_gpu_configs)
            self.items = [(lidar_sweep(s, args.kitti_points), lidar_sweep(s + 100, args.kitti_points)) for s in seeds]
            self.what = ("SURVEY §8d config #4 (NOT the config #2 headline): KITTI-like stacks of two DIFFERENT synthetic LiDAR "
                         "sweeps (64 rings, %dk raw pts each) -> grid subsample 0.3 m" % (args.kitti_points // 1000))
        elif args.demo:
            self.name = "demo"
            self.cfg = threedmatch_config()
            self.W = build_variables(self.cfg, seed=42).values
            self.stage0 = False
            g = os.path.join(ROOT, "tests", "golden")
            self.items = [np.load(os.path.join(g, "demo_bin0_sub003.npy")), np.load(os.path.join(g, "demo_bin1_sub003.npy"))]
            self.what = ("SURVEY §8d config #1 (NOT the config #2 headline): the reference's demo pair after its own 0.03 m grid "
                         "subsample (14 007 / 13 530 pts), each cloud")
        else:
            self.name = "config3" if args.config3 else "config2"
            self.cfg = threedmatch_config()
            self.W = build_variables(self.cfg, seed=42).values
            if args.config3:
                # sizes U(15 k, 45 k) after the subsample: points scale with the surface, so the room edge goes with sqrt(size) and
                # the raw count with the size (config #2: 300 k raw points, edge 1.68 m -> ~30 k points)
                targets = np.random.default_rng(1234 + rank).uniform(15000, 45000, nfr)
                self.items = [room_fragment(s, n_raw=int(args.raw_points * t / 30000.0), edge=args.edge * float(np.sqrt(t / 30000.0)))
                              for s, t in zip(seeds, targets)]
                self.what = ("SURVEY §8d config #3 shape (NOT the config #2 headline): synthetic 3DMatch room fragments of U(15 k, 45 k) "
                             "points after the 0.03 m subsample, capacities of the largest")
            else:
                self.items = [room_fragment(s, n_raw=args.raw_points, edge=args.edge) for s in seeds]
                self.what = "SURVEY §8d config #2: synthetic 3DMatch room fragment, 300k raw pts -> grid subsample 0.03 m"

    def clouds(self, item):
        return list(item) if self.two else [item]

    def to_device(self, item, device):
        import torch
        return tuple(torch.from_numpy(c).to(device) for c in item) if self.two else torch.from_numpy(item).to(device)

    def raw_points(self, item):
        return sum(int(c.shape[0]) for c in self.clouds(item))

    def kept(self, rec):
        """What a step contributes to the shard: the first cloud's records of a stacked self-pair (utils/tester.py:208-229 keeps
        in_batches[0]); both clouds of a stack of two different frames."""
        return rec if self.two else rec[: rec.shape[0] // 2]


def main():
    args = parse()
    if args.bf16_features:
        args.bf16 = True
    from d3feat_amd import launch
    if launch.needs_launch(args.gpus):
        # `python bench.py --gpus N` without a launcher around it: start the N ranks (or refuse: fewer than N devices)
        sys.exit(launch.relaunch(args.gpus, os.path.abspath(__file__), sys.argv[1:]))
    if args.config4 or args.demo:
        args.pool = max(args.pool if args.config4 else 2, 2)
        if args.demo:
            args.pool, args.cpu_fragments = 2, min(args.cpu_fragments, 2)
    schedule = [int(x) for x in args.schedule.split(",")] if args.schedule else None
    if schedule:
        assert sum(schedule) == args.steps and min(schedule) >= 1, "--schedule must sum to --steps"
        args.batch = max(schedule)
        if args.slots <= 0:
            args.slots = min(len(schedule), 4)
    if args.slots <= 0:
        # a job of one round of replays (the driver's 20 fragments): three replays of 7 beat four of 5 since the contractions moved
        # to the operand-split form (1469 against 1437 fragments/s, three alternating pairs in one visit: r04_experiments.txt x9)
        args.slots = 4 if (args.steps >= 64 or args.batch > 0) else 3
    if args.batch <= 0:
        args.batch = (12 if args.steps >= 384 else 8) if args.steps >= 64 else max(1, min(8, -(-args.steps // max(args.slots, 1))))
        if args.config4:
            args.batch = min(args.batch, 4)          # a stack of two 120k-pt sweeps is four 30k-pt clouds' worth of rows
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not launch.check_world(args.gpus, world):
        sys.exit(2)
    if not torch.cuda.is_available() or torch.cuda.device_count() <= local_rank:
        sys.stderr.write("bench.py: rank %d needs GPU %d, this box exposes %d: refusing to run\n"
                         % (rank, local_rank, torch.cuda.device_count() if torch.cuda.is_available() else 0))
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1 or "RANK" in os.environ:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend="nccl", device_id=device)
        world = dist.get_world_size()            # n_gpus in the line = what RCCL saw

    from d3feat_amd import ops, parallel
    from d3feat_amd.datasets.common import FragmentDataset
    from d3feat_amd.models.KPFCNN_model import KernelPointFCNN
    from d3feat_amd import tf_custom_ops as tfo

    if args.ablate:
        args.no_cpu_baseline = args.no_instrument = args.no_mirror_extra = args.no_pcie_extra = args.no_latency = True
    do_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline
    if not do_cpu:
        args.cpu_fragments = 0
    wl = Workload(args, rank if args.seed_rank < 0 else args.seed_rank)
    cfg, W = wl.cfg, wl.W
    secondary_ok = wl.name in ("config2", "config3")           # mirror / PCIe / marginal-cost extras: headline workload only
    # this rank's synthetic items, resident in HBM before timing starts; the first `pool` are cycled through the timed region,
    # the first `cpu_fragments` form the CPU / parity sample
    raws_host = wl.items
    raws_all = [wl.to_device(r, device) for r in raws_host]
    raws = raws_all[: args.pool]

    # neighbourhood limits: calibrated like init_test_input_pipeline on this rank's pool (every cloud as a self-pair: the
    # searches are per cloud, so the histogram proportions are those of the real stacks), histograms summed over ranks
    subs = [(tfo.grid_subsampling(c, cfg.first_subsampling_dl) if wl.stage0 else c).cpu().numpy()
            for r in raws for c in (r if wl.two else (r,))]
    cal = FragmentDataset(subs)
    hist_n = int(np.ceil(4 / 3 * np.pi * (cfg.density_parameter + 1) ** 3))
    cal.neighborhood_limits = np.full(cfg.num_layers, hist_n, np.int32)
    hists = cal.calibrate_neighbors(cfg, samples_threshold=10 ** 9)
    hists = parallel.allreduce_histograms(hists, device)
    cumsum = np.cumsum(hists.T, axis=0)
    limits = np.sum(cumsum < (0.8 * cumsum[hist_n - 1, :]), axis=0).astype(np.int32)
    if args.limits:
        limits = np.array([int(x) for x in args.limits.split(",")], np.int32)
        assert limits.shape == (cfg.num_layers,), "--limits wants %d numbers" % cfg.num_layers

    model = KernelPointFCNN(None, cfg, weights=W, device=device)
    step = Step(cfg, model, limits, device, bf16=args.bf16, bf16_features=args.bf16_features, two=wl.two, stage0=wl.stage0)
    if args.ablate:
        install_ablation(args.ablate.split(","))     # after the calibration (which reads its results back)

    def sync():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(device)

    engine = None
    per = 2 if wl.two else 1
    # voxels per ITEM (both clouds of a two-frame stack together: the engine's capacities bound their sum)
    n0_items = [sum(len(x) for x in subs[i * per:(i + 1) * per]) for i in range(len(raws))]
    n0_max = max(n0_items)
    raw_max = max(wl.raw_points(r) for r in raws_host)
    if world > 1:
        # the overlapped exchange addresses fragment k of EVERY rank at rows [k * frag_rows, ...): the stride (and with it the
        # engine's capacities) must be one number on all ranks, not a function of a rank's own pool
        agree = torch.tensor([n0_max, raw_max], dtype=torch.int64, device=device)
        dist.all_reduce(agree, op=dist.ReduceOp.MAX)
        n0_max, raw_max = int(agree[0].item()), int(agree[1].item())

    def make_engine(batch, slots, streams=None, mirror=False, bf16=args.bf16, bf16_features=args.bf16_features):
        from d3feat_amd.engine import FragmentEngine
        raw_cap = int(raw_max * 1.05) + 1024
        n0_cap = (int(n0_max * wl.cap_factor) + 1023) // 1024 * 1024
        return FragmentEngine(cfg, W, limits, raw_cap=raw_cap, n0_cap=n0_cap, level_ratio=wl.level_ratio, slots=slots, device=device,
                              n0_hint=int(np.mean(n0_items)), mirror_self_pair=mirror, batch=batch, bf16=bf16,
                              bf16_features=bf16_features, streams=streams, two_clouds=wl.two, stage0=wl.stage0)
    if not args.eager:
        # the fragment engine: whole stack = one replayed HIP graph with device-resident sizes, `slots` in flight
        engine = make_engine(args.batch, args.slots, mirror=args.mirror)
    # this rank's shard: every step's [xyz | desc | score] records stay in HBM until the final gather
    # (N > 1: overlapped mode -- fixed stride per step, chunks of --gather-chunk steps exchanged asynchronously while the next
    # replays run; every rank holds the same number of steps, so the collective order is the same everywhere)
    dst = None if args.gather_to == "all" else int(args.gather_to)
    frag_rows = (int(n0_max * wl.cap_factor) + 1023) // 1024 * 1024
    if args.gather_chunk > 0:
        # fixed stride per step: the replays write their records straight into the shard (FragmentEngine.submit(out=...)), and
        # with N > 1 finished chunks are exchanged while the next replays run
        if world > 1:
            strides = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
            dist.all_gather(strides, torch.tensor([frag_rows], dtype=torch.int64, device=device))
            assert all(int(x.item()) == frag_rows for x in strides), "fragment stride differs between ranks: %s" % [int(x.item()) for x in strides]
        keep_steps = (max(args.steps, 128) if world == 1 else args.steps) + args.gather_chunk
        shard = parallel.ShardCollector(rows_cap=keep_steps * frag_rows * (2 if wl.two else 1), width=36, device=device,
                                        chunk_frags=args.gather_chunk, frag_rows=frag_rows * (2 if wl.two else 1), dst=dst)
    else:
        shard = parallel.ShardCollector(rows_cap=((max(args.steps, 128) if world == 1 else args.steps) + 8) * int(n0_max * 1.02 + 64),
                                        width=36, device=device, dst=dst)

    def run(nsteps, engine=engine, collect=None, pool=raws):
        """nsteps steps through the hot path; every step's record block goes to `collect` (ShardCollector)."""
        if engine is None:
            for i in range(nsteps):
                rec = step(pool[i % len(pool)])
                if collect is not None:
                    collect.add(wl.kept(rec))
            return
        S, F = len(engine.slots), engine.F
        busy = [False] * S

        inplace = [False] * S

        def drain(sl):
            for rec in engine.fetch(sl, packed=True):
                if collect is not None:
                    collect.add(rec if inplace[sl] else wl.kept(rec))      # (submit(out=...): the kept clouds only, already)
            busy[sl] = False
        i = k = 0
        sizes = schedule if (schedule and nsteps == args.steps and engine.F >= max(schedule)) else None
        while i < nsteps:                        # replays of up to F steps each, round-robin over the slots
            sl = k % S
            if busy[sl]:
                drain(sl)
            nb = min(F, nsteps - i) if sizes is None else sizes[k]
            dsts = collect.slots(nb) if (collect is not None and not engine.mirror) else None
            inplace[sl] = dsts is not None
            engine.submit(sl, [pool[(i + j) % len(pool)] for j in range(nb)], out=dsts)
            busy[sl] = True
            i += nb
            k += 1
        for kk in range(k, k + S):               # drain in submission order
            if busy[kk % S]:
                drain(kk % S)

    # at least W untimed steps; with the engine, enough of them to replay every slot's graph once
    run(max(args.warmup, (args.slots * args.batch) if engine is not None else 0), collect=shard)
    if world > 1:
        shard.gather(compact=False)
    shard.reset()
    sync()
    # ---- the timed region: R windows of EXACTLY K steps each, every window bracketed by barrier + synchronize, MAX over ranks per
    # window; `value` is the median window.  (One window of the driver's 20-fragment job lasts 14 ms -- no steady state; single
    # windows spread 1290..1430 fragments/s across boxes, which hid a whole round of kernel work: r03 x30/x34.)
    # the interpreter's cyclic GC runs between the windows, not in them (one GC pass is a quarter of a 14 ms window, r03 x29)
    import gc
    windows = []
    R = max(1, args.windows)
    for w in range(R):
        gc.collect()
        gc.disable()
        sync()
        if w == R // 2:
            ops.trace_marker(1, device)          # (named kernel: tools/rocpd_summary.py --timed-region cuts a trace here ...
            sync()
        t0 = time.perf_counter()
        run(args.steps, collect=shard)
        gathered = shard.gather(compact=False) if shard.chunk_frags > 0 else shard.gather()
        sync()
        dt = time.perf_counter() - t0
        gc.enable()
        if w == R // 2:
            ops.trace_marker(2, device)          #  ... and here: the middle window; outside the clock)
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        if world > 1:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        windows.append(float(tmax.item()))
        npts = int(np.mean(shard.frag_rows)) if shard.frag_rows else 0
        gathered_rows = [int(sum(g[1])) for g in gathered]
        gathered_frags = [len(g[1]) for g in gathered]
        received = [g[0] is not None for g in gathered]
        if w == R - 1:
            # outside the clock: a float64 checksum of every received shard's VALID rows (rank 0's view of the job; a single process
            # started with --seed-rank r --limits ... reproduces rank r's number: tests/test_gpu_multi.py)
            checksums = []
            for g in gathered:
                if g[0] is None:
                    checksums.append(None)
                    continue
                parts = g[0] if isinstance(g[0], (list, tuple)) else [g[0]]       # overlapped mode: one view per fragment
                checksums.append(float(sum(float(p.double().sum().item()) for p in parts)))
        del gathered
        shard.reset()
    dt = float(np.median(windows))

    # the secondary measurements below are steady-state figures: never shorter than 128 steps, whatever --steps the
    # headline was asked for
    sec_steps = max(args.steps, 128)
    # ---- single-step latency: F = 1, one slot, graph path -- the reference's own timing metric is the time of ONE sess.run
    # ("Avergae Feature Extraction Time", utils/tester.py:195-200,233), not a throughput
    latency = None
    if rank == 0 and engine is not None and not args.no_latency:
        try:
            eng1 = make_engine(1, 1, streams=[engine.slots[0].stream])
            for i in range(4):
                eng1.submit(0, raws[i % len(raws)])
                eng1.fetch(0, packed=True)
            lat = []
            for i in range(48):
                torch.cuda.synchronize(device)
                t1 = time.perf_counter()
                eng1.submit(0, raws[i % len(raws)])
                eng1.fetch(0, packed=True)               # waits for the replay's completion event + the status read-back
                lat.append((time.perf_counter() - t1) * 1e3)
            latency = {"median": round(float(np.median(lat)), 4), "p10": round(float(np.percentile(lat, 10)), 4),
                       "p90": round(float(np.percentile(lat, 90)), 4), "n": len(lat), "unit": "ms per step, submit -> results in HBM",
                       "execution": "HIP-graph replay of ONE stack (F = 1), nothing else in flight",
                       "engine_fallbacks": eng1.fallbacks,
                       "what": "the reference's own metric: wall time of one sess.run (utils/tester.py:195-200,233)"}
            del eng1
        except Exception as exc:      # a secondary number must never cost the headline line
            latency = {"error": repr(exc)[:200]}

    # ---- secondary number (N = 1): PCIe-inclusive -- raw fragments start in pinned HOST memory, results end there -------
    pcie = None
    if rank == 0 and world == 1 and engine is not None and not args.no_pcie_extra and secondary_ok:
        try:
            hraws = [r.cpu().pin_memory() for r in raws]
            S, F = len(engine.slots), engine.F
            cap_rows = 2 * engine.n0_cap
            hout = [[torch.empty((cap_rows, 36), dtype=torch.float32).pin_memory() for _ in range(F)] for _ in range(S)]

            def drain_host(sl):
                for j, rec in enumerate(engine.fetch(sl, packed=True)):   # device views -> pinned host, asynchronously
                    hout[sl][j][: rec.shape[0]].copy_(rec, non_blocking=True)

            def run_host(nsteps):
                busy = [False] * S
                i = k = 0
                while i < nsteps:
                    sl = k % S
                    if busy[sl]:
                        drain_host(sl)
                    nb = min(F, nsteps - i)
                    engine.submit(sl, [hraws[(i + j) % len(hraws)] for j in range(nb)])
                    busy[sl] = True
                    i += nb
                    k += 1
                for kk in range(k, k + S):
                    if busy[kk % S]:
                        drain_host(kk % S)
                        busy[kk % S] = False

            run_host(S * F)
            torch.cuda.synchronize(device)
            t1 = time.perf_counter()
            run_host(sec_steps)
            torch.cuda.synchronize(device)
            dt3 = time.perf_counter() - t1
            pcie = {"value": round(sec_steps / dt3, 3), "unit": "fragments/s", "ms_per_step": round(dt3 / sec_steps * 1e3, 4),
                    "steps": sec_steps,
                    "h2d_bytes_per_fragment": int(np.mean([r.shape[0] for r in raws]) * 12),
                    "d2h_bytes_per_fragment": int(2 * npts * 36 * 4),
                    "note": "NOT the headline: raw clouds read from pinned host memory, the record blocks copied back "
                            "to pinned host memory, copies on the slot streams overlapped with the other replays"}
        except Exception as exc:  # the secondary number must never cost the headline line
            pcie = {"error": repr(exc)[:200]}

    # ---- secondary number (N = 1): the same fragments with the self-pair computed once and mirrored ---------------------
    mirror_extra = None
    if rank == 0 and world == 1 and engine is not None and not args.mirror and not args.no_mirror_extra and not args.bf16 and secondary_ok:
        eng2 = make_engine(args.batch, args.slots, streams=[sl.stream for sl in engine.slots], mirror=True, bf16=False, bf16_features=False)
        run(args.warmup, eng2)
        torch.cuda.synchronize(device)
        t1 = time.perf_counter()
        run(sec_steps, eng2)
        torch.cuda.synchronize(device)
        dt2 = time.perf_counter() - t1
        mirror_extra = {"value": round(sec_steps / dt2, 3), "unit": "fragments/s", "ms_per_step": round(dt2 / sec_steps * 1e3, 4),
                        "steps": sec_steps, "engine_fallbacks": eng2.fallbacks,
                        "note": "NOT the headline: the two halves of the reference's stacked self-pair are identical by construction "
                                "(per-cloud searches and head normalisation), so this mode computes one copy and mirrors it into "
                                "the stacked outputs (FragmentEngine(mirror_self_pair=True)); same results to fp32 summation order"}
        del eng2

    # ---- instrumented pass (untimed): per-launch HIP events on the launch stream --------------------------------
    layers = roof = roofs = fam_flops = None
    if rank == 0 and not args.no_instrument:
        Fp = engine.F if engine is not None else 1
        roof, roofs, layers, fam_flops = instrumented_pass(cfg, step, raws, Fp, max(2, min(args.steps, 8) // Fp), device)

    # ---- marginal cost of the two matrix-pipe families inside the timed regime (N = 1) --------------------------------------
    # A launch timed alone (the instrumented pass) cannot fill the chip with these small shapes; with four replays in flight
    # the question is what a family costs the THROUGHPUT.  Measured by leaving its library calls out of a second engine
    # (the replay keeps its shape: sizes are device-resident and do not depend on feature values; outputs are garbage).
    marginal = None
    if rank == 0 and world == 1 and engine is not None and fam_flops and not args.no_marginal and not args.ablate and not args.mirror \
            and not args.bf16 and secondary_ok:
        marginal = marginal_costs(args, cfg, W, limits, engine, run, shard, device, sec_steps, fam_flops)

    # ---- CPU baseline + parity at the benchmarked configuration (rank 0, N=1) ---------------------------------------------
    cpu = parity = None
    if do_cpu:
        cpu, refs = cpu_baseline(wl, limits, raws_host[: max(1, args.cpu_fragments)], one_thread=not args.no_cpu_1thread)
        parity = parity_check(cfg, engine, step, raws_all[: len(refs)], refs, device, BF16_TOL if args.bf16 else PARITY_TOL)

    if rank == 0:
        frames = wl.frames
        unit_note = ("a step is one stack of two different frames = 2 fragments" if wl.two else "a step is one fragment (computed as the "
                     "reference's stacked self-pair)")
        res = {
            "metric": "fragments/sec (30k-pt clouds)" + ((" -- configs[4]: bf16 features + bf16 contraction" if args.bf16_features
                                                         else " -- configs[4]: bf16 contraction") if args.bf16 else "") +
                      {"config2": "", "config3": " -- configs[2] shape: fragment sizes U(15k, 45k)",
                       "config4": " -- configs[3]: KITTI-like frames (~%dk pts at 0.3 m), two different frames per stack" % round(npts / 2000),
                       "demo": " -- configs[0]: the reference's demo pair"}[wl.name],
            "value": round(world * args.steps * frames / dt, 3), "unit": "fragments/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": (("bf16 feature storage between the layers, bf16 operands / f32 accumulate in the unary + unfused KPConv "
                       "contractions, f32 arithmetic inside the gather kernels (NOT the fp32 parity path)") if args.bf16_features else
                      "bf16 operands / f32 accumulate in the unary + unfused KPConv contractions; f32 elsewhere (NOT the fp32 "
                      "parity path)" if args.bf16 else "f32"), "data": "synthetic" if wl.name != "demo" else "the reference's demo clouds",
            "timing": {"windows": R, "window_ms": [round(x * 1e3, 3) for x in windows],
                       "value_per_window": [round(world * args.steps * frames / x, 1) for x in windows],
                       "p10": round(world * args.steps * frames / float(np.percentile(windows, 90)), 1),
                       "p90": round(world * args.steps * frames / float(np.percentile(windows, 10)), 1),
                       "what": "R back-to-back windows of exactly `steps` steps, each bracketed by barrier + synchronize, max over "
                               "ranks per window; value / ms_per_step = the MEDIAN window"},
            "latency_ms": latency,
            "config": {"workload": wl.what + " (~%dk pts per %s)" % (round(npts / 1000), "stack" if wl.two else "cloud") +
                                   (" -> self-pair" if not wl.two else "") + " -> 5-level pyramid -> full KPFCNN forward (" + wl.weights +
                                   ") -> 32-d descriptors + scores; " + unit_note,
                       "points_per_cloud_pool": sorted(int(len(x)) for x in subs),
                       "points_per_cloud": npts, "neighborhood_limits": [int(x) for x in limits],
                       "fragments_per_gpu": args.steps * frames, "parallelism": "fragment-dp%d" % world,
                       "contraction": ("bf16 operands (configs[4])" if args.bf16 else
                                       contraction_description()),
                       "rccl": ({"backend": dist.get_backend(), "world_size": dist.get_world_size()} if dist.is_initialized() else None),
                       "final_gather": {"ranks": len(gathered_rows), "to": "rank %d" % dst if dst is not None else "every rank",
                                        "received_on_rank0": received,
                                        "fragments_per_rank": gathered_frags,
                                        "rows_per_rank": gathered_rows, "bytes_per_rank": [r * 144 for r in gathered_rows],
                                        "checksum_per_rank": checksums,
                                        "what": "every rank's whole shard of [xyz | desc | score] records (144 B/point, the first cloud "
                                                "of every stacked self-pair: what utils/tester.py:208-229 keeps per fragment), inside "
                                                "the timed region: " + ("asynchronous exchanges of %d-step chunks behind the compute"
                                                                        % shard.chunk_frags if shard.chunk_frags > 0 else
                                                                        "one padded exchange at the end")},
                       "execution": ("eager op-by-op launches" if engine is None else
                                     "HIP-graph replay of %d stacked step(s), device-resident sizes, %d replays in flight%s"
                                     % (engine.F, len(engine.slots),
                                        "; self-pair computed once and mirrored" if args.mirror else "")),
                       "fragments_per_replay": (engine.F if engine is not None else 1), "schedule": schedule,
                       "capacities": (None if engine is None else
                                      {"raw_points_per_fragment": engine.raw_cap, "voxels_per_cloud": engine.n0_cap,
                                       "rows_per_level": [int(c) for c in engine.caps]}),
                       "engine_fallbacks": (engine.fallbacks if engine is not None else None),
                       "engine_isolated_replays": (engine.isolated if engine is not None else None)},
            "parity": parity, "roofline": roof, "rooflines": roofs, "marginal_cost": marginal, "kpconv_layers_ms": layers,
            "cpu_baseline": cpu,
            "mirror_self_pair": mirror_extra, "pcie_inclusive": pcie,
        }
        if cpu:
            res["vs_cpu_baseline"] = round(res["value"] / cpu["value"], 2)
        if args.ablate:
            res = {"INVALID": "ablation run (--ablate %s): op families skipped, outputs are garbage" % args.ablate,
                   "value": res["value"], "ms_per_step": res["ms_per_step"], "ablate": args.ablate}
        # the lab notebook (every family's roofline, per-launch tables, marginal costs, secondary measurements) goes to a FILE;
        # the driver-facing stdout line is the compact object only (r04: a 20 KB line was cut by the driver's 8 KB tail and
        # BENCH_r04.parsed came out null)
        detail_path = write_detail(res, args.detail_out)
        if args.detail_stdout:
            print(json.dumps(res))
        print(compact_line(res, detail_path))
        sys.stdout.flush()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if parity is not None and not parity["ok"]:
        print("PARITY FAILURE at the benchmarked configuration: %s" % json.dumps(parity), file=sys.stderr)
        sys.exit(3)


# ---------------------------------------------------------------------------------------------------------------------
COMPACT_MAX_BYTES = 3072     # the driver keeps the last 8 KB of stdout; the line it must parse stays well inside


def _pick(obj, keys):
    return {k: obj[k] for k in keys if isinstance(obj, dict) and k in obj}


def _short(v, n=200):
    return v if not isinstance(v, str) or len(v) <= n else v[: n - 3] + "..."


def _finite(o):
    """Strict JSON: NaN / Infinity (json.dumps would print them bare) become null."""
    if isinstance(o, float):
        return o if o == o and o not in (float("inf"), float("-inf")) else None
    if isinstance(o, dict):
        return {k: _finite(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_finite(v) for v in o]
    return o


def compact_line(res, detail_path=None):
    """The ONE stdout line of a run: the contract's keys + parity / roofline / cpu_baseline / latency, every string bounded,
    per-rank lists summarised, <= COMPACT_MAX_BYTES.  Everything else lives in the detail file."""
    if "INVALID" in res:
        return json.dumps(_finite(res), allow_nan=False)
    out = _pick(res, ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                      "data"])
    out["metric"] = _short(out.get("metric"), 120)
    dt = res.get("dtype") or ""
    out["dtype"] = "f32" if dt == "f32" else ("bf16" if dt.startswith("bf16") else _short(dt, 40))
    cfg = res.get("config") or {}
    c = {"workload": _short(cfg.get("workload"), 240)}
    c.update(_pick(cfg, ["points_per_cloud", "neighborhood_limits", "fragments_per_gpu", "parallelism", "fragments_per_replay", "schedule",
                         "engine_fallbacks", "engine_isolated_replays", "rccl"]))
    c["contraction"] = _short(cfg.get("contraction"), 110)
    c["execution"] = _short(cfg.get("execution"), 120)
    fg = cfg.get("final_gather")
    if fg and cfg.get("rccl"):
        rec = fg.get("received_on_rank0") or []
        fr, rows = fg.get("fragments_per_rank") or [], fg.get("rows_per_rank") or []
        c["final_gather"] = {"ranks": fg.get("ranks"), "to": fg.get("to"), "received_on_rank0": rec,
                             "fragments_per_rank": fr if len(fr) <= 16 else [min(fr), max(fr)],
                             "rows_per_rank": rows if len(rows) <= 16 else [min(rows), max(rows)],
                             "rows_total": int(sum(rows)), "bytes_total": int(sum(rows)) * 144}
        if fg.get("checksum_per_rank") and len(rows) <= 16:
            c["final_gather"]["checksum_per_rank"] = fg["checksum_per_rank"]
    out["config"] = c
    tm = res.get("timing") or {}
    out["timing"] = _pick(tm, ["windows", "window_ms", "p10", "p90"])
    if len(out["timing"].get("window_ms", [])) > 16:
        out["timing"]["window_ms"] = out["timing"]["window_ms"][:16]
    if res.get("parity") is not None:
        out["parity"] = _pick(res["parity"], ["ok", "fragments", "points_equal", "idx_equal", "desc_max_abs", "score_max_abs", "tolerance",
                                              "engine_fallbacks", "idx_rows_differing_only_inside_bit_equal_distance_ties"])
    else:
        out["parity"] = None
    rf = res.get("roofline")
    if rf:
        r = _pick(rf, ["kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "alg_flops_per_launch", "alg_bytes_per_launch",
                       "avg_launch_us", "launches_per_step", "fragments_per_launch", "traffic_source", "traffic_stale",
                       "mfma_busy", "mfma_busy_source", "mfma_busy_stale"])
        mp = rf.get("matrix_pipe") or {}
        if mp:
            r["peak_is"] = "dense bf16 MFMA peak / 6 (six bf16 products per f32 product)"
            r.update(_pick(mp, ["achieved_over_fp32_mfma_peak", "issued_tflops"]))
        out["roofline"] = r
    else:
        out["roofline"] = None
    fams = res.get("rooflines")
    if fams:   # one number per family: fraction of the roof that binds it (full objects: detail file)
        out["roofline_frac_by_kernel"] = {_short(f.get("kernel"), 40): f.get("frac") for f in fams[:16]}
    cb = res.get("cpu_baseline")
    if cb:
        b = _pick(cb, ["value", "unit", "cores", "host_cores", "kind", "geometry_kind", "value_1thread", "value_multithread", "pinned",
                       "reference_python"])
        if cb.get("fragment_s"):
            b["fragment_s_p10_p90"] = [cb["fragment_s"].get("p10"), cb["fragment_s"].get("p90")]
        b["sample"] = _short(cb.get("sample"), 180)
        out["cpu_baseline"] = b
    else:
        out["cpu_baseline"] = None
    lat = res.get("latency_ms")
    out["latency_ms"] = _pick(lat, ["median", "p10", "p90", "n", "engine_fallbacks", "error"]) if lat else None
    if "vs_cpu_baseline" in res:
        out["vs_cpu_baseline"] = res["vs_cpu_baseline"]
    if res.get("kpconv_layers_ms"):
        out["ms_per_kpconv_layer"] = [round(l["total_ms_per_fragment"], 4) for l in res["kpconv_layers_ms"]
                                      if l.get("total_ms_per_fragment") is not None][:16]
    out["detail"] = detail_path
    line = json.dumps(_finite(out), allow_nan=False, separators=(",", ":"))
    for k in ("roofline_frac_by_kernel", "ms_per_kpconv_layer", "timing"):     # never reached by today's objects; a guard, not a plan
        if len(line) <= COMPACT_MAX_BYTES:
            break
        out.pop(k, None)
        line = json.dumps(_finite(out), allow_nan=False, separators=(",", ":"))
    return line


def write_detail(res, path):
    """The full object of the run as a file (tracked copies go to profiles/ by hand); returns the path relative to the repo, or
    None when the directory is not writable -- a notebook must never cost the headline line."""
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(_finite(res), f)
            f.write("\n")
        return os.path.relpath(path, ROOT)
    except OSError:
        return None


# ---------------------------------------------------------------------------------------------------------------------
def marginal_costs(args, cfg, W, limits, engine, run, shard, device, nsteps, flops):
    import torch
    from d3feat_amd import _lib
    from d3feat_amd.engine import FragmentEngine

    def timed(eng):
        # the faster of three windows: one disturbed window (r06_v28: 0.43 instead of 0.34 ms for the ablated engine) would otherwise
        # halve a family's marginal cost
        run(args.warmup, eng)
        best = None
        for _ in range(3):
            torch.cuda.synchronize(device)
            t = time.perf_counter()
            run(nsteps, eng, collect=shard)                  # same bookkeeping as the headline region
            torch.cuda.synchronize(device)
            ms = (time.perf_counter() - t) / nsteps * 1e3
            shard.reset()
            best = ms if best is None else min(best, ms)
        return best
    base_ms = timed(engine)
    out = {"method": "throughput of a second engine whose replays leave one op family's library calls out, same fragments, "
                     "same F x slots, the fastest of three windows each; ms_per_fragment = base - ablated", "steps": nsteps, "base_ms_per_fragment": round(base_ms, 4)}
    real_load = _lib.load
    try:
        for fam in ("gemm", "kpconv"):
            install_ablation([fam] + (["rowpos"] if fam == "kpconv" else []))
            eng = FragmentEngine(cfg, W, limits, raw_cap=engine.raw_cap, n0_cap=engine.n0_cap, level_ratio=engine.level_ratio,
                                 slots=len(engine.slots), device=device,
                                 n0_hint=engine.n0_hint, streams=[sl.stream for sl in engine.slots], batch=engine.F)
            ms = timed(eng)
            _lib.load = real_load
            d = base_ms - ms
            e = {"ablated_ms_per_fragment": round(ms, 4), "ms_per_fragment": round(d, 4)}
            if d > 0:
                tf = flops[fam] / (d * 1e-3) / 1e12
                from d3feat_amd import ops as _ops
                x3 = fam == "gemm" and _ops.GEMM_X3
                pk = MFMA_X3_PEAK_TF if x3 else MFMA_F32_PEAK_TF
                e.update(alg_flops_per_fragment=int(flops[fam]), tflops=round(tf, 2), peak=round(pk, 1), frac=round(tf / pk, 4),
                         bound=("mfma (six bf16 products per fp32 product: dense bf16 peak / 6; the tall layers of the fine levels are "
                                "HBM-bound: rooflines[gemm_x3r_kernel])" if x3 else "mfma") if fam == "gemm"
                         else "valu (aggregation: 157.3 TF/s of packed fp32 FMAs) + contraction (since round 5 on the bf16 matrix "
                              "cores in the operand-split form); priced against 157.3")
                if x3:
                    e["over_fp32_mfma_peak"] = round(tf / MFMA_F32_PEAK_TF, 4)
            out[fam] = e
            del eng
    finally:
        _lib.load = real_load
    return out


def install_ablation(families):
    """Replace the torch-level front ends of the named op families by allocations without a library call (bench.py --ablate)."""
    import torch
    from d3feat_amd import _lib, ops
    lib = _lib.load()

    class _Skip:
        def __init__(self, names):
            self.names = set(names)

        def __getattr__(self, name):
            if name in self.names:
                return lambda *a, **k: 0
            return getattr(lib, name)
    skip = set()
    for f in families:
        skip |= {"gemm": {"d3f_gemm_f32", "d3f_gemm_upsample_cat_f32", "d3f_gemm_f32t", "d3f_gemm_x3"},
                 "kpconv": {"d3f_kpconv_aggregate", "d3f_kpconv_fused_c1", "d3f_kpconv_fused32", "d3f_kpconv_fused32_x3", "d3f_kpconv_fused", "d3f_kpconv_fused_x3"},
                 "kpconv_deep": {"d3f_kpconv_fused", "d3f_kpconv_fused_x3", "d3f_kpconv_aggregate"},
                 "kpconv32": {"d3f_kpconv_fused32", "d3f_kpconv_fused32_x3"}, "kpconv_c1": {"d3f_kpconv_fused_c1"},
                 "rowpos": {"d3f_row_positive"}, "maxpool": {"d3f_ind_max_pool"}, "head": {"d3f_detect_head"},
                 "pack": {"d3f_pack_descriptors"},
                 # geometry: only meaningful together with the whole network ablated (nothing consumes the index matrices then)
                 "nb_search": {"d3f_neighbor_grid_search"}, "nb_build": {"d3f_neighbor_grid_build"}}[f.strip()]
    proxy = _Skip(skip)
    _lib.load = lambda: proxy


# op records whose "launch" is a group of many small dependent kernels: op by op their HIP events mostly measure launch
# gaps, so such a family is reported in `rooflines` but is never the headline `roofline` object (tagged by the RECORD NAME
# the library front end emits -- not by anything in the display label)
MULTI_LAUNCH_OPS = frozenset({"nb_grid_build", "grid_subsample"})


def classify_record(cfg, name, info):
    """One timed op record -> (family key, algorithmic bytes, algorithmic flops) per SURVEY.md §8(d), or None."""
    flops = 0.0
    if name == "kpconv_aggregate":
        key = "kpconv_agg_vec4<Cin=%d>" % info["Cin"] if info["Cin"] % 4 == 0 else "kpconv_agg_scalar<Cin=%d>" % info["Cin"]
        # every KPConv of the shipped architecture has Cout == Cin except the first (1 -> 64); the contraction of this
        # layer is a separate gemm_f32 record, so only the aggregation flops count here
        cout = info["Cin"] if info["Cin"] > 1 else cfg.first_features_dim
        nbytes = kpconv_alg_bytes(info["Nq"], info["Ns"], info["K"], info["Cin"], cout)
        flops = kpconv_flops(info["Nq"], info["K"], info["Cin"], cout)[1]
    elif name in ("kpconv_fused_c1", "kpconv_fused32", "kpconv_fused"):
        cin, cout = info["Cin"], info["Cout"]
        key = {"kpconv_fused_c1": "kpconv_c1_kp_kernel", "kpconv_fused32": "kpconv_fused32_kernel"}.get(
            name, "kpconv_fused_kernel<Cin=%d>" % cin)
        nbytes = kpconv_alg_bytes(info["Nq"], info["Ns"], info["K"], cin, cout)
        flops = sum(kpconv_flops(info["Nq"], info["K"], cin, cout))
    elif name in ("gemm_f32", "gemm_x3", "gemm_x3r"):
        # the contraction families (+ their split-K reduce kernel): operand-split form on the bf16 matrix cores -- its tile kernel
        # (gemm_x3_kernel) and, since round 5, its resident-W persistent form for the tall layers of the fine levels
        # (gemm_x3r_kernel: a streaming kernel, HBM-bound) --, LDS-DMA fp32 MFMA tile kernel (whatever d3f_gemm_x3 cannot address)
        key = {"gemm_x3": "gemm_x3_kernel", "gemm_x3r": "gemm_x3r_kernel"}.get(name, "gemm_dma_kernel")
        nbytes = 4.0 * (info["M"] * info["K"] + info["K"] * info["N"] + info["M"] * info["N"])
        flops = 2.0 * info["M"] * info["N"] * info["K"]
    elif name == "nb_search":  # SURVEY §8(d) bytes_alg = 12*(Nq+Ns) + 4*Nq*K_out
        key = "nb_search_kernel<first_only=%d>" % info["first_only"]
        nbytes = 12.0 * (info["Nq"] + info["Ns"]) + 4.0 * info["Nq"] * (1 if info["first_only"] else info["width"])
    elif name == "nb_grid_build":   # read the supports, write the cell-sorted float4 copy + the index permutation
        key = "nb_grid_build"
        nbytes = (12.0 + 16.0 + 4.0) * info["Ns"]
    elif name == "grid_subsample":  # 12 N + 12 M
        key = "grid_subsample"
        nbytes = 12.0 * info["N"] + 12.0 * (info["M"] or 0)
    elif name == "ind_max_pool":    # index matrix + every finer-level row once + the pooled rows
        key = "maxpool_kernel"
        nbytes = 4.0 * (info["N2"] * info["K"] + info["N1"] * info["C"] + info["N2"] * info["C"])
    elif name == "detect_head":     # index matrix + last_unary output read once + descriptors and scores written
        key = "head32_kernel (+ per-cloud max)"
        nbytes = 4.0 * (info["N"] * info["K"] + 2 * info["N"] * info["C"] + info["N"])
    else:
        return None
    return key, float(nbytes), float(flops)


def accumulate_families(cfg, timed):
    """timed: [(record name, info, ms)] -> {family key: totals}; pure (tests/test_host_logic.py feeds it synthetic records)."""
    fam = {}
    for name, info, ms in timed:
        c = classify_record(cfg, name, info)
        if c is None:
            continue
        key, nbytes, flops = c
        f = fam.setdefault(key, dict(ms=0.0, launches=0, bytes=0.0, flops=0.0, roof_ms=0.0, multi=name in MULTI_LAUNCH_OPS))
        f["ms"] += ms
        f["launches"] += 1
        f["bytes"] += nbytes
        f["flops"] += flops
        f["roof_ms"] += 1e3 * max(flops / (matrix_peak(key) * 1e12), nbytes / (HBM_PEAK_GBS * 1e9))
    return fam


def matrix_peak(family):
    """TFLOP/s of ALGORITHMIC (fp32) flops the family's matrix instructions could deliver at best."""
    return MFMA_X3_PEAK_TF if family.startswith("gemm_x3") else MFMA_F32_PEAK_TF


def select_dominant(fam):
    """Family keys by descending time, and the headline one: the largest SINGLE-KERNEL family (never a multi-launch group)."""
    order = sorted(fam, key=lambda k: -fam[k]["ms"])
    single = [k for k in order if not fam[k]["multi"]]
    return order, (single[0] if single else None)


def instrumented_pass(cfg, step, raws, Fp, npass, device):
    """Same stack shape as the timed region: F fragments per pass, op by op instead of a replayed graph (HIP events cannot
    be placed between the nodes of a graph).  -> (dominant-family roofline, all families, per-KPConv-layer table)."""
    import torch
    from d3feat_amd import ops
    nprof = npass * Fp                      # fragments covered
    ops.PROFILE = []
    for i in range(npass):
        ops.PROFILE.append(("step", {}, None, None))
        step([raws[(i * Fp + j) % len(raws)] for j in range(Fp)] if Fp > 1 else raws[i % len(raws)])
    torch.cuda.synchronize(device)
    recs, ops.PROFILE = ops.PROFILE, None
    per_step_agg, per_step_gemm = [], []
    timed = []
    for name, info, s, e in recs:
        if name == "step":
            per_step_agg.append([])
            per_step_gemm.append([])
            continue
        ms = s.elapsed_time(e)
        timed.append((name, info, ms))
        if name == "kpconv_aggregate":
            per_step_agg[-1].append((info, ms))
        elif name in ("kpconv_fused_c1", "kpconv_fused32", "kpconv_fused"):
            per_step_agg[-1].append((dict(info, fused=True), ms))
        elif name in ("gemm_f32", "gemm_x3", "gemm_x3r"):
            per_step_gemm[-1].append((dict(info, kernel={"gemm_x3": "gemm_x3_kernel", "gemm_x3r": "gemm_x3r_kernel"}.get(name, "gemm_dma_kernel")), ms))
    fam = accumulate_families(cfg, timed)

    traffic, traffic_src, traffic_stale = load_traffic()
    mfma, mfma_src, mfma_stale = load_counters("*_mfma_counters.json")

    def describe(name):
        d = fam[name]
        avg_ms = d["ms"] / d["launches"]
        # which roof binds the family: algorithmic flops against the fp32 matrix peak or algorithmic bytes against HBM,
        # whichever takes longer (the level-0 unary contractions move 10 flop per byte: HBM; the deep ones 100+: MFMA)
        peak_tf = matrix_peak(name)
        t_f, t_b = d["flops"] / (peak_tf * 1e12), d["bytes"] / (HBM_PEAK_GBS * 1e9)
        tf_s = d["flops"] / d["launches"] / (avg_ms * 1e-3) / 1e12
        gb_s = d["bytes"] / d["launches"] / (avg_ms * 1e-3) / 1e9
        if t_f >= t_b:
            r = dict(kernel=name, bound="mfma", achieved=round(tf_s, 3), peak=round(peak_tf, 1), unit="TFLOP/s",
                     frac=round(tf_s / peak_tf, 5), traffic=None)
        else:
            r = dict(kernel=name, bound="hbm", achieved=round(gb_s, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                     frac=round(gb_s / HBM_PEAK_GBS, 5), traffic=None)
        if name.startswith("gemm_x3"):
            r["note"] = ("frac = achieved / peak with peak = the roof of the instruction stream this kernel issues (dense bf16 MFMA peak / 6 "
                         "products per fp32 product = %.1f TFLOP/s of fp32 products); against the fp32 MFMA peak of %.1f TFLOP/s -- the "
                         "roof the contraction family was priced against in rounds 1-3 (0.43, 0.50) -- the same achieved figure is %.3f"
                         % (peak_tf, MFMA_F32_PEAK_TF, tf_s / MFMA_F32_PEAK_TF))
            # algorithmic = fp32 products and sums (2 M N K); the kernel ISSUES six exact bf16 products per fp32 product
            r["matrix_pipe"] = dict(instruction="v_mfma_f32_32x32x16_bf16", products_per_fp32_product=X3_PRODUCTS,
                                    issued_tflops=round(X3_PRODUCTS * tf_s, 1), dense_peak=MFMA_BF16_PEAK_TF,
                                    peak_is="dense bf16 MFMA peak / 6 = the fp32-product rate this instruction stream can reach at best",
                                    fp32_mfma_peak=MFMA_F32_PEAK_TF, achieved_over_fp32_mfma_peak=round(tf_s / MFMA_F32_PEAK_TF, 4),
                                    measured_limit="operand traffic L2 -> CU (~9 TB/s) and launch size: DESIGN.md section 5")
        # matrix-pipe utilisation from the SQ counters (one `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES ... GRBM_GUI_ACTIVE` pass of this
        # same command at the headline's --batch 12 on one slot, the launches of the timed region only, tools/gpu_visit.sh mfma ->
        # tools/pmc_mfma.py; rounds up to r06_v22 averaged over the warm-up and F = 1 latency replays too): busy cycles of the matrix pipes / (1024 SIMDs x
        # the launch's cycles), time-weighted over the kernel's template instances.  It is a count of issued MFMAs (32 busy cycles
        # per v_mfma_f32_32x32x16_bf16, 16 per 16x16x32), so it must equal issued flops / (1024 flops per cycle and SIMD): the
        # `issued_frac_clock_free` column of the counter file is that figure from SQ_INSTS_VALU_MFMA_MOPS_BF16 of the same pass.
        # Against this line's own issued_tflops / 2516.6 it differs by (a) the fragments per launch when the pass ran another batch, (b) the
        # clock (2516.6 assumes 2.4 GHz; the chip runs 2.0-2.5 under this load) and (c) the counter pass serialising the streams.
        if mfma is not None:
            fm = (mfma.get("__families__") or {}).get(name.split("<")[0].split(" ")[0])
            if fm:
                r["mfma_busy"] = fm["mfma_busy"]
                r["mfma_busy_issued_frac_of_the_same_pass"] = fm.get("issued_frac_clock_free")
                r["mfma_busy_fragments_per_launch"] = int(mfma.get("__fragments_per_launch__", 4))
                r["mfma_busy_timed_region_only"] = bool(mfma.get("__timed_region_only__", False))
                r["mfma_busy_source"] = mfma_src
                r["mfma_busy_stale"] = mfma_stale
        r["alg_flops_per_launch"] = int(d["flops"] / d["launches"])
        r["alg_bytes_per_launch"] = int(d["bytes"] / d["launches"])
        r["tflops"], r["hbm_gbs"] = round(tf_s, 3), round(gb_s, 2)
        # per launch the tighter of the two roofs, summed: the share of the family's time that no implementation could remove
        r["frac_of_launchwise_roof"] = round(d["roof_ms"] / d["ms"], 5)
        r["avg_launch_us"] = round(avg_ms * 1e3, 2)
        r["launches_per_step"] = round(d["launches"] / nprof, 2)
        r["ms_per_step"] = round(d["ms"] / nprof, 4)
        r["fragments_per_launch"] = Fp
        # HBM traffic per launch: from the newest committed pair of `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes
        # of this same command (tools/gpu_visit.sh pmc -> tools/pmc_summary.py, corrections of MI355X_MICROARCH.md §HBM applied
        # there); launch-weighted mean over the template instantiations of the kernel.  `traffic_stale`: the counter file was
        # collected for different kernel sources than the ones running now.
        if traffic is not None:
            base = name.split("<")[0].split(" ")[0]
            names = ("gemm_x3r_kernel",) if name.startswith("gemm_x3r") else ("gemm_x3_kernel",) if name.startswith("gemm_x3") else \
                ("gemm_dma_kernel", "gemm_fast_kernel", "gemm_f32_kernel") if name.startswith("gemm") else (base,)
            ent = [v for k, v in traffic.items() if isinstance(v, dict) and k.split("<")[0] in names
                   and "traffic_bytes_per_launch" in v]
            # families that are ONE template instance take that instance's counters, not the launch-weighted mean of all
            # instances of the base name (r04: both nb_search variants and both kpconv_agg_vec4 widths reported one number)
            m = re.match(r"(kpconv_fused_kernel|kpconv_agg_vec4)<Cin=(\d+)>", name)     # template argument = lanes per query = Cin / 4
            prefix = None
            if m:
                prefix = "%s<%d," % (m.group(1), int(m.group(2)) // 4)
            m = re.match(r"nb_search_kernel<first_only=(\d)>", name)
            if m:
                prefix = "nb_search_kernel<%s," % ("true" if m.group(1) == "1" else "false")
            if prefix:
                ent = [v for k, v in traffic.items() if isinstance(v, dict) and "traffic_bytes_per_launch" in v and k.startswith(prefix)]
            nl = sum(e["launches"] for e in ent)
            if nl:
                # the counter run's launches held F_pmc fragments each, this pass's hold Fp: traffic scales with the rows
                f_pmc = int(traffic.get("__fragments_per_launch__", Fp) or Fp)
                r["traffic"] = int(sum(e["traffic_bytes_per_launch"] * e["launches"] for e in ent) / nl * Fp / f_pmc)
                r["traffic_fragments_per_launch_of_counter_run"] = f_pmc
                r["traffic_timed_region_only"] = bool(traffic.get("__timed_region_only__", False))
                r["traffic_source"] = traffic_src
                r["traffic_stale"] = traffic_stale
        return r

    order, dominant = select_dominant(fam)
    roofs = [describe(k) for k in order]
    for r in roofs:
        r["multi_launch_group"] = fam[r["kernel"]]["multi"]
    roof = dict(next(r for r in roofs if r["kernel"] == dominant))
    roof["timed_kernels_ms_per_step"] = {k: round(v["ms"] / nprof, 4) for k, v in sorted(fam.items())}
    roof["timing"] = "HIP events around each launch in an op-by-op pass over the same stacked shapes (not inside the replayed graph)"
    # ms per KPConv layer (call order inside a step = network order): aggregation + its contraction
    nl = len(per_step_agg[0])
    layers = []
    for li in range(nl):
        infos = [st[li][0] for st in per_step_agg]
        agg_ms = float(np.mean([st[li][1] for st in per_step_agg]))
        gem = []
        if not infos[0].get("fused") and infos[0]["Cin"] > 1:
            for st_a, st_g in zip(per_step_agg, per_step_gemm):
                d = st_a[li][0]
                cand = [m for (g, m) in st_g if g["M"] == d["Nq"] and g["K"] == cfg.num_kernel_points * d["Cin"]]
                if cand:
                    gem.append(cand[0])
        g_ms = float(np.mean(gem)) if gem else 0.0
        layers.append(dict(layer=li, Nq=int(np.mean([d["Nq"] for d in infos])), Ns=int(np.mean([d["Ns"] for d in infos])),
                           K=infos[0]["K"], Cin=infos[0]["Cin"], fused=bool(infos[0].get("fused")), agg_ms=round(agg_ms, 4),
                           gemm_ms=round(g_ms, 4) if gem else None, total_ms=round(agg_ms + g_ms, 4),
                           fragments_per_launch=Fp, total_ms_per_fragment=round((agg_ms + g_ms) / Fp, 4)))
    # the contraction launches one by one (launch order of a pass; mean over the passes): which shapes sit under which roof
    if per_step_gemm and roof is not None:
        shapes = []
        for j, (g, _) in enumerate(per_step_gemm[-1]):
            ms_j = float(np.mean([st[j][1] for st in per_step_gemm if len(st) == len(per_step_gemm[-1])]))
            fl = 2.0 * g["M"] * g["N"] * g["K"]
            by = 4.0 * (g["M"] * g["K"] + g["K"] * g["N"] + g["M"] * g["N"])
            pk = matrix_peak(g.get("kernel", "gemm_dma_kernel"))
            roof_ms = 1e3 * max(fl / (pk * 1e12), by / (HBM_PEAK_GBS * 1e9))
            shapes.append(dict(M=int(g["M"]), N=int(g["N"]), K=int(g["K"]), kernel=g.get("kernel", "gemm_dma_kernel"),
                               us=round(ms_j * 1e3, 1),
                               tflops=round(fl / (ms_j * 1e-3) / 1e12, 1), gbs=round(by / (ms_j * 1e-3) / 1e9, 0),
                               bound="mfma" if fl / (pk * 1e12) > by / (HBM_PEAK_GBS * 1e9) else "hbm",
                               frac_of_roof=round(roof_ms / ms_j, 3)))
        roof["contraction_launches"] = shapes
    # algorithmic flops per fragment of the two matrix-pipe families, as launched (shapes of the instrumented pass)
    fam_flops = {"gemm": sum(v["flops"] for k, v in fam.items() if k.startswith("gemm")) / nprof,
                 "kpconv": sum(v["flops"] for k, v in fam.items() if k.startswith("kpconv")) / nprof}
    return roof, roofs, layers, fam_flops


def load_counters(pattern):
    """Newest committed counter summary under profiles/ matching `pattern` -> (object, file name, stale?)."""
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)),
                   key=lambda q: [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", os.path.basename(q))])
    if not cands:
        return None, None, None
    try:
        tj = json.load(open(cands[-1]))
    except Exception:
        return None, None, None
    return tj, os.path.basename(cands[-1]), tj.get("__source_hash__") != source_hash()


def load_traffic():
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.json")),
                   key=lambda q: [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", os.path.basename(q))])  # v10 after v9
    if not cands:
        return None, None, None
    try:
        tj = json.load(open(cands[-1]))
    except Exception:
        return None, None, None
    return tj, os.path.basename(cands[-1]), tj.get("__source_hash__") != source_hash()


# ---------------------------------------------------------------------------------------------------------------------
def cpu_baseline(wl, limits, raws_host, one_thread=True):
    """The same step on the host cores of this box (reported baseline, not the target).  Geometry: the reference's own
    C++ (oracle/_ref) on 1 thread (the reference ops are single-threaded and test_3dmatch.py:55 uses one input thread).
    Network: the torch-CPU restatement of the TF graph on every core, and once on 1 thread.  -> (object, references)."""
    import torch
    from oracle import clib, parity as par
    cfg, W = wl.cfg, wl.W
    use_ref = clib.ref_available()
    co = clib.COracle()
    rl = clib.RefLib() if use_ref else None
    gsub = par.geometry_ops(co, rl)[2]

    def reference(item):
        """One step on the host, stage 0 included in t_geometry where the step has one."""
        if not wl.two and wl.stage0:
            return par.fragment_reference(cfg, W, item, limits, co=co, rl=rl)
        t = time.perf_counter()
        clouds = [gsub(c, np.float32(cfg.first_subsampling_dl)) for c in item] if wl.two else [item, item]
        t = time.perf_counter() - t
        ref = par.fragment_reference(cfg, W, None, limits, co=co, rl=rl, clouds=clouds)
        ref["t_geometry"] += t
        return ref
    ncores = os.cpu_count() or 1
    from oracle import network_np as onp

    # ---- a stable host leg (0.36 ... 0.82 fragments/s across visits, the 64-thread figure once BELOW the
    # 1-thread one).  The torch-CPU graph stops scaling well before a 256-thread box is full and collapses when its threads wander
    # across sockets: every thread of this process is pinned to ONE contiguous block of cores for the duration of the leg, the
    # intra-op thread count is the best of a short probe (16 / 32 / 64 on that block), two warm-ups precede the timed steps, and
    # the reported value is the better of {pinned multi-thread, 1 thread}.
    def pin(cpus):
        try:
            for tid in os.listdir("/proc/self/task"):
                try:
                    os.sched_setaffinity(int(tid), cpus)
                except OSError:
                    pass
            return True
        except Exception:
            return False
    try:
        avail = sorted(os.sched_getaffinity(0))
    except Exception:
        avail = list(range(ncores))
    block = avail[:min(len(avail), 64)]
    reference(raws_host[0])     # warm-up 1 (page-in, thread pools: the pool's threads must exist before they can be pinned)
    pinned = pin(set(block))
    first = reference(raws_host[0])     # warm-up 2 (pinned)
    probe = {}
    for nt in sorted({min(len(block), t) for t in (16, 32, 64)}):
        torch.set_num_threads(nt)
        onp.forward(cfg, W, first["inp"])
        t = time.perf_counter()
        onp.forward(cfg, W, first["inp"])
        probe[nt] = time.perf_counter() - t
    nthreads = min(probe, key=probe.get)
    torch.set_num_threads(nthreads)
    refs = [reference(r) for r in raws_host]
    pre = np.asarray([r["t_geometry"] for r in refs])
    net = np.asarray([r["t_network"] for r in refs])
    tot = pre + net

    def stats(a):
        return {"median": round(float(np.median(a)), 4), "p10": round(float(np.percentile(a, 10)), 4),
                "p90": round(float(np.percentile(a, 90)), 4), "n": int(len(a))}
    net1 = None
    if one_thread:
        torch.set_num_threads(1)
        t1 = []
        for r in refs[:2]:
            t = time.perf_counter()
            onp.forward(cfg, W, r["inp"])
            t1.append(time.perf_counter() - t)
        net1 = float(np.median(t1))
        torch.set_num_threads(nthreads)
    if pinned:
        pin(set(avail))
    v_multi = float(len(refs) * wl.frames / tot.sum())
    v_one = wl.frames / (float(np.median(pre)) + net1) if net1 is not None else None
    best_one = v_one is not None and v_one > v_multi
    out = {"value": round(v_one if best_one else v_multi, 4), "unit": "fragments/s", "cores": 1 if best_one else nthreads,
           "host_cores": ncores,
           # geometry = the reference's own C++ (oracle/_ref) when available; the network half has no runnable reference
           # (TensorFlow 1 is not installable here), it is the torch-CPU restatement -> "port" for the sum
           "kind": "port", "geometry_kind": "reference" if use_ref else "port",
           "sample": "%d step(s) of the same workload after 2 warm-ups, every thread pinned to cores %d-%d; geometry (stage-0 subsample "
                     "+ pyramid) by %s on 1 thread: %.3f s/fragment; network = torch-CPU restatement of the TF graph on %d threads "
                     "(best of %s; box: %d cores): %.3f s/fragment; value = the better of {that, 1 thread}"
                     % (len(refs), block[0], block[-1], "the reference's own C++ (oracle/_ref)" if use_ref else "the C restatement",
                        float(pre.mean()), nthreads, "/".join("%d: %.2f s" % (k, v) for k, v in sorted(probe.items())), ncores,
                        float(net.mean())),
           "geometry_s": stats(pre), "network_s": stats(net), "fragment_s": stats(tot),
           "value_multithread": round(v_multi, 4), "threads_probe_s": {str(k): round(v, 3) for k, v in sorted(probe.items())},
           "network_1thread_s": round(net1, 3) if net1 is not None else None,
           "value_1thread": round(v_one, 4) if v_one is not None else None, "pinned": bool(pinned)}
    return out, refs


def parity_check(cfg, engine, step, raws_dev, refs, device, tol):
    """The CPU sample's fragments through the SAME execution the timed region used (engine: F fragments per replay, all
    slots, graph path), compared with the oracle results the CPU leg just produced."""
    import torch
    from oracle import parity as par
    n = len(refs)
    worst = dict(points_equal=True, idx_equal=True, idx_tie_rows=0, desc_max_abs=0.0, score_max_abs=0.0)

    def fold(c):
        worst["points_equal"] &= c["points_equal"]
        worst["idx_equal"] &= c.get("idx_equal", True)
        worst["idx_tie_rows"] += c.get("idx_tie_rows", 0)
        worst["desc_max_abs"] = max(worst["desc_max_abs"], c["desc_max_abs"])
        worst["score_max_abs"] = max(worst["score_max_abs"], c["score_max_abs"])
    if engine is None:
        for raw, ref in zip(raws_dev, refs):
            rec = step(raw).cpu().numpy()
            fold(par.compare_fragment(ref, rec[:, :3], rec[:, 3:35], rec[:, 35:36]))
        how = "eager op-by-op path"
    else:
        S, F = len(engine.slots), engine.F
        i = 0
        while i < n:                              # one wave of up to S replays in flight, like the timed loop
            wave = []
            for sl in range(S):
                if i >= n:
                    break
                nb = min(F, n - i)
                engine.submit(sl, raws_dev[i:i + nb])
                wave.append((sl, i, nb))
                i += nb
            for sl, i0, nb in wave:
                fb0 = engine.fallbacks
                outs = engine.fetch(sl, packed=True)
                slot = engine.slots[sl]
                graph_path = engine.fallbacks == fb0 and not engine.mirror   # (a fallback's pyramid is not the slot's)
                nb0 = engine.reference_order_flat(sl)[cfg.num_layers].cpu().numpy() if graph_path else None
                total = int(slot.pts.n_dev.item()) if graph_path else None
                row0 = 0
                for j in range(nb):
                    rec = outs[j].cpu().numpy()
                    fold(par.compare_fragment(refs[i0 + j], rec[:, :3], rec[:, 3:35], rec[:, 35:36], nb0=nb0, row0=row0,
                                              total=total))
                    row0 += rec.shape[0]
        how = "graph engine, %d fragment(s) per replay, %d replays in flight" % (F, S)
    ok = bool(worst["points_equal"] and worst["idx_equal"] and worst["desc_max_abs"] <= tol and worst["score_max_abs"] <= tol)
    return {"ok": ok, "fragments": n, "points_equal": worst["points_equal"], "idx_equal": worst["idx_equal"],
            "idx_rows_differing_only_inside_bit_equal_distance_ties": worst["idx_tie_rows"],
            "desc_max_abs": float("%.3e" % worst["desc_max_abs"]), "score_max_abs": float("%.3e" % worst["score_max_abs"]),
            "tolerance": tol, "against": "oracle (reference C++ geometry when built + torch-CPU restatement of the TF graph)",
            "execution": how, "engine_fallbacks": engine.fallbacks if engine is not None else None}


if __name__ == "__main__":
    main()
