"""Sharded descriptor extraction: the reference's tester loop (utils/tester.py:196-229, test_3dmatch.py:55-69 -- one sess.run
per fragment, three result files each) as fragment-level data parallelism over the GPUs of one node.

One process per GPU (torchrun; backend "nccl" = RCCL over xGMI), no data-path collective:
  1. every rank learns the fragment list and the point count of every fragment (file headers only);
  2. static partition: greedy longest-processing-time by point count (parallel.shard_fragments), identical on every rank;
  3. start-up calibration: each rank histograms the neighbour counts of ITS fragments, one all_reduce(SUM) of the
     [layers, bins] histograms makes every rank derive the same neighborhood_limits a single process would
     (datasets/common.py:629-670 is a pure histogram sum);
  4. one FragmentEngine per rank (HIP-graph replays, several fragments in flight); every fragment's result is written by
     its owner in the reference's per-fragment layout and kept in HBM as [xyz | desc | score] records;
  5. ONE padded all_gather of the whole shards at the end (parallel.gather_shard) -- optional, the files are complete
     without it.
`make_engine` / `calibrate` are injectable so that the control flow runs under gloo on CPU tensors (tests/test_runner_gloo.py).
"""
import os

import numpy as np
import torch
import torch.distributed as dist

from . import parallel


def limits_from_histograms(hists, keep_ratio=0.8):
    """datasets/common.py:667-668."""
    hists = np.asarray(hists)
    hist_n = hists.shape[1]
    cumsum = np.cumsum(hists.T, axis=0)
    return np.sum(cumsum < (keep_ratio * cumsum[hist_n - 1, :]), axis=0).astype(np.int32)


def _gather_index_lists(mine, n_items, device):
    """Every rank's fragment indices (variable length) -> list over ranks."""
    rank, ws = parallel.world()
    if ws == 1:
        return [list(mine)]
    t = torch.full((n_items,), -1, dtype=torch.int64, device=device)
    if mine:
        t[: len(mine)] = torch.tensor(list(mine), dtype=torch.int64, device=device)
    out = torch.empty((ws * n_items,), dtype=torch.int64, device=device)
    parallel.all_gather_into(out, t)
    return [[int(v) for v in o if v >= 0] for o in out.view(ws, n_items).tolist()]


def run_sharded(fragment_ids, sizes, load, config, weights, make_engine, calibrate, device, out_root=None, gather=True,
                save=None, log=None, keep="first", overlap_chunk=0, dst=0):
    """fragment_ids: list of id strings ('<scene>/cloud_bin_<k>.ply'); sizes: raw point count per fragment (all ranks pass the
    same lists); load(i) -> float32 [n,3] raw cloud of fragment i (called by the owner only).
    make_engine(config, weights, limits, raw_cap, n0_cap_hint) -> engine with .F, .slots, submit(slot, [raw...]),
    fetch(slot, packed=True) -> list of record tensors [2n, 36] (stacked self-pair) in submission order.
    calibrate(raws) -> int64 histograms [layers, bins] of this rank's fragments.
    keep: what a fragment contributes to the shard that is gathered -- "first": the first cloud's records, which is what
    utils/tester.py:208-229 keeps of a stacked self-pair; "pair": the whole stacked block (KITTI pairs).  `save` always
    receives the whole block.
    overlap_chunk > 0: the shards are exchanged while they are produced, `overlap_chunk` fragments per asynchronous collective
    (parallel.ShardCollector overlapped mode; fixed stride = the engine's row capacity of one contribution, `engine.n0_cap` --
    required, the same number on every rank; a fragment beyond it, i.e. the engine's eager fallback of an oversize cloud, goes
    through the collector's trailing variable-length exchange).
    dst: the rank that receives the shards (default 0: north_star's "gather of descriptors only at the end"); None = every rank.
    -> dict(limits, mine, order (rank 0..W-1 -> fragment indices), shards (list over ranks of (records, frag_rows)) | None)."""
    assert keep in ("first", "pair")
    rank, world = parallel.world()
    n = len(fragment_ids)
    mine = parallel.shard_fragments(n, rank, world, sizes=sizes)
    # chunks every rank can exchange while it still produces: the shortest shard decides (same collective order everywhere)
    async_chunks = (min(len(parallel.shard_fragments(n, r, world, sizes=sizes)) for r in range(world)) // overlap_chunk
                    if overlap_chunk > 0 else 0)
    raws = {i: load(i) for i in mine}
    # ---- calibration: local histograms, summed over ranks
    hists = calibrate([raws[i] for i in mine])
    hists = parallel.allreduce_histograms(np.asarray(hists, np.int64), device)
    limits = limits_from_histograms(hists)
    # ---- one engine per rank, sized for the largest fragment of the WHOLE list (same graph shape on every rank)
    engine = make_engine(config, weights, limits, int(max(sizes) * 1.05) + 1024 if sizes else 1024)
    collector = None
    stride = 0
    if overlap_chunk > 0 and gather:
        if not hasattr(engine, "n0_cap"):
            raise ValueError("run_sharded(overlap_chunk > 0) needs engine.n0_cap: the per-fragment stride of the chunk collectives "
                             "must be the same number on every rank")
        stride = int(engine.n0_cap) * (1 if keep == "first" else 2)
    S, F = len(engine.slots), engine.F
    pending = [None] * S
    produced = []

    def drain(sl):
        nonlocal collector
        for i, rec in zip(pending[sl], engine.fetch(sl, packed=True)):
            if collector is None:
                if overlap_chunk > 0 and gather:
                    # fixed stride = the engine's row capacity of one contribution: the SAME number on every rank (the chunk
                    # collectives have one size; engines are built from the same arguments everywhere)
                    collector = parallel.ShardCollector(rows_cap=stride * max(len(mine), 1), width=rec.shape[1], device=rec.device,
                                                        chunk_frags=overlap_chunk, frag_rows=stride, async_chunks=async_chunks,
                                                        dst=dst)
                else:
                    collector = parallel.ShardCollector(rows_cap=max(int(rec.shape[0]) * max(len(mine), 1), 1), width=rec.shape[1],
                                                        device=rec.device, dst=dst)
            collector.add(rec[: rec.shape[0] // 2] if keep == "first" else rec)
            produced.append(i)
            if save is not None:
                save(fragment_ids[i], rec)
            if log is not None:
                log("rank %d: %s (%d rows)" % (rank, fragment_ids[i], rec.shape[0]))
        pending[sl] = None
    k = 0
    for b0 in range(0, len(mine), F):
        sl = k % S
        if pending[sl] is not None:
            drain(sl)
        batch = mine[b0:b0 + F]
        engine.submit(sl, [raws[i] if isinstance(raws[i], torch.Tensor) else torch.from_numpy(raws[i]) for i in batch])
        pending[sl] = batch
        k += 1
    for kk in range(k, k + S):
        if pending[kk % S] is not None:
            drain(kk % S)
    assert produced == list(mine)
    order = _gather_index_lists(mine, max(n, 1), device)
    shards = None
    if gather:
        if collector is None:     # a rank without fragments still takes part in every collective
            collector = (parallel.ShardCollector(rows_cap=1, width=36, device=device, chunk_frags=overlap_chunk, async_chunks=0,
                                                 frag_rows=stride, dst=dst)
                         if overlap_chunk > 0 else parallel.ShardCollector(rows_cap=1, width=36, device=device, dst=dst))
        shards = collector.gather()
    return dict(limits=limits, mine=list(mine), order=order, shards=shards, fallbacks=getattr(engine, "fallbacks", 0),
                isolated=getattr(engine, "isolated", 0))


def save_records_3dmatch(root):
    """save(fragment_id, records) writing the three files of utils/tester.py:215-229 for the FIRST cloud of the stacked pair."""
    from .utils.results import save_3dmatch_results

    def save(fid, rec):
        r = rec.cpu().numpy()
        save_3dmatch_results(root, fid, r[:, :3], r[:, 3:-1], r[:, -1:], r.shape[0] // 2)
    return save


def gpu_engine_factory(slots=4, batch=4, mirror=False):
    """make_engine for real GPUs: d3feat_amd.engine.FragmentEngine; n0_cap from a first-level estimate, fragments that
    exceed a capacity take the engine's eager fallback."""
    def make(config, weights, limits, raw_cap):
        from .engine import FragmentEngine
        n0_cap = max(int(raw_cap * 0.16), 4096)          # ~0.1 of the raw points survive the 0.03 m grid on 3DMatch fragments
        dev = torch.device("cuda", torch.cuda.current_device())
        return FragmentEngine(config, weights, limits, raw_cap=raw_cap, n0_cap=n0_cap, slots=slots, device=dev,
                              mirror_self_pair=mirror, batch=batch)
    return make


def gpu_calibrate(config):
    """calibrate(raws) for real GPUs: stage-0 subsample + untruncated pyramids of this rank's fragments
    (datasets/common.py:572-673 histogram part)."""
    def cal(raws):
        from . import tf_custom_ops as tfo
        from .datasets.common import FragmentDataset
        dev = torch.device("cuda", torch.cuda.current_device())
        hist_n = int(np.ceil(4 / 3 * np.pi * (config.density_parameter + 1) ** 3))
        if not raws:
            return np.zeros((config.num_layers, hist_n), np.int64)
        subs = [tfo.grid_subsampling(torch.from_numpy(np.ascontiguousarray(r, np.float32)).to(dev),
                                     config.first_subsampling_dl).cpu().numpy() for r in raws]
        ds = FragmentDataset(subs)
        ds.neighborhood_limits = np.full(config.num_layers, hist_n, np.int32)
        return ds.calibrate_neighbors(config, samples_threshold=10 ** 12)
    return cal
