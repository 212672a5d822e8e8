"""The N>1 path on CPU: two processes over the gloo backend exercise d3feat_amd/parallel.py exactly as bench.py /
a test driver use it on N GPUs over RCCL (same calls, CPU tensors): fragment sharding, the start-up histogram
all-reduce that makes every rank derive identical neighborhood_limits, and the final variable-length gather of
(xyz, descriptors, scores)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from d3feat_amd import parallel
    from oracle import network_np as onp
    try:
        assert parallel.world() == (rank, world)
        # ---- sharding: every fragment exactly once, same answer on every rank
        n_frag = 11
        sizes = np.random.default_rng(5).integers(15000, 45000, n_frag)
        mine = parallel.shard_fragments(n_frag, rank, world, sizes=sizes)
        t = torch.zeros(n_frag, dtype=torch.int64)
        t[mine] = 1
        dist.all_reduce(t)
        assert torch.all(t == 1)
        # ---- calibration: per-rank histograms summed == histogram of the union -> identical limits on all ranks
        hist_n = 905

        def rank_hist(r):
            rg = np.random.default_rng(100 + r)
            return np.stack([np.bincount(rg.integers(5, 60, 4000), minlength=hist_n)[:hist_n] for _ in range(5)]).astype(np.int64)
        summed = parallel.allreduce_histograms(rank_hist(rank))
        both = sum(rank_hist(r) for r in range(world))
        assert np.array_equal(summed, both)
        limits = onp.limits_from_histograms(summed)
        lt = torch.from_numpy(limits.astype(np.int64))
        gathered = [torch.zeros_like(lt) for _ in range(world)]
        dist.all_gather(gathered, lt)
        assert all(torch.equal(g, lt) for g in gathered)
        # ---- final gather: variable N per rank, payload intact and in rank order
        n = 1000 + 37 * rank
        g = torch.Generator().manual_seed(rank)
        xyz, desc, score = torch.rand(n, 3, generator=g), torch.rand(n, 32, generator=g), torch.rand(n, 1, generator=g)
        res = parallel.gather_descriptors(xyz, desc, score)
        assert len(res) == world
        for r, (x, d, s) in enumerate(res):
            gr = torch.Generator().manual_seed(r)
            nr = 1000 + 37 * r
            assert x.shape == (nr, 3) and d.shape == (nr, 32) and s.shape == (nr, 1)
            assert torch.equal(x, torch.rand(nr, 3, generator=gr))
            assert torch.equal(d, torch.rand(nr, 32, generator=gr))
            assert torch.equal(s, torch.rand(nr, 1, generator=gr))
        # ---- empty shard on one rank (fewer fragments than ranks) still gathers
        e = 0 if rank == 1 else 5
        res = parallel.gather_descriptors(torch.ones(e, 3), torch.ones(e, 32), torch.ones(e, 1))
        assert [r[0].shape[0] for r in res] == [5 if r != 1 else 0 for r in range(world)]
        # ---- whole-shard gather: every fragment of every rank arrives, in order, with its row count
        col = parallel.ShardCollector(rows_cap=64, width=36)           # deliberately small: grows
        nfr = 3 + rank
        for i in range(nfr):
            gi = torch.Generator().manual_seed(1000 * rank + i)
            col.add(torch.rand(50 + 7 * i + rank, 36, generator=gi))
        shards = col.gather()
        assert len(shards) == world
        for r, (rec, fr) in enumerate(shards):
            assert fr == [50 + 7 * i + r for i in range(3 + r)] and rec.shape == (sum(fr), 36)
            o = 0
            for i, n in enumerate(fr):
                gi = torch.Generator().manual_seed(1000 * r + i)
                assert torch.equal(rec[o:o + n], torch.rand(n, 36, generator=gi))
                o += n
        # ---- the same in OVERLAPPED mode: fixed stride per fragment, chunks of 2 fragments exchanged while the fragments are added
        # (1 early chunk on both ranks: the shorter shard -- 3 fragments -- decides), buffer deliberately too small (grows), both
        # result forms (compacted / per-fragment views), and a second run after reset()
        col = parallel.ShardCollector(rows_cap=100, width=36, chunk_frags=2, frag_rows=80, async_chunks=3 // 2)
        for rep in range(2):
            for i in range(nfr):
                gi = torch.Generator().manual_seed(1000 * rank + i)
                col.add(torch.rand(50 + 7 * i + rank, 36, generator=gi))
            assert len(col._works) == 1
            for compact in ((True, False) if rep == 0 else (False,)):
                shards = col.gather(compact=compact)
                for r, (rec, fr) in enumerate(shards):
                    assert fr == [50 + 7 * i + r for i in range(3 + r)]
                    parts = rec if not compact else None
                    o = 0
                    for i, n in enumerate(fr):
                        gi = torch.Generator().manual_seed(1000 * r + i)
                        want = torch.rand(n, 36, generator=gi)
                        assert torch.equal(parts[i] if parts is not None else rec[o:o + n], want)
                        o += n
            col.reset()
        # ---- rank-0 gather (north_star: "gather of descriptors only at the end"): only rank 0 receives, in both modes; a
        # fragment LARGER than the fixed stride (the engine's eager fallback of an oversize cloud) survives the overlapped mode
        # through the trailing variable-length exchange (it used to raise mid-run on one rank and hang the others)
        def frag(r, i):
            n = 50 + 7 * i + r + (60 if (r == 1 and i == 2) else 0)       # rank 1's fragment 2: 125 rows > stride 80
            return torch.rand(n, 36, generator=torch.Generator().manual_seed(1000 * r + i))
        for dst in (0, None):
            for mode in ("plain", "overlapped"):
                col = (parallel.ShardCollector(rows_cap=64, width=36, dst=dst) if mode == "plain" else
                       parallel.ShardCollector(rows_cap=100, width=36, chunk_frags=2, frag_rows=80, async_chunks=1, dst=dst))
                for i in range(nfr):
                    col.add(frag(rank, i))
                shards = col.gather(compact=False) if mode == "overlapped" else col.gather()
                assert len(shards) == world
                for r, (rec, fr) in enumerate(shards):
                    assert fr == [frag(r, i).shape[0] for i in range(3 + r)]          # the row counts reach every rank
                    if dst is not None and rank != dst and r != rank:
                        assert rec is None                                              # ... the payload only the receiver
                        continue
                    parts = rec if mode == "overlapped" else list(torch.split(rec, fr))
                    for i, n in enumerate(fr):
                        assert torch.equal(parts[i], frag(r, i)), (dst, mode, r, i)
                if dst == 0 and rank != 0 and mode == "overlapped":
                    assert col._recv_chunks is None                                     # no receive memory off the root
                assert torch.equal(col.records(), torch.cat([frag(rank, i) for i in range(nfr)]))
        open(os.path.join(out_dir, "ok_%d" % rank), "w").write("ok")
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_gloo(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / ("ok_%d" % r)) for r in range(world))


def test_single_process_degenerates_to_identity():
    from d3feat_amd import parallel
    assert parallel.world() == (0, 1)
    h = np.arange(10, dtype=np.int64).reshape(2, 5)
    assert parallel.allreduce_histograms(h) is h
    x = torch.rand(4, 3)
    res = parallel.gather_descriptors(x, torch.rand(4, 32), torch.rand(4, 1))
    assert len(res) == 1 and res[0][0] is x
