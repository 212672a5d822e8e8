"""The sharded runner (d3feat_amd/runner.py -- the reference's tester loop as fragment-level data parallelism) under gloo on
CPU tensors, world size 2: same control flow as on N GPUs over RCCL; the engine and the calibration are stand-ins (the HIP
path itself has no CPU form).  Checks: every fragment is produced exactly once, by the rank that owns it; the limits derived
after the histogram all-reduce are identical on every rank and equal the single-process limits; every rank ends up with
every rank's whole shard; the per-fragment files exist."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

N_FRAG = 9
SIZES = [int(x) for x in np.random.default_rng(3).integers(200, 900, N_FRAG)]
IDS = ["scene%d/cloud_bin_%d.ply" % (i % 2, i) for i in range(N_FRAG)]


def _load(i):
    return np.random.default_rng(100 + i).random((SIZES[i], 3)).astype(np.float32)


def _records(raw):
    n = min(len(raw), 50)
    p = torch.from_numpy(np.concatenate([raw[:n], raw[:n]]))
    return torch.cat([p, p.sum(1, keepdim=True).repeat(1, 32), p[:, :1] * 2], 1)


class _Engine:
    """Stand-in with the FragmentEngine interface the runner uses."""
    F = 2

    def __init__(self):
        self.slots = [None, None]
        self.held = {}
        self.fallbacks = 0
        # row capacity of one fragment's contribution = the stride of the overlapped exchange.  Deliberately BELOW the largest
        # fragments: those stand for the eager fallback of an oversize cloud and must survive the exchange
        self.n0_cap = int(np.sort(np.asarray(SIZES))[len(SIZES) * 2 // 3])

    def submit(self, slot, raws):
        assert slot not in self.held and 1 <= len(raws) <= self.F
        self.held[slot] = [r.numpy() for r in raws]

    def fetch(self, slot, packed=False):
        assert packed
        return [_records(r) for r in self.held.pop(slot)]


def _hist(raws, layers=5, bins=905):
    h = np.zeros((layers, bins), np.int64)
    for r in raws:
        g = np.random.default_rng(len(r))
        for l in range(layers):
            h[l] += np.bincount(g.integers(5, 60, 300), minlength=bins)[:bins]
    return h


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir, overlap_chunk=0, dst=0):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from d3feat_amd import runner
    try:
        seen_limits = {}

        def make_engine(cfg, W, limits, raw_cap):
            seen_limits["limits"] = np.asarray(limits).copy()
            assert raw_cap >= max(SIZES)
            return _Engine()
        saved = []
        res = runner.run_sharded(IDS, SIZES, _load, None, None, make_engine, _hist, torch.device("cpu"),
                                 save=lambda fid, rec: saved.append(fid), overlap_chunk=overlap_chunk, dst=dst)
        want_limits = runner.limits_from_histograms(_hist([_load(i) for i in range(N_FRAG)]))
        assert np.array_equal(res["limits"], want_limits) and np.array_equal(seen_limits["limits"], want_limits)
        # ownership: a partition of the fragment list, the same on every rank
        assert sorted(i for o in res["order"] for i in o) == list(range(N_FRAG))
        assert res["order"][rank] == res["mine"] and saved == [IDS[i] for i in res["mine"]]
        # whole shards on every rank, fragment by fragment
        assert len(res["shards"]) == world
        for r, (rec, rows) in enumerate(res["shards"]):
            assert len(rows) == len(res["order"][r])
            assert rows == [_records(_load(i)).shape[0] // 2 for i in res["order"][r]]      # row counts reach every rank
            if dst is not None and rank != dst and r != rank:
                assert rec is None                     # rank-0 gather: the payload lands on the receiver only
                continue
            o = 0
            for i, n in zip(res["order"][r], rows):
                full = _records(_load(i))
                assert n == full.shape[0] // 2 and torch.equal(rec[o:o + n], full[:n])      # keep="first": the tester's rows
                o += n
            assert o == rec.shape[0]
        # per-fragment files in the reference's layout
        save = runner.save_records_3dmatch(os.path.join(out_dir, "res"))
        for i in res["mine"]:
            save(IDS[i], _records(_load(i)))
        open(os.path.join(out_dir, "ok_%d" % rank), "w").write(",".join(str(i) for i in res["mine"]))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("overlap_chunk,dst", [(0, 0), (2, 0), (3, 0), (0, None), (2, None)])
def test_sharded_runner_two_ranks_gloo(tmp_path, overlap_chunk, dst):
    """overlap_chunk > 0: the shards are exchanged in asynchronous chunks of that many fragments while they are produced (shards
    of 5 and 4 fragments: the shorter one decides how many chunks go early, the rest follows in gather)."""
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), overlap_chunk, dst), nprocs=world, join=True)
    owned = [open(tmp_path / ("ok_%d" % r)).read().split(",") for r in range(world)]
    assert sorted(int(i) for o in owned for i in o if i) == list(range(N_FRAG))
    for i in range(N_FRAG):
        scene, k = IDS[i].split("/")[0], int(IDS[i].split("_")[-1][:-4])
        for sub, name in (("descriptors", "cloud_bin_%d.D3Feat.npy" % k), ("keypoints", "cloud_bin_%d.npy" % k),
                          ("scores", "cloud_bin_%d.npy" % k)):
            assert os.path.exists(tmp_path / "res" / sub / scene / name), (sub, scene, name)
    d = np.load(tmp_path / "res" / "descriptors" / "scene0" / "cloud_bin_0.D3Feat.npy")
    assert d.shape[1] == 32


def test_sharded_runner_single_process():
    from d3feat_amd import runner
    res = runner.run_sharded(IDS, SIZES, _load, None, None, lambda c, w, l, r: _Engine(), _hist, torch.device("cpu"))
    assert res["mine"] == list(range(N_FRAG)) and res["order"] == [list(range(N_FRAG))]
    rec, rows = res["shards"][0]
    assert len(rows) == N_FRAG and rec.shape[0] == sum(rows)
    pair = runner.run_sharded(IDS, SIZES, _load, None, None, lambda c, w, l, r: _Engine(), _hist, torch.device("cpu"), keep="pair")
    assert pair["shards"][0][1] == [2 * n for n in rows]
