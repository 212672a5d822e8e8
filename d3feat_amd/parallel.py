"""Fragment-level data parallelism: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI).

The reference has no multi-GPU path; its unit of work -- one fragment = one sess.run
(utils/tester.py:196-199) -- is independent of every other, so the path shards with NO data-path collective:
  * `shard_fragments`     static partition of the fragment list (round-robin, or greedy by point count);
  * `allreduce_histograms` start-up only: neighbour-count histograms are summed so that every rank derives the same
                           neighborhood_limits as a single process would (datasets/common.py:629-670 is a pure sum);
  * `gather_descriptors`  the only exchange, once at the end: variable-length all_gather of
                           (xyz f32[N,3], desc f32[N,32], score f32[N,1]) = 144 B/point.  On a fully connected
                           8-GPU xGMI node this is one small message per peer (latency bound), so a single padded
                           all_gather is used rather than a ring of per-tensor collectives.
Works unchanged with backend "gloo" on CPU tensors (tests/test_parallel_gloo.py, world_size 2).
"""
import numpy as np
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_fragments(n_items, rank, world_size, sizes=None):
    """Indices of the fragments this rank processes.  sizes=None: round-robin.  With per-fragment point counts:
    greedy longest-processing-time assignment (deterministic, identical on every rank)."""
    if sizes is None:
        return list(range(rank, n_items, world_size))
    order = sorted(range(n_items), key=lambda i: (-int(sizes[i]), i))
    load = [0] * world_size
    mine = []
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        load[r] += int(sizes[i])
        if r == rank:
            mine.append(i)
    return sorted(mine)


def _host_staged():
    """backend gloo moves HOST memory: device tensors are staged through the host around every collective.  (The production
    backend is "nccl" = RCCL, device to device over xGMI; gloo with device tensors is the two-ranks-on-one-GPU test --
    RCCL refuses two ranks on one device -- and costs a synchronising copy each way.)"""
    return dist.get_backend() == "gloo"


def all_gather_into(out, inp, async_op=False):
    """dist.all_gather_into_tensor(out, inp) for device or host tensors under either backend; -> work handle or None."""
    if inp.is_cuda and _host_staged():
        h_out = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(h_out, inp.cpu())
        out.copy_(h_out)
        return None
    return dist.all_gather_into_tensor(out, inp, async_op=async_op)


def exchange_into(out, inp, dst=None, async_op=False):
    """The payload exchange of the path.  dst=None: every rank receives every rank's `inp` (all_gather into `out`
    [ranks * rows, W]).  dst=r: ONLY rank r receives (dist.gather -- under RCCL one point-to-point xGMI transfer per peer into
    the matching slice of `out`, nothing lands on the other ranks, whose `out` is None): BASELINE.json north_star's "gather of
    descriptors only at the end".  -> work handle or None."""
    if dst is None:
        return all_gather_into(out, inp, async_op=async_op)
    rank, ws = world()
    if inp.is_cuda and _host_staged():
        h_out = torch.empty((ws,) + tuple(inp.shape), dtype=inp.dtype) if rank == dst else None
        dist.gather(inp.cpu(), list(h_out.unbind(0)) if rank == dst else None, dst=dst)
        if rank == dst:
            out.view((ws,) + tuple(inp.shape)).copy_(h_out)
        return None
    glist = list(out.view((ws,) + tuple(inp.shape)).unbind(0)) if rank == dst else None
    return dist.gather(inp, glist, dst=dst, async_op=async_op)


def allreduce_histograms(hists, device=None):
    """Sum int64 histograms [layers, bins] over ranks (no-op for a single process)."""
    rank, ws = world()
    if ws == 1:
        return hists
    t = torch.as_tensor(np.ascontiguousarray(hists), dtype=torch.int64)
    if device is not None and not _host_staged():
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def gather_descriptors(xyz, desc, score):
    """All ranks receive every rank's (xyz, desc, score).  Inputs: f32[N,3], f32[N,C], f32[N,1] on this rank's
    device (N may differ per rank).  Returns a list (one entry per rank) of (xyz, desc, score) tensors."""
    rank, ws = world()
    if ws == 1:
        return [(xyz, desc, score)]
    dev = xyz.device
    C = desc.shape[1]
    n = torch.tensor([xyz.shape[0]], dtype=torch.int64, device=dev)
    sizes = [torch.zeros_like(n) for _ in range(ws)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    nmax = max(sizes)
    width = 3 + C + 1
    payload = torch.zeros((nmax, width), dtype=torch.float32, device=dev)
    payload[: xyz.shape[0], 0:3] = xyz
    payload[: xyz.shape[0], 3:3 + C] = desc
    payload[: xyz.shape[0], 3 + C:] = score.reshape(-1, 1)
    out = [torch.empty_like(payload) for _ in range(ws)]
    dist.all_gather(out, payload)
    return [(o[:s, 0:3], o[:s, 3:3 + C], o[:s, 3 + C:]) for o, s in zip(out, sizes)]


class ShardCollector:
    """This rank's results, kept in HBM as they are produced: one f32[rows, width] buffer of [xyz | desc | score] records
    (ops.pack_descriptors) plus the row count of every fragment.  `add` is one contiguous device-to-device copy (plumbing,
    issued on the current stream); `gather` is the path's only data collective.

    dst: who receives the shards.  None = every rank (all_gather: O(ranks) receive memory on every rank); an int = that rank
    only (the default of bench.py / runner.py: rank 0 -- north_star's "gather of descriptors only at the end"; the other ranks
    allocate no receive memory and `gather` returns (None, frag_rows) for the shards they did not receive).

    chunk_frags > 0 (with frag_rows = the row capacity of one fragment's contribution): OVERLAPPED mode.  Fragment k lives at
    the fixed rows [k * frag_rows, ...), so no size has to be agreed on before data moves: after every `chunk_frags` fragments
    the finished chunk is exchanged asynchronously (RCCL runs it on its own stream behind the compute; point-to-point xGMI:
    one 4.2 MB-per-fragment message per peer), `gather` only waits, sends the last partial chunk and exchanges the row counts.
    Chunk c of every rank lands in row block c of ONE receive buffer allocated once (grown by doubling; only on receiving
    ranks) -- no allocation per chunk.  A fragment larger than the stride (the engine's eager fallback of an oversize cloud
    returns more rows than its capacity) is kept aside and exchanged by one trailing variable-length collective that every rank
    enters iff any rank holds such a fragment -- the receivers recognise it by its row count > stride."""

    def __init__(self, rows_cap, width=36, device=None, chunk_frags=0, frag_rows=0, async_chunks=None, dst=None):
        # async_chunks: how many chunks are exchanged WHILE fragments are produced.  Collectives must be issued in the same
        # order on every rank, so with shards of different lengths (LPT partition) pass min over ranks of n_r // chunk_frags
        # (every rank can compute it: the partition is deterministic); None = no limit (equal shards: bench.py)
        self.async_chunks = async_chunks
        self.dst = dst
        self.chunk_frags, self.frag_rows_cap = int(chunk_frags), int(frag_rows)
        if self.chunk_frags > 0:
            assert self.frag_rows_cap > 0
            nchunks = max(-(-int(rows_cap) // (self.chunk_frags * self.frag_rows_cap)), 1)
            rows_cap = nchunks * self.chunk_frags * self.frag_rows_cap
        self.buf = torch.empty((int(rows_cap), int(width)), dtype=torch.float32, device=device)
        self.rows = 0
        self.frag_rows = []
        self._works = []
        self._recv_chunks = None       # [chunk capacity, ranks * chunk rows, W] on receiving ranks
        self._over = []                # records of the fragments larger than the stride, in order

    def receives(self):
        return self.dst is None or world()[0] == self.dst

    def reset(self):
        for w in self._works:
            if w is not None:
                w.wait()
        self.rows, self.frag_rows = 0, []
        self._works, self._over = [], []
        self._reserved = 0

    def _grow(self, need):
        grown = torch.empty((max(2 * self.buf.shape[0], need), self.buf.shape[1]), dtype=torch.float32, device=self.buf.device)
        grown[: self.rows].copy_(self.buf[: self.rows])
        self.buf = grown

    def _wait_all(self):
        for w in self._works:
            if w is not None:
                w.wait()

    def slots(self, nfrags):
        """Fixed-stride mode: the buffer rows reserved for the NEXT nfrags fragments, one f32[frag_rows, width] view each -- hand
        them to FragmentEngine.submit(out=...) and the replay writes the records in place; `add` of a tensor that already lives
        in its slot copies nothing.  None when the mode or the buffer's capacity does not allow it (add then copies, and grows).
        Fragments must be added in the order they were reserved."""
        if self.chunk_frags <= 0:
            return None
        k0, R = getattr(self, "_reserved", 0), self.frag_rows_cap
        k0 = max(k0, len(self.frag_rows))
        if (k0 + nfrags) * R > self.buf.shape[0]:
            return None
        self._reserved = k0 + nfrags
        return [self.buf[(k0 + j) * R:(k0 + j + 1) * R] for j in range(nfrags)]

    def add(self, packed):
        n = int(packed.shape[0])
        if self.chunk_frags > 0:
            k, R = len(self.frag_rows), self.frag_rows_cap
            if (k + 1) * R > self.buf.shape[0]:
                self._wait_all()               # the chunks in flight read the old buffer
                self.rows = k * R
                self._grow((k + self.chunk_frags) * R)
            if n > R:
                self._over.append(packed.clone())      # its slot stays unused; exchanged by the trailing collective
            elif n > 0 and packed.data_ptr() == self.buf[k * R:].data_ptr():
                pass                                   # written in place by the replay (slots)
            else:
                self.buf[k * R:k * R + n].copy_(packed, non_blocking=True)
            self.frag_rows.append(n)
            self.rows = (k + 1) * R
            if (k + 1) % self.chunk_frags == 0 and (self.async_chunks is None or len(self._works) < self.async_chunks):
                self._launch_chunk((k + 1) // self.chunk_frags - 1, async_op=True)
            return
        if self.rows + n > self.buf.shape[0]:
            self._grow(self.rows + n)
        self.buf[self.rows:self.rows + n].copy_(packed, non_blocking=True)
        self.rows += n
        self.frag_rows.append(n)

    def _recv_block(self, c):
        """Row block c of the receive buffer (receiving ranks only)."""
        ws = world()[1]
        rows = self.chunk_frags * self.frag_rows_cap
        cap = 0 if self._recv_chunks is None else self._recv_chunks.shape[0]
        if c >= cap:
            self._wait_all()                   # chunks in flight write the old buffer
            ncap = max(2 * cap, c + 1, self.buf.shape[0] // rows)
            grown = torch.empty((ncap, ws * rows, self.buf.shape[1]), dtype=torch.float32, device=self.buf.device)
            if cap:
                grown[:cap].copy_(self._recv_chunks)
            self._recv_chunks = grown
        return self._recv_chunks[c]

    def _launch_chunk(self, c, async_op):
        ws = world()[1]
        if ws == 1:
            return
        rows = self.chunk_frags * self.frag_rows_cap
        recv = self._recv_block(c) if self.receives() else None
        self._works.append(exchange_into(recv, self.buf[c * rows:(c + 1) * rows], dst=self.dst, async_op=async_op))

    def records(self):
        if self.chunk_frags > 0:
            R = self.frag_rows_cap
            over = iter(self._over)
            parts = [self.buf[k * R:k * R + n] if n <= R else next(over) for k, n in enumerate(self.frag_rows)]
            return torch.cat(parts) if parts else self.buf[:0]
        return self.buf[: self.rows]

    def gather(self, compact=True):
        """-> list over ranks of (records, frag_rows); records is None for a shard this rank did not receive (dst mode).
        Plain mode: the receive buffer is sized for what the ranks actually hold and kept (views of it are returned: valid until
        the next gather), so a repeated timed gather allocates nothing.  Overlapped mode: waits for the chunks in flight, sends
        the last partial chunk, exchanges the row counts (and the oversize fragments, if any rank has one); compact=False
        returns each rank's records as a LIST of per-fragment views (no compaction copy)."""
        if self.chunk_frags > 0:
            return self._gather_chunked(compact)

        def receive(rows_total):
            if getattr(self, "_recv", None) is None or self._recv.shape[0] < rows_total:
                self._recv = torch.empty((rows_total, self.buf.shape[1]), dtype=torch.float32, device=self.buf.device)
            return self._recv[:rows_total]
        return gather_shard(self.records(), self.frag_rows, backing=self.buf, receive=receive, dst=self.dst)

    def _gather_chunked(self, compact):
        rank, ws = world()
        C, R = self.chunk_frags, self.frag_rows_cap
        if ws == 1:
            over = iter(self._over)
            parts = [self.buf[k * R:k * R + n] if n <= R else next(over) for k, n in enumerate(self.frag_rows)]
            return [((torch.cat(parts) if parts else self.buf[:0]) if compact else parts, list(self.frag_rows))]
        dev = self.buf.device
        nf = torch.tensor([len(self.frag_rows)], dtype=torch.int64, device=dev)
        nfs = torch.empty((ws,), dtype=torch.int64, device=dev)
        all_gather_into(nfs, nf)
        nfs = [int(v) for v in nfs.tolist()]
        fmax = max(max(nfs), 1)
        # ranks hold different numbers of fragments (LPT shards): every rank takes part in ceil(fmax / C) chunk collectives
        nch = -(-fmax // C)
        if nch * C * R > self.buf.shape[0]:
            self._wait_all()
            self.rows = len(self.frag_rows) * R
            self._grow(nch * C * R)
        for c in range(len(self._works), nch):
            self._launch_chunk(c, async_op=False)
        self._wait_all()
        fr = torch.zeros((fmax,), dtype=torch.int64, device=dev)
        if self.frag_rows:
            fr[: len(self.frag_rows)] = torch.tensor(self.frag_rows, dtype=torch.int64).to(dev)
        frs = torch.empty((ws * fmax,), dtype=torch.int64, device=dev)
        all_gather_into(frs, fr)                      # the row counts go to every rank (a few bytes): all agree on what follows
        frs = frs.view(ws, fmax).tolist()
        # ---- trailing exchange of the fragments larger than the stride: entered by every rank iff any rank holds one
        over_rows = [[int(v) for v in frs[r][: nfs[r]] if v > R] for r in range(ws)]
        over_recv = None
        omax = max(sum(o) for o in over_rows)
        if omax > 0:
            payload = torch.zeros((omax, self.buf.shape[1]), dtype=torch.float32, device=dev)
            if self._over:
                mine = torch.cat(self._over)
                payload[: mine.shape[0]] = mine
            over_recv = torch.empty((ws * omax, self.buf.shape[1]), dtype=torch.float32, device=dev) if self.receives() else None
            exchange_into(over_recv, payload, dst=self.dst)
            if over_recv is not None:
                over_recv = over_recv.view(ws, omax, -1)
        out = []
        for r in range(ws):
            rows = [int(v) for v in frs[r][: nfs[r]]]
            if not self.receives() and r != rank:
                out.append((None, rows))
                continue
            parts, oo = [], 0
            local_over = iter(self._over)
            for k, n in enumerate(rows):
                if n > R:
                    if r == rank and over_recv is None:
                        parts.append(next(local_over))
                    else:
                        parts.append(over_recv[r, oo:oo + n])
                    oo += n
                elif r == rank and not self.receives():
                    parts.append(self.buf[k * R:k * R + n])
                else:
                    ch = self._recv_chunks[k // C].view(ws, C * R, -1)[r]
                    parts.append(ch[(k % C) * R:(k % C) * R + n])
            out.append(((torch.cat(parts) if parts else self.buf[:0]) if compact else parts, rows))
        return out


def gather_shard(records, frag_rows, backing=None, receive=None, dst=None):
    """Variable-length exchange of every rank's WHOLE shard: records f32[rows, W] (rows differ per rank) and the
    per-fragment row counts.  Sizes first (to every rank: a few bytes), then ONE payload collective into one
    [ranks * rows_max, W] tensor (dst=None: all_gather_into_tensor on every rank; dst=r: dist.gather, rank r only -- RCCL over
    xGMI on GPUs, gloo on CPU tensors; no per-rank staging copies).  `backing`: the buffer `records` is the head of -- when it
    holds rows_max rows the payload is sent from it in place (the rows past this rank's count are never read by the receiver),
    otherwise the records are padded into a fresh buffer.  `receive(rows)` -> contiguous f32[rows, W] to receive into (default:
    a fresh tensor).
    -> list over ranks of (records f32[rows_r, W] | None when not received here, frag_rows list)."""
    rank, ws = world()
    if ws == 1:
        return [(records, list(frag_rows))]
    dev = records.device
    W = records.shape[1]
    meta = torch.tensor([records.shape[0], len(frag_rows)], dtype=torch.int64, device=dev)
    metas = torch.empty((ws * 2,), dtype=torch.int64, device=dev)        # concatenated along dim 0 (gloo chunks it that way)
    all_gather_into(metas, meta)
    metas = [[int(v) for v in m] for m in metas.view(ws, 2).tolist()]
    rmax, fmax = max(m[0] for m in metas), max(max(m[1] for m in metas), 1)
    fr = torch.zeros((fmax,), dtype=torch.int64, device=dev)
    if frag_rows:
        fr[: len(frag_rows)] = torch.tensor(list(frag_rows), dtype=torch.int64).to(dev)
    frs = torch.empty((ws * fmax,), dtype=torch.int64, device=dev)
    all_gather_into(frs, fr)
    frs = frs.view(ws, fmax).tolist()
    if rmax == records.shape[0] and records.is_contiguous():
        payload = records
    elif backing is not None and backing.shape[0] >= rmax and backing.is_contiguous() and backing.data_ptr() == records.data_ptr():
        payload = backing[:rmax]
    else:
        payload = torch.zeros((rmax, W), dtype=torch.float32, device=dev)
        payload[: records.shape[0]] = records
    receives = dst is None or rank == dst
    out = None
    if receives:
        out = receive(ws * rmax) if receive is not None else torch.empty((ws * rmax, W), dtype=torch.float32, device=dev)
    exchange_into(out, payload, dst=dst)
    if receives:
        out = out.view(ws, rmax, W)
    return [((out[r, : m[0]] if receives else (records if r == rank else None)), [int(v) for v in f[: m[1]]])
            for r, (m, f) in enumerate(zip(metas, frs))]
