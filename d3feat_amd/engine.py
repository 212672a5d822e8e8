"""Fragment engine: the whole hot path of one fragment as ONE replayable HIP graph, several fragments in flight.

What the reference does per fragment is one `sess.run` (utils/tester.py:196-199) preceded by the CPU input pipeline
(datasets/common.py:1301-1413): raw cloud -> voxel subsample -> self-pair -> 5-level pyramid -> KPFCNN -> descriptors
and scores.  On the MI355X that is ~350 small kernels; launched one by one from the host with five data-dependent
synchronisations the GPU idles a third of the time.  Here instead:

  * every size that depends on the data stays in HBM (capacity mode of the C ABI, include/d3feat_amd.h): the launch
    sequence of a fragment is fixed by its CAPACITY class, not by its point count;
  * the sequence -- the very same Python model / pyramid code the eager path runs -- is captured once per slot into a
    HIP graph (torch.cuda.CUDAGraph: PyTorch is the stream / graph / allocator plumbing) and replayed per fragment:
    host cost per fragment = one copy-in, one graph launch, one 16-word status read-back;
  * `slots` graphs live on separate streams, so consecutive fragments overlap and the small deep-level kernels of one
    fragment fill the CUs the other leaves idle.
A fragment that does not fit its capacity class (or needs the large neighbour-ordering budget) raises a device-side
flag; `fetch` then recomputes it through the eager path, so results never depend on the capacities.  When the flagged
replay held several fragments they are first isolated (one short replay each): only the outliers go eager.
"""
import warnings

import numpy as np
import torch

from . import _lib, ops
from . import tf_custom_ops as tfo
from .datasets.common import FragmentDataset
from .models.KPFCNN_model import KernelPointFCNN


def level_caps(n0_cap, num_layers, ratio=0.4, clouds=2):
    """Row capacities of the stacked self-pair pyramid: level 0 holds 2*n0_cap rows, every further level `ratio` of the
    previous.  Grid subsampling at a doubled cell size keeps ~0.25-0.27 of surface samples; a cloud that
    exceeds a capacity is flagged on the device and recomputed by the eager path, so the ratio trades memory / idle
    workgroups against fallbacks, never correctness."""
    caps = [clouds * int(n0_cap)]
    for _ in range(1, num_layers):
        caps.append(max(int(np.ceil(caps[-1] * ratio)), 256))
    return caps


def level_hints(n0_hint, num_layers, ratio=0.26, clouds=2):
    """Expected row counts per level (launch planning only, e.g. the K split of the skinny deep-layer contractions)."""
    h = [clouds * int(n0_hint)]
    for _ in range(1, num_layers):
        h.append(max(int(h[-1] * ratio), 64))
    return h


class _Slot:
    pass


class FragmentEngine:
    def __init__(self, config, weights, neighborhood_limits, raw_cap=320000, n0_cap=40000, level_ratio=0.4, slots=2,
                 device=None, seed=42, n0_hint=None, mirror_self_pair=False, streams=None, two_clouds=False, batch=1,
                 bf16=False, bf16_features=False, stage0=True, internal_order=None):
        """raw_cap / n0_cap: raw points / voxels per FRAGMENT that a slot can take (a fragment beyond them is recomputed by
        the eager path).
        batch: fragments per graph replay.  The per-fragment cost of this path is dominated by the ~270 dependent launches
        of one replay, not by bytes or flops, so F independent fragments are stacked into ONE pyramid / network pass
        ([c_1; c_1; c_2; c_2; ...], lens on the device -- the reference's own stacking mechanism, datasets/common.py:453-496,
        with batch_num = F): every per-cloud result is unchanged (searches, subsampling and the head's normalisation are per
        batch element) while the launch chain is paid once per F fragments.
        mirror_self_pair=False: the stacked self-pair [cloud; cloud] is computed row by row, exactly the work of the
        reference's test generators (datasets/ThreeDMatch.py:190-192).  True: the pair's two halves are identical by
        construction (per-cloud searches, per-cloud head normalisation), so ONE copy is computed (stack of one cloud) and the
        outputs are mirrored into the stacked layout -- same results to fp32 summation order, half the work.
        two_clouds=True: every fragment is a pair of DIFFERENT clouds (the KITTI test generator, datasets/KITTI.py:94-106):
        submit(slot, (raw_a, raw_b)); raw_cap / n0_cap then bound the SUM over the two clouds.
        bf16=True: every unary / unfused KPConv contraction runs with bf16 operands and fp32 accumulation (ops.bf16_contraction;
        BASELINE configs[4]) -- NOT the parity path: results differ from fp32 by the operand rounding.
        bf16_features=True (implies bf16): the activations between the layers are additionally STORED as bfloat16 -- "bf16
        features with MFMA contraction"; arithmetic inside every kernel stays fp32.
        internal_order (default: on; D3F_INTERNAL_ORDER=0 switches it off): inside a replay every level is kept in the cell order of
        its own neighbour grid -- index matrices, point arrays and activations (datasets/common.py: _descriptor_input_internal) --
        so that the rows a workgroup gathers are neighbours in memory too; the last kernel of the sequence writes the records
        back in the reference's row order.  Results are those of the reference numbering (bit for bit in this implementation:
        every row's arithmetic is unchanged); reference_order_flat(slot) renumbers the slot's pyramid back for parity checks.
        stage0=False: the submitted clouds are ALREADY at the first subsampling resolution (the reference's scripts subsample
        before the dataset sees a cloud: demo_registration.py:24, datasets/ThreeDMatch.py:349) -- no stage-0 voxelisation, the
        cloud is stacked with itself as it is; raw_cap is then the voxel capacity n0_cap.
    Weights are captured BY ADDRESS: a replayed graph reads the model's tensors and their packed copies (transposed / pre-split planes,
    made once per tensor: ops._packed_on_tensor) through raw pointers.  `refresh_weights(values)` is the supported way to change
    them under live graphs: device tensors and packed copies are rewritten IN PLACE (same addresses), so the next replay computes
    with the new weights.  `.data =` / `set_()` re-bindings are not seen by a captured graph at all.
    """
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.cfg, self.device = config, device
        self.limits = np.asarray(neighborhood_limits, np.int32)
        self.raw_cap, self.n0_cap = int(raw_cap), int(n0_cap)
        self.mirror = bool(mirror_self_pair)
        self.two = bool(two_clouds)
        if internal_order is None:
            import os
            internal_order = os.environ.get("D3F_INTERNAL_ORDER", "1") != "0"
        self.internal = bool(internal_order)
        self.stage0 = bool(stage0)
        if not self.stage0:
            if self.mirror or self.two:
                raise ValueError("stage0=False is implemented for stacked self-pairs only")
            self.raw_cap = self.n0_cap
        self.F = int(batch)
        self.bf16 = bool(bf16) or bool(bf16_features)
        self.bf16_features = bool(bf16_features)
        if self.two and self.mirror:
            raise ValueError("mirror_self_pair and two_clouds exclude each other")
        if self.F < 1 or 2 * self.F * (2 if self.two else 1) > _lib.MAX_BATCH:
            raise ValueError("batch out of range")
        self.nin = self.F * (2 if self.two else 1)            # clouds handed to the stage-0 subsampling per replay
        # clouds of a fragment's stack whose records a submit(out=...) destination receives: what the reference's testers keep of
        # a stacked self-pair is its FIRST cloud (utils/tester.py:208-229: in_batches[0]); both frames of a pair of different clouds
        self.keep_clouds = 2 if self.two else 1
        clouds = self.F * (1 if (self.mirror or self.two) else 2)
        self.level_ratio = float(level_ratio)
        self.caps = level_caps(n0_cap, config.num_layers, level_ratio, clouds)
        self.cap_units = clouds                                # a level's capacity is `clouds` per-fragment capacities
        self.n0_hint = int(n0_hint if n0_hint is not None else n0_cap / 1.3)
        self.hints = level_hints(self.n0_hint, config.num_layers, clouds=clouds)
        self.model = KernelPointFCNN(None, config, weights=weights, seed=seed, device=device)
        # eager fallback path (also the warm-up that uploads the weights before any capture)
        self._eager_ds = FragmentDataset([], fast=True)
        self._eager_ds.device = device
        self._eager_ds.neighborhood_limits = self.limits
        self._eager_map = self._eager_ds.get_tf_mapping(config)
        # stand-in for the missing fragments of a partial batch: a tiny cloud (a handful of voxels, no flags raised)
        g = torch.Generator(device="cpu").manual_seed(1)
        self._dummy = (torch.rand((256, 3), generator=g) * (6.0 * config.first_subsampling_dl)).to(device)
        # one HIP stream per slot (pass `streams` to share them between engines: the runtime multiplexes every stream of the
        # process onto a few hardware queues, so idle extra streams still cost concurrency)
        self.neighbor_cap = 192            # hits a query can order in LDS on the fast path (sticky upgrade, see fetch)
        self.slots = [self._build_slot(streams[i] if streams else None) for i in range(int(slots))]
        self.fallbacks = 0             # fragments that took the eager path
        self.isolated = 0              # flagged multi-fragment replays that were split into single-fragment replays
        self.fragments = 0
        self._hit_overflows = 0
        self._warned = False

    def refresh_weights(self, values):
        """New values ({checkpoint name: array}) for some of the model's variables while captured graphs are alive: every
        device tensor (plain copies, folded batch-norm vectors, stacked branch weights) and every packed copy (transposed /
        pre-split bf16 planes) is rewritten in place -- nothing a captured graph points at is freed or moved -- and the next
        replay of every slot uses them.  No replay may be in flight (fetch every submitted slot first)."""
        assert not any(sl.busy for sl in self.slots), "refresh_weights with replays in flight"
        torch.cuda.synchronize(self.device)
        touched = self.model.variables.update_in_place(values)
        n = ops.refresh_packed_weights(touched)
        torch.cuda.synchronize(self.device)
        return n

    # ---- the fixed launch sequence -------------------------------------------------------------------------------
    def _sequence(self, sl):
        cfg = self.cfg
        if self.stage0:
            # the raw clouds are read IN PLACE through a table of addresses (sl.raw_ptrs): a cloud that already lives in HBM is
            # never copied into the slot; one that arrives from the host (or as file records) is written to the slot's staging
            # buffer sl.raw and its address there goes into the table
            sub, sub_l, st0 = ops.batch_grid_subsample_async(None, sl.raw_len, cfg.first_subsampling_dl, self.F * self.n0_cap,
                                                             status=sl.status0, m_hint=self.F * self.n0_hint,
                                                             elem_cap=self.n0_cap, clouds=sl.raw_ptrs, n_cap=self.F * self.raw_cap)
        else:
            sub, sub_l = sl.raw, sl.raw_len            # already at first_subsampling_dl: the stack is made of the clouds as fed
            sub.n_hint = self.F * self.n0_hint
        if self.mirror or self.two:
            pts, lens = sub, sub_l     # the stack as subsampled: lens = [m_1 .. m_F] (mirror) or [m_a1, m_b1, ...] (two clouds)
        else:
            pts, lens = ops.stack_self_pair(sub, sub_l)        # [c_1; c_1; c_2; c_2; ...]
        flat = sl.map(pts, None, None, None, lens, ("a", "a"), pts)
        with ops.bf16_contraction(self.bf16, features=self.bf16_features):
            desc, score = self.model.run(flat)
        # the pyramid itself stays readable after a replay (parity checks, calibration): static buffers of the graph
        # (internal numbering: in the cell order of every level -- reference_order_flat renumbers it back)
        sl.flat, sl.level_lengths = flat, sl.ds.level_lengths
        row_map = None
        if self.internal:
            row_map = sl.ds.orders[0]                 # internal row -> reference row of level 0
            ipts = flat[0]
            ipts.n_dev = pts.n_dev
            sl.orders = sl.ds.orders
        # one 144-byte record [xyz | desc | score] per point: a fragment's result is one contiguous block (fetch(packed=True))
        # (submit(out=...): the records of a fragment's kept clouds go straight to the caller's buffer instead, through sl.dst_ptrs)
        per = 1 if self.mirror else 2                       # stack entries per fragment
        sl.packed = ops.pack_descriptors(flat[0] if self.internal else pts, desc, score, lens=lens, group=per, keep=self.keep_clouds,
                                         dst=sl.dst_ptrs, row_map=row_map)
        if self.internal:
            # the separate outputs of fetch(packed=False) in the reference's row order: column views of the record block
            desc, score = sl.packed[:, 3:3 + desc.shape[1]], sl.packed[:, 3 + desc.shape[1]:]
        return pts, desc, score, sl.ds.static_status, lens

    def _build_slot(self, stream=None):
        dev = self.device
        sl = _Slot()
        sl.stream = stream if stream is not None else torch.cuda.Stream(device=dev)
        sl.raw = torch.zeros((self.F * self.raw_cap, 3), dtype=torch.float32, device=dev)
        # per-replay uploads, ONE block [addresses of the nin clouds (int64) | destinations of the F fragments' records (int64) |
        # lengths of the clouds (int32)] and its pinned host mirror
        nw = 2 * self.nin + 2 * self.F
        sl.meta_dev = torch.zeros((nw + self.nin,), dtype=torch.int32, device=dev)
        sl.meta_host = torch.zeros((nw + self.nin,), dtype=torch.int32).pin_memory()
        sl.raw_ptrs = sl.meta_dev[: 2 * self.nin].view(torch.int64)
        sl.dst_ptrs = sl.meta_dev[2 * self.nin: nw].view(torch.int64)
        sl.raw_len = sl.meta_dev[nw:]
        sl.host_ptrs = sl.meta_host[: 2 * self.nin].view(torch.int64)
        sl.host_dst = sl.meta_host[2 * self.nin: nw].view(torch.int64)
        sl.host_n = sl.meta_host[nw:]
        sl.status0 = torch.zeros((2,), dtype=torch.int32, device=dev)
        sl.ds = FragmentDataset([], fast=True)
        sl.ds.device = dev
        sl.ds.neighborhood_limits = self.limits
        sl.ds.caps = self.caps
        sl.ds.hints = self.hints
        sl.ds.cap_units = self.cap_units
        # every fragment is its own reference stack (a pair; one cloud when mirrored): the head's per-cloud normalisation
        # must not see the stack mates (models/D3Feat.py:84-85, datasets/common.py:453-496)
        sl.ds.stack_group = 1 if self.mirror else 2
        sl.ds.internal_order = self.internal
        sl.ds._neighbor_cap = self.neighbor_cap
        sl.cap = self.neighbor_cap
        sl.map = sl.ds.get_tf_mapping(self.cfg)
        sl.busy = False
        sl.raw_src = None
        # warm-up on tiny synthetic clouds (uploads weights, sizes the allocator), then capture
        with torch.cuda.stream(sl.stream):
            g = torch.Generator(device="cpu").manual_seed(0)
            per = min(2048, self.raw_cap)
            warm = (torch.rand((per * self.nin, 3), generator=g) * torch.tensor([1.0, 1.0, 0.05])).to(dev)
            sl.raw[: per * self.nin].copy_(warm)
            sl.raw_len.fill_(per)
            sl.raw_ptrs.copy_(torch.tensor([sl.raw.data_ptr() + 12 * per * i for i in range(self.nin)], dtype=torch.int64))
            with ops.private_workspace():
                _, _, _, w_status, w_lens = self._sequence(sl)
            w_status[:, 1].zero_()        # the searches' flag words are sticky: the replay's last node clears them (pack_status)
        sl.stream.synchronize()
        # one packed read-back per replay: [n_total, status0(2), status(k,2)..., lens(nb)]; the packing copies are nodes of
        # the graph (no host calls per replay), only the copy to the host follows the replay
        sl.nl = w_lens.numel()
        sl.nstat = 1 + 2 + 2 * w_status.shape[0]
        sl.host_stat = torch.zeros((sl.nstat + sl.nl,), dtype=torch.int32).pin_memory()
        sl.dev_stat = torch.zeros((sl.nstat + sl.nl,), dtype=torch.int32, device=dev)
        sl.graph = torch.cuda.CUDAGraph()
        with ops.private_workspace() as pw:
            with torch.cuda.graph(sl.graph, stream=sl.stream):
                sl.pts, sl.desc, sl.score, sl.status, sl.lens = self._sequence(sl)
                # [n_total | status0 | statuses | lens] in one launch, the sticky status words cleared for the next replay
                ops.pack_status(sl.dev_stat, [sl.pts.n_dev, sl.status0, sl.status, sl.lens], clear=sl.status)
        sl.keep = pw.kept          # scratch buffers referenced by the graph
        assert sl.lens.numel() == sl.nl and 1 + 2 + 2 * sl.status.shape[0] == sl.nstat
        sl.done = torch.cuda.Event()
        return sl

    # ---- per-fragment API -------------------------------------------------------------------------------------------
    def submit(self, slot, raw, out=None):
        """Start a replay on slot `slot`; returns immediately.  `raw`: one fragment (float32 [n,3] on the device or the host;
        a pair (raw_a, raw_b) with two_clouds), or -- batch > 1 -- a list of up to `batch` fragments.
        out (with fetch(packed=True)): one destination per fragment, a contiguous f32[rows >= its kept rows, 36] device tensor -- the
        [xyz | desc | score] records of the fragment's kept clouds (the first cloud of a stacked self-pair; both of two different
        clouds) are written THERE by the replay's last kernel and fetch returns a view of it: no copy into the caller's shard."""
        sl = self.slots[slot]
        assert not sl.busy, "slot %d still holds an unfetched fragment" % slot
        if sl.cap != self.neighbor_cap:
            # the ordering budget was raised after repeated overflows: re-capture this slot once, on its own stream
            sl = self.slots[slot] = self._build_slot(sl.stream)
        single = not isinstance(raw, list)
        frags = [raw] if single else list(raw)
        if not 1 <= len(frags) <= self.F:
            raise ValueError("submit: %d fragments for a batch-%d engine" % (len(frags), self.F))
        sl.single, sl.raw_src, sl.nfrag = single, frags, len(frags)
        sl.busy = True
        outs = None
        if out is not None and not self.mirror:
            outs = [out] if (single and isinstance(out, torch.Tensor)) else list(out)
            assert len(outs) == len(frags) and all(o.is_cuda and o.dtype == torch.float32 and o.is_contiguous() and o.dim() == 2
                                                   and o.shape[1] == 36 for o in outs)
        sl.out_dst = outs
        sl.out_given = outs is not None
        parts = []
        for fr in frags:
            p = list(fr) if self.two else [fr]
            if self.two and len(p) != 2:
                raise ValueError("two_clouds engine: a fragment is a pair (raw_a, raw_b)")
            parts += p
        per_frag = 2 if self.two else 1
        sizes = [sum(int(x.shape[0]) for x in parts[i * per_frag:(i + 1) * per_frag]) for i in range(len(frags))]
        if max(sizes) > self.raw_cap or min(int(x.shape[0]) for x in parts) == 0:
            sl.oversize = True
            return
        sl.oversize = False
        parts += [self._dummy] * (self.nin - len(parts))        # partial batch: stand-ins, dropped again in fetch
        for i, p in enumerate(parts):
            sl.host_n[i] = int(p.shape[0])
        cur = torch.cuda.current_stream(self.device)
        with torch.cuda.stream(sl.stream):
            # `raw` may have been produced on the caller's stream, and (mirror mode) the previous result of this slot may
            # still be being copied out there
            sl.stream.wait_stream(cur)
            o = 0
            sl.keep_in = []
            for i, p in enumerate(parts):
                n = int(p.shape[0])
                if isinstance(p, ops.RawRecords):      # file records: bytes to the device, xyz decoded in place (stage-0 ingestion)
                    p.decode(self.device, out=sl.raw[o:o + n])
                    addr = sl.raw[o:].data_ptr()
                elif self.stage0 and isinstance(p, torch.Tensor) and p.is_cuda and p.device == self.device:
                    # already in HBM: read where it is (the tensor is kept alive until the slot is fetched)
                    q = p if (p.dtype == torch.float32 and p.is_contiguous()) else p.to(torch.float32).contiguous()
                    sl.keep_in.append(q)
                    addr = q.data_ptr()
                else:
                    if p.data_ptr() != sl.raw[o:].data_ptr():     # a producer may have written straight into the slot's buffer
                        sl.raw[o:o + n].copy_(p, non_blocking=True)
                    addr = sl.raw[o:].data_ptr()
                sl.host_ptrs[i] = addr
                o += n
            for f in range(self.F):
                sl.host_dst[f] = sl.out_dst[f].data_ptr() if (sl.out_dst is not None and f < len(sl.out_dst)) else 0
            sl.meta_dev.copy_(sl.meta_host, non_blocking=True)
            sl.graph.replay()          # ends by packing [n_total | status0 | statuses | lens] into dev_stat
            sl.host_stat.copy_(sl.dev_stat, non_blocking=True)
            sl.done.record(sl.stream)

    def fetch(self, slot, packed=False):
        """Wait for slot `slot`; -> (points f32[2n,3], descriptors f32[2n,32], scores f32[2n,1]) device tensors of the stacked
        pair (views into the slot's buffers: valid until the slot is submitted again); a list of such tuples when the
        submit was given a list.  packed=True: one f32[2n, 36] tensor of [xyz | desc | score] records per fragment instead of
        the tuple (a contiguous view: the unit the sharded runner keeps and gathers)."""
        sl = self.slots[slot]
        assert sl.busy, "slot %d is empty" % slot
        sl.busy = False
        # submit(out=...): every packed result of this fetch holds the KEPT clouds only, whichever path produced it
        kept_only = packed and getattr(sl, "out_given", False)

        def keep(rec):
            return rec if (not kept_only or self.two) else rec[: rec.shape[0] // 2]
        outs, flags = None, 0
        if not sl.oversize:
            sl.done.synchronize()
            st = sl.host_stat.numpy()
            flags = int(st[2]) | (int(np.bitwise_or.reduce(st[4:sl.nstat:2])) if sl.nstat > 3 else 0)
            if flags == 0:
                lens = [int(x) for x in st[sl.nstat:]]
                per = 1 if self.mirror else 2                   # stack entries per fragment
                outs, o = [], 0
                for i in range(sl.nfrag):
                    n = sum(lens[i * per:(i + 1) * per])
                    if packed and sl.out_dst is not None:
                        nk = sum(lens[i * per:i * per + self.keep_clouds])
                        if nk <= sl.out_dst[i].shape[0]:
                            outs.append(sl.out_dst[i][:nk])              # written in place by the replay (kept clouds only)
                        else:       # the destination was too small for this fragment: its rows are lost, recompute (never in bench)
                            outs = None
                            break
                    elif packed:
                        r = sl.packed[o:o + n]
                        outs.append(torch.cat([r, r]) if self.mirror else r)
                    else:
                        p, d, s = sl.pts[o:o + n], sl.desc[o:o + n], sl.score[o:o + n]
                        if self.mirror:   # stacked layout of the reference: both halves hold the cloud
                            p, d, s = torch.cat([p, p]), torch.cat([d, d]), torch.cat([s, s])
                        outs.append((p, d, s))
                    o += n
        if outs is None and sl.nfrag > 1:
            # A flagged replay of several fragments: the flags are per stacked call, not per cloud, so the fragments are
            # isolated -- each goes through the graph again as a batch of one (stand-ins fill the stack: a fraction of a full
            # replay's time), and only those that still do not fit take the eager path.  One outlier costs its own eager run
            # plus nfrag short replays, not nfrag eager runs.
            # (an in-place producer's fragment lives in the slot's own raw buffer, which the re-submits below overwrite)
            base, end = sl.raw.data_ptr(), sl.raw.data_ptr() + sl.raw.numel() * 4
            srcs = [tuple(x.clone() if isinstance(x, torch.Tensor) and x.is_cuda and base <= x.data_ptr() < end else x for x in fr)
                    if isinstance(fr, (tuple, list)) else
                    (fr.clone() if isinstance(fr, torch.Tensor) and fr.is_cuda and base <= fr.data_ptr() < end else fr)
                    for fr in sl.raw_src]
            single = sl.single
            self.isolated += 1
            outs = []
            for fr in srcs:
                if (sum(int(x.shape[0]) for x in fr) if self.two else int(fr.shape[0])) > self.raw_cap:
                    self.fragments += 1
                    self.fallbacks += 1
                    o = self.run_eager(fr)
                    outs.append(keep(ops.pack_descriptors(*o)) if packed else o)
                    continue
                self.submit(slot, [fr])
                o = self.fetch(slot, packed)[0]                 # (counts the fragment, and its fallback if it takes one)
                outs.append(keep(o).clone() if packed else tuple(t.clone() for t in o))   # the slot's buffers are reused at once
            return outs[0] if single else outs
        self.fragments += sl.nfrag
        if outs is None:
            # capacity exceeded / large ordering budget needed / degenerate cloud: the eager path decides (and raises the
            # reference-level errors where they apply)
            self.fallbacks += sl.nfrag
            if not sl.oversize and (flags & _lib.ST_HIT_OVERFLOW) and self.neighbor_cap < _lib.NEIGHBOR_CAP:
                # a query with more in-radius supports than the fast LDS budget: a property of the data set (dense clouds),
                # not of one fragment -- after the second such replay every slot is re-captured with the full budget
                self._hit_overflows += 1
                if self._hit_overflows >= 2:
                    self.neighbor_cap = _lib.NEIGHBOR_CAP
                    self._eager_ds._neighbor_cap = _lib.NEIGHBOR_CAP
            if not self._warned and self.fallbacks >= 8 and self.fallbacks * 10 > self.fragments:
                self._warned = True
                warnings.warn("FragmentEngine: %d of %d fragments took the eager fallback (capacities raw_cap=%d n0_cap=%d "
                              "too small for this data, or degenerate clouds): throughput is that of the op-by-op path"
                              % (self.fallbacks, self.fragments, self.raw_cap, self.n0_cap))
            outs = [self.run_eager(fr) for fr in sl.raw_src]
            if packed:
                outs = [keep(ops.pack_descriptors(*o)) for o in outs]
        return outs[0] if sl.single else outs

    def reference_order_flat(self, slot):
        """The pyramid of the slot's LAST replay as the reference numbers it (datasets/common.py:1301-1413): points, neighbors, pools,
        upsamples of every level, renumbered back from the internal cell order (plain torch indexing on the real rows; a test / parity
        aid, never on the timed path).  Without the internal numbering: the slot's own flat list.  Rows beyond a level's real count
        are not defined."""
        sl = self.slots[slot]
        if not self.internal:
            return sl.flat
        L = self.cfg.num_layers
        n = [int(sl.flat[l].n_dev.item()) if getattr(sl.flat[l], "n_dev", None) is not None else int(sl.flat[l].shape[0]) for l in range(L)]
        order = [sl.orders[l][: n[l]].long() for l in range(L)]

        def rows_back(x, l):            # internal row j -> reference row order[l][j]
            out = torch.empty_like(x[: n[l]])
            out[order[l]] = x[: n[l]]
            return out

        def values_back(x, l, pad):     # entries: positions in level l's cell order -> reference indices; padding stays
            ext = torch.cat([order[l], torch.tensor([0], device=x.device)])
            v = x.long()
            valid = (v >= 0) & (v < n[l])
            return torch.where(valid, ext[torch.where(valid, v, torch.full_like(v, n[l]))], v).to(torch.int32)
        flat = list(sl.flat)
        for l in range(L):
            flat[l] = ops._tag(rows_back(sl.flat[l], l), sl.flat[l])
            nb = sl.flat[L + l]
            if nb.shape[0] > 0:
                flat[L + l] = values_back(rows_back(nb, l), l, None)
            pool = sl.flat[2 * L + l]
            if pool.shape[0] > 0:
                flat[2 * L + l] = values_back(rows_back(pool, l + 1), l, None)
            up = sl.flat[3 * L + l]
            if up.shape[0] > 0:
                flat[3 * L + l] = values_back(rows_back(up, l), l + 1, None)
        return flat

    def run_eager(self, raw):
        if isinstance(raw, ops.RawRecords):
            raw = raw.decode(self.device)
        elif isinstance(raw, (tuple, list)):
            raw = tuple(r.decode(self.device) if isinstance(r, ops.RawRecords) else r for r in raw)
        if self.two:
            subs = [tfo.grid_subsampling(p if p.is_cuda else p.to(self.device), self.cfg.first_subsampling_dl) for p in raw]
            pts = torch.cat(subs, 0)
            lens = ops.as_lens([int(x.shape[0]) for x in subs], self.device)
        else:
            raw = raw if raw.is_cuda else raw.to(self.device)
            sub = tfo.grid_subsampling(raw, self.cfg.first_subsampling_dl) if self.stage0 else raw
            n = sub.shape[0]
            pts = torch.cat([sub, sub], 0)
            lens = ops.as_lens([n, n], self.device)
        flat = self._eager_map(pts, None, None, None, lens, ("a", "a"), pts)
        with ops.bf16_contraction(self.bf16, features=self.bf16_features):
            desc, score = self.model.run(flat)
        return pts, desc, score

    def run(self, raw, slot=0):
        self.submit(slot, raw)
        return self.fetch(slot)
