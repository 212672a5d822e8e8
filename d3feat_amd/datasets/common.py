"""Input pipeline of the D3Feat descriptor network on the MI355X: the 5-level pyramid of points / neighbour /
pooling / upsampling index matrices, neighbour-limit calibration and the test pipeline plumbing.

Mirrors the inference part of the reference's datasets/common.py (class Dataset):
    tf_batch_subsampling / tf_batch_neighbors          :67-72
    big_neighborhood_filter                            :399-406
    tf_get_batch_inds / tf_stack_batch_inds            :408-496
    calibrate_neighbors                                :572-673
    init_test_input_pipeline                           :776-857
    tf_descriptor_input                                :1301-1413
with torch tensors on the GPU instead of TF graph tensors and eager execution instead of tf.data.  A subclass
provides `get_batch_gen(split, config)` and `get_tf_mapping(config)` exactly like the reference's datasets
(demo_registration.py:30-110); generator tuples are
    (points f32[N,3], anc_keypts, pos_keypts, obj_inds, stack_lengths i32[B], ids, backup_points f32[N,3]).
"""
import time

import numpy as np
import torch

from .. import _lib, ops
from .. import tf_custom_ops


def tf_batch_subsampling(points, batches_len, sampleDl):
    """datasets/common.py:67-68."""
    return tf_custom_ops.batch_grid_subsampling(points, batches_len, sampleDl)


def tf_batch_neighbors(queries, supports, q_batches, s_batches, radius):
    """datasets/common.py:71-72."""
    return tf_custom_ops.batch_ordered_neighbors(queries, supports, q_batches, s_batches, radius)


def _host_lens(lens):
    return ops.host_lens(lens)


class Dataset:
    """Base class: holds `neighborhood_limits` and builds the network inputs."""

    def __init__(self, name):
        self.name = name
        self.path = ''
        self.label_to_names = {}
        self.num_classes = 0
        self.ignored_labels = np.array([])
        self.label_names = []
        self.label_to_idx = {}
        self.network_model = 'descriptor'
        self.num_threads = 1
        self.neighborhood_limits = None
        self.num_test = 0
        self.device = torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else None
        self.flat_inputs = None
        self.test_init_op = None

    # ---- utility methods ---------------------------------------------------------------------------------------
    def big_neighborhood_filter(self, neighbors, layer):
        """datasets/common.py:399-406: keep the first neighborhood_limits[layer] columns."""
        return neighbors[:, :int(self.neighborhood_limits[layer])]

    def tf_get_batch_inds(self, stacks_len):
        """datasets/common.py:408-451: [3, 2, 5] -> [0,0,0,1,1,2,2,2,2,2] (int32, on the device)."""
        lens = _host_lens(stacks_len)
        host = np.repeat(np.arange(len(lens), dtype=np.int32), lens)
        return torch.from_numpy(host).to(self.device)

    def tf_stack_batch_inds(self, stacks_len):
        """datasets/common.py:453-496: int32[B, max_len(+1)] rows arange padded with the total point count;
        an extra pad column when all lengths are equal."""
        lens = _host_lens(stacks_len)
        n, mx = sum(lens), max(lens)
        width = mx + (1 if n == mx * len(lens) else 0)
        host = np.full((len(lens), width), n, dtype=np.int32)
        p = 0
        for b, l in enumerate(lens):
            host[b, :l] = np.arange(p, p + l, dtype=np.int32)
            p += l
        return torch.from_numpy(host).to(self.device)

    # ---- the pyramid ------------------------------------------------------------------------------------------------
    def tf_descriptor_input(self, config, stacked_points, stacked_features, stacked_lengths, batch_inds,
                            exact_shapes=True, up_first_column_only=False, timings=None, caps=None):
        """datasets/common.py:1301-1413.

        exact_shapes=True reproduces the reference's output shapes: every index matrix has
        min(Kmax, neighborhood_limits[layer]) columns, which costs one host synchronisation per search.
        exact_shapes=False (the fast path used by the model) always allocates neighborhood_limits[layer] columns
        and lets the kernel pad with the shadow index -- identical semantics for every consumer
        (KPConv / max-pool / closest-pool / detection head all treat the shadow index as "no neighbour") -- and
        defers all status checks to one synchronisation at the end.  up_first_column_only additionally computes
        only the nearest neighbour for the upsampling matrices (the only column closest_pool reads,
        models/network_blocks.py:81).

        caps (list of per-level row capacities, implies exact_shapes=False): capacity mode.  Nothing is read back to the
        host: every level's point tensor has caps[l] rows and carries its real row count as `n_dev` (device int32),
        the subsamplings run through d3f_batch_grid_subsample_async, and the status words of all ops are left in
        `self.static_status` (int32[k,2] on the device) for ONE check by the caller.  The launch sequence is then
        independent of the data, which is what lets d3feat_amd.engine capture it in a HIP graph.
        """
        if caps is not None:
            exact_shapes = False
            if getattr(self, 'internal_order', False):
                return self._descriptor_input_internal(config, stacked_points, stacked_features, stacked_lengths, caps)
        dev = stacked_points.device
        first_points, first_lengths = stacked_points, stacked_lengths
        lens_dev = ops.as_lens(stacked_lengths, dev)
        stacked_lengths = lens_dev
        # batch weights (datasets/common.py:1307-1310) -- unused at inference, built on the host
        host_lens = _host_lens(lens_dev) if exact_shapes else None
        if exact_shapes:
            bw = (np.float32(min(host_lens)) / np.asarray(host_lens, dtype=np.float32)).astype(np.float32)
            stacked_weights = torch.from_numpy(np.repeat(bw, host_lens)).to(dev)
        else:
            stacked_weights = None   # only the training losses read it

        r_normal = config.first_subsampling_dl * config.KP_extent * 2.5
        layer_blocks = []
        input_points, input_neighbors, input_pools, input_upsamples, input_batches_len = [], [], [], [], []
        pending = []
        if caps is not None:
            # capacity mode (a replayed launch sequence): ONE persistent block of sticky status words per dataset object, zero when
            # the sequence starts; the caller that reads `static_status` clears it again (ops.pack_status at the end of the engine's
            # replay) -- no fill node per replay
            status_all = getattr(self, "_status_all", None)
            if status_all is None or status_all.device != dev:
                status_all = self._status_all = torch.zeros((64, 2), dtype=torch.int32, device=dev)
            elif not torch.cuda.is_current_stream_capturing():
                # eager use of the capacity mode (a tool, a test, the engine's warm-up): the flag words are sticky and nobody
                # is guaranteed to have cleared the previous call's -- start clean.  Inside a capture the sequence
                # relies on its own last node (ops.pack_status clear=...) instead: no fill node per replay
                status_all[:, 1].zero_()
        else:
            status_all = torch.zeros((64, 2), dtype=torch.int32, device=dev)   # one fill: the searches do not reset theirs
        arch = config.architecture
        cap = getattr(self, '_neighbor_cap', 192)
        grids = {}

        reqs = []      # searches not issued yet: (q, s, ql, sl, r, width, first_only, nn_hint, sink list, sink index)

        def request(q, s, ql, sl, r, layer, sink, first_only=False, nn_hint=0.0):
            """Ask for one index matrix; it lands in sink[-1] (a placeholder is appended now).  On the fast path the searches of
            one grid -- conv_i and pool_i of a level and the level above's up_i: same supports, same radius -- are issued
            together by flush(), after the level's subsampling (whose launch chain then overlaps with nothing it needs)."""
            lim = int(self.neighborhood_limits[layer])
            if first_only and not exact_shapes:
                lim = 1     # only the nearest support is computed AND stored: closest_pool reads column 0 (network_blocks.py:81);
                            # a full-width row of padding per point was 39 MB of writes per level-0 launch
            sink.append(None)
            if exact_shapes:
                sink[-1] = tf_batch_neighbors(q, s, ql, sl, r)[:, :lim]
                return
            reqs.append((q, s, ql, sl, float(r), lim, first_only, nn_hint, sink, len(sink) - 1))

        def flush():
            groups = {}
            for rq in reqs:
                groups.setdefault((rq[1].data_ptr(), rq[4]), []).append(rq)
            del reqs[:]
            for key, grp in groups.items():
                s_, sl_, r_ = grp[0][1], grp[0][3], grp[0][4]
                grid = grids.get(key)
                if grid is None:
                    grid = grids[key] = ops.NeighborGrid(s_, sl_, r_)
                    if getattr(s_, 'order', None) is None:
                        s_.order = grid.order   # cell-sorted visiting order of this level's points (ops._order)
                        s_.grid = grid          # ... and the grid itself: the visiting order of the searches these points QUERY
                # one launch per search: issuing the two or three searches of a grid as ONE launch (blockIdx.y = query set) was
                # measured 5-7 % slower end to end (profiles/r03_experiments.txt x8)
                for (q, _, ql, _, _, lim, fo, hint, sink, at) in grp:
                    out, status = grid.search(q, ql, lim, cap=cap, first_only=fo, status=status_all[len(pending)],
                                              reset_status=False, want_kmax=False, nn_hint=hint,
                                              query_grid=getattr(q, 'grid', None) if fo else None)
                    pending.append(status)
                    sink[at] = out

        empty_i = torch.zeros((0, 1), dtype=torch.int32, device=dev)
        for block_i, block in enumerate(arch):
            if 'global' in block or 'upsample' in block:
                break
            if not ('pool' in block or 'strided' in block):
                layer_blocks += [block]
                if block_i < len(arch) - 1 and not ('upsample' in arch[block_i + 1]):
                    continue
            layer = len(input_points)
            pooled = 'pool' in block or 'strided' in block
            if pooled:      # the next level's points first: pool_i's queries (the searches of this level go out together)
                dl = 2 * r_normal / (config.KP_extent * 2.5)
                if caps is not None:
                    hints = getattr(self, 'hints', None)
                    # one cloud of the stack holds at most its share of the level's capacity (the iteration-order rounds of
                    # the subsampler are launched for that size, not for the whole stack)
                    units = max(int(getattr(self, 'cap_units', 1)), 1)
                    pool_p, pool_b, _ = ops.batch_grid_subsample_async(stacked_points, stacked_lengths, dl, caps[layer + 1],
                                                                       status=status_all[len(pending)],
                                                                       m_hint=hints[layer + 1] if hints else 0,
                                                                       elem_cap=-(-caps[layer + 1] // units),
                                                                       elem_points=-(-caps[layer] // units))
                    pending.append(status_all[len(pending)])
                else:
                    pool_p, pool_b = tf_batch_subsampling(stacked_points, stacked_lengths, dl)
            if layer_blocks:
                if np.any(['deformable' in blck for blck in layer_blocks[:-1]]):
                    r = r_normal * config.density_parameter / (config.KP_extent * 2.5)
                else:
                    r = r_normal
                request(stacked_points, stacked_points, stacked_lengths, stacked_lengths, r, layer, input_neighbors)
            else:
                input_neighbors.append(empty_i)
            if pooled:
                if 'deformable' in block:
                    r = r_normal * config.density_parameter / (config.KP_extent * 2.5)
                else:
                    r = r_normal
                request(pool_p, stacked_points, pool_b, stacked_lengths, r, layer, input_pools)
                flush()      # conv_i, pool_i of this level + up_i of the level above: one grid, one launch
                # the supports of up_i are the voxel barycentres (edge dl) of the queries themselves: every query has one within
                # sqrt(3) dl -- a hint for the nearest-only search (8 cells instead of 27), never a constraint.  Its grid (the
                # next level's points, radius 2 r) is the next level's conv grid: issued with that level's searches
                request(stacked_points, pool_p, stacked_lengths, pool_b, 2 * r, layer, input_upsamples,
                        first_only=up_first_column_only, nn_hint=1.75 * dl)
            else:
                flush()
                input_pools.append(empty_i)
                pool_p = torch.zeros((0, 3), dtype=torch.float32, device=dev)
                pool_b = torch.zeros((0,), dtype=torch.int32, device=dev)
                input_upsamples.append(empty_i)
            input_points += [stacked_points]
            input_batches_len += [stacked_lengths]
            stacked_points = pool_p
            stacked_lengths = pool_b
            r_normal *= 2
            layer_blocks = []
        flush()

        overflow = False
        self.static_status = status_all[:len(pending)] if caps is not None else None
        self.level_lengths = input_batches_len      # per-level stack lengths (device int32[B]); not part of the flat list
        if pending and caps is None:
            # one read-back for all searches of the pyramid
            for kmax, flags in status_all[:len(pending)].tolist():
                if flags & _lib.ST_HIT_OVERFLOW and cap < _lib.NEIGHBOR_CAP:
                    overflow = True
                else:
                    ops._raise_flags(flags, 'tf_descriptor_input/neighbors')
        if overflow:
            # some query has more in-radius supports than the fast LDS budget: redo with the full budget (sticky)
            self._neighbor_cap = _lib.NEIGHBOR_CAP
            return self.tf_descriptor_input(config, first_points, stacked_features, first_lengths, batch_inds,
                                            exact_shapes=exact_shapes, up_first_column_only=up_first_column_only)

        # rows of pools[l] are the points of level l+1: give them that level's spatially coherent visiting order
        for l in range(len(input_pools) - 1):
            o = getattr(input_points[l + 1], 'order', None)
            if o is not None and input_pools[l].shape[0] > 0:
                input_pools[l].order = o

        if exact_shapes:
            stacked_batch_inds_0 = self.tf_stack_batch_inds(input_batches_len[0])
            stacked_batch_inds_1 = self.tf_stack_batch_inds(input_batches_len[-1])
        else:
            # the head kernel works from stack_lengths; the index matrices are only an output of the reference API
            # (a batched engine stack holds several reference stacks: ops.StackGroups carries their size to the head)
            stacked_batch_inds_0 = ops.StackGroups(getattr(self, 'stack_group', 0))
            stacked_batch_inds_1 = None
        li = input_points + input_neighbors + input_pools + input_upsamples
        li += [stacked_features, stacked_weights, stacked_batch_inds_0, stacked_batch_inds_1]
        return li

    def _descriptor_input_internal(self, config, stacked_points, stacked_features, stacked_lengths, caps):
        """The capacity-mode pyramid in the INTERNAL numbering (round 6): every level lives in the cell order of its own conv grid.
        Rows of neighbors[l] / pools[l] / upsamples[l] are the queries in their grid's cell order, the entries are positions in
        the supports' cell order, points[l] is the level's cloud in that order (views into the grid objects).  Which supports
        a row holds, and in which order, is exactly the reference's (datasets/common.py:1301-1413): renumbering back gives the
        reference-order matrices bit for bit (FragmentEngine.reference_order_flat, tests/test_gpu_internal_order.py).  The
        per-point kernels of the model then gather rows that are neighbours in MEMORY as well as in space; the subsamplings
        still read the reference-order clouds (the barycentres are fp32 sums in input order), and the one return to the
        reference's row order is the last kernel of the sequence (ops.pack_descriptors row_map = orders[0]).
        -> the flat list; self.orders[l] (device i32: internal row -> reference row) / self.level_lengths beside it."""
        dev = stacked_points.device
        lens_dev = ops.as_lens(stacked_lengths, dev)
        r_normal = config.first_subsampling_dl * config.KP_extent * 2.5
        status_all = getattr(self, "_status_all", None)
        if status_all is None or status_all.device != dev:
            status_all = self._status_all = torch.zeros((64, 2), dtype=torch.int32, device=dev)
        elif not torch.cuda.is_current_stream_capturing():
            status_all[:, 1].zero_()
        arch = config.architecture
        cap = getattr(self, '_neighbor_cap', 192)
        pending = []

        def status():
            st = status_all[len(pending)]
            pending.append(st)
            return st

        def search(grid, q, ql, qgrid, layer, first_only=False, nn_hint=0.0):
            lim = 1 if first_only else int(self.neighborhood_limits[layer])
            out, _ = grid.search(q, ql, lim, cap=cap, first_only=first_only, status=status(), reset_status=False, want_kmax=False,
                                 nn_hint=nn_hint, query_grid=qgrid, internal=True)
            return out
        empty_i = torch.zeros((0, 1), dtype=torch.int32, device=dev)
        points, lens_l, grids = [stacked_points], [lens_dev], []
        neighbors, pools, ups = [], [], []
        layer_blocks = []
        hints = getattr(self, 'hints', None)
        units = max(int(getattr(self, 'cap_units', 1)), 1)
        grids.append(ops.NeighborGrid(stacked_points, lens_dev, r_normal))
        for block_i, block in enumerate(arch):
            if 'global' in block or 'upsample' in block:
                break
            if 'deformable' in block:
                raise ValueError("internal numbering: deformable blocks are outside the inference path")
            if not ('pool' in block or 'strided' in block):
                layer_blocks += [block]
                if block_i < len(arch) - 1 and not ('upsample' in arch[block_i + 1]):
                    continue
            layer = len(neighbors)
            P, L, G = points[layer], lens_l[layer], grids[layer]
            pooled = 'pool' in block or 'strided' in block
            if pooled:      # the next level's points and its grid first: pool_i's queries bring their own cell order
                dl = 2 * r_normal / (config.KP_extent * 2.5)
                pool_p, pool_b, _ = ops.batch_grid_subsample_async(P, L, dl, caps[layer + 1], status=status(),
                                                                   m_hint=hints[layer + 1] if hints else 0,
                                                                   elem_cap=-(-caps[layer + 1] // units),
                                                                   elem_points=-(-caps[layer] // units))
                Gn = ops.NeighborGrid(pool_p, pool_b, 2 * r_normal)
            neighbors.append(search(G, P, L, G, layer) if layer_blocks else empty_i)
            if pooled:
                pools.append(search(G, pool_p, pool_b, Gn, layer))
                # the supports of up_i are the voxel barycentres (edge dl) of the queries themselves: a hint, never a constraint
                ups.append(search(Gn, P, L, G, layer, first_only=True, nn_hint=1.75 * dl))
                points.append(pool_p)
                lens_l.append(pool_b)
                grids.append(Gn)
            else:
                pools.append(empty_i)
                ups.append(empty_i)
            r_normal *= 2
            layer_blocks = []
        self.static_status = status_all[:len(pending)]
        self.level_lengths = lens_l
        self.orders = [g.order for g in grids]            # internal row -> reference row, per level
        self.level_points = points                        # the reference-order clouds (what the subsamplings read)
        self._grids = grids                               # (the index matrices and point views live in the grid objects)
        li = [g.xyz for g in grids] + neighbors + pools + ups
        li += [stacked_features, None, ops.StackGroups(getattr(self, 'stack_group', 0)), None]
        return li

    # ---- neighbour-limit calibration -----------------------------------------------------------------------------------
    def calibrate_neighbors(self, config, keep_ratio=0.8, samples_threshold=10000, verbose=False):
        """datasets/common.py:572-673: histogram the number of valid neighbours (index < rows) per layer over at
        most one epoch of the test split until every layer has >= samples_threshold samples; the limit of a layer
        is the number of histogram bins whose cumulative count is below keep_ratio of the total."""
        split = 'train' if len(getattr(self, 'anc_points', {}).get('train', [])) > 0 else 'test'
        gen_function, _, _ = self.get_batch_gen(split, config)
        map_func = self.get_tf_mapping(config)
        hist_n = int(np.ceil(4 / 3 * np.pi * (config.density_parameter + 1) ** 3))
        neighb_hists = np.zeros((config.num_layers, hist_n), dtype=np.int64)
        for sample in gen_function():
            if np.min(np.sum(neighb_hists, axis=1)) >= samples_threshold:
                break
            flat = map_func(*self._to_device(sample))
            neighbors = flat[config.num_layers:2 * config.num_layers]
            for l, mat in enumerate(neighbors):
                neighb_hists[l] += self.neighbor_histogram(mat, hist_n)
        cumsum = np.cumsum(neighb_hists.T, axis=0)
        percentiles = np.sum(cumsum < (keep_ratio * cumsum[hist_n - 1, :]), axis=0)
        self.neighborhood_limits = percentiles.astype(np.int32)
        if verbose:
            print('neighborhood_limits', self.neighborhood_limits)
        return neighb_hists

    @staticmethod
    def neighbor_histogram(neighb_mat, hist_n):
        """datasets/common.py:645-647 for one matrix.  The matrix is copied to the host for the bincount (it is a
        start-up calibration pass, not the hot path)."""
        mat = neighb_mat.cpu().numpy()
        counts = np.sum(mat < mat.shape[0], axis=1)
        return np.bincount(counts, minlength=hist_n)[:hist_n].astype(np.int64)

    # ---- test pipeline ------------------------------------------------------------------------------------------------------
    def _to_device(self, sample):
        """generator tuple -> map_func arguments: the two point arrays and stack_lengths become device tensors."""
        pts, anc_k, pos_k, obj, lens, ids, backup = sample
        return (torch.from_numpy(np.ascontiguousarray(pts, dtype=np.float32)).to(self.device), anc_k, pos_k, obj,
                torch.from_numpy(np.ascontiguousarray(lens, dtype=np.int32)).to(self.device), ids,
                torch.from_numpy(np.ascontiguousarray(backup, dtype=np.float32)).to(self.device))

    def init_test_input_pipeline(self, config):
        """datasets/common.py:776-857: set the limits to the upper bound, calibrate them, then expose the test
        split as an endless iterator of flat input lists (`self.flat_inputs`; call next() on it, or pass it to
        KernelPointFCNN which pulls one element per run -- the role of iter.get_next())."""
        config.num_classes = self.num_classes - len(self.ignored_labels)
        config.network_model = self.network_model
        hist_n = int(np.ceil(4 / 3 * np.pi * (config.density_parameter + 1) ** 3))
        self.neighborhood_limits = np.full(config.num_layers, hist_n, dtype=np.int32)
        self.calibrate_neighbors(config)
        print("self.neighborhood:", self.neighborhood_limits)
        gen_function, _, _ = self.get_batch_gen('test', config)
        map_func = self.get_tf_mapping(config)
        dataset = self

        class _Repeat:
            def __init__(self):
                self.it = None

            def reset(self):
                self.it = iter(gen_function())

            def __iter__(self):
                return self

            def __next__(self):
                if self.it is None:
                    self.reset()
                try:
                    sample = next(self.it)
                except StopIteration:  # .repeat()
                    self.reset()
                    sample = next(self.it)
                return map_func(*dataset._to_device(sample))

        self.flat_inputs = _Repeat()
        self.test_init_op = self.flat_inputs.reset


class FragmentDataset(Dataset):
    """Test-time dataset over a list of point clouds (numpy float32 [N,3], already at the first subsampling
    resolution).  Like the reference's 3DMatch / ETH / demo test generators (datasets/ThreeDMatch.py:180-192,
    demo_registration.py:30-95) every fragment is fed stacked with itself: anc == pos."""

    def __init__(self, clouds, ids=None, fast=False):
        Dataset.__init__(self, 'Fragments')
        self.anc_points = {'train': [], 'test': [np.ascontiguousarray(c, dtype=np.float32) for c in clouds]}
        self.ids_list = {'train': [], 'test': list(ids) if ids is not None else ['cloud_%d' % i for i in range(len(clouds))]}
        self.num_test = len(clouds)
        self.fast = fast
        self.caps = None      # per-level row capacities: capacity mode of tf_descriptor_input (see d3feat_amd.engine)
        self.hints = None     # per-level expected row counts (launch planning only)

    def get_batch_gen(self, split, config):
        def gen():
            for i, pts in enumerate(self.anc_points[split]):
                yield (np.concatenate([pts, pts], axis=0), np.array([], dtype=np.int32), np.array([], dtype=np.int32),
                       np.array([i, i], dtype=np.int32), np.array([pts.shape[0], pts.shape[0]], dtype=np.int32),
                       np.array([self.ids_list[split][i]] * 2), np.concatenate([pts, pts], axis=0))
        gen_types = ('float32', 'int32', 'int32', 'int32', 'int32', 'string', 'float32')
        gen_shapes = ([None, 3], [None], [None], [None], [None], [None], [None, 3])
        return gen, gen_types, gen_shapes

    def get_tf_mapping(self, config):
        def tf_map(anc_points, anc_keypts, pos_keypts, obj_inds, stack_lengths, ply_id, backup_points):
            batch_inds = None if self.fast else self.tf_get_batch_inds(stack_lengths)
            # stacked_features = ones([N, 1])  (demo_registration.py:102): a device-side fill
            if self.fast and self.caps is not None:
                # capacity mode: the all-ones column is a constant of the launch sequence -- filled once, not once per replay
                key = (int(anc_points.shape[0]), anc_points.device)
                if getattr(self, "_ones_key", None) != key:
                    self._ones_key, self._ones = key, torch.ones((key[0], 1), dtype=torch.float32, device=key[1])
                ones = self._ones
            else:
                ones = torch.ones((anc_points.shape[0], 1), dtype=torch.float32, device=anc_points.device)
            stacked_features = ops._tag(ones, anc_points)
            li = self.tf_descriptor_input(config, anc_points, stacked_features, stack_lengths, batch_inds,
                                          exact_shapes=not self.fast, up_first_column_only=self.fast, caps=self.caps)
            return li + [stack_lengths, anc_keypts, pos_keypts, ply_id, backup_points]
        return tf_map
