"""The fragment engine (d3feat_amd/engine.py): whole fragment as one replayed HIP graph with device-resident sizes.
Its results must not depend on the capacities: indices / points bit-equal to the eager op-by-op path, descriptors and
scores equal up to fp32 summation order (the split-K plan of the contraction depends on the capacity), and within the
1e-4 bar of the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup(device):
    from d3feat_amd.models.variables import build_variables
    from d3feat_amd.utils.config import threedmatch_config
    cfg = threedmatch_config()
    W = build_variables(cfg, seed=42, randomize_bn=True).values
    limits = np.asarray([37, 35, 36, 38, 38], np.int32)
    return cfg, W, limits


def _frag(seed, n_raw=40000):
    from d3feat_amd.utils.synthetic import room_fragment
    return room_fragment(seed, n_raw=n_raw, edge=1.0)


def _close(a, b, tol):
    a, b = a.cpu().numpy().astype(np.float64), b.cpu().numpy().astype(np.float64)
    assert a.shape == b.shape
    assert np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max())


def test_engine_matches_eager_and_oracle(device, setup, coracle):
    from d3feat_amd.engine import FragmentEngine
    from oracle import network_np as onp
    cfg, W, limits = setup
    eng = FragmentEngine(cfg, W, limits, raw_cap=60000, n0_cap=14000, slots=2, device=device)
    raws = [torch.from_numpy(_frag(s, n)).to(device) for s, n in ((7, 40000), (8, 30000), (9, 52000))]
    # two fragments in flight, then a third reusing slot 0: the same graphs replayed for three different sizes
    eng.submit(0, raws[0])
    eng.submit(1, raws[1])
    outs = [tuple(t.clone() for t in eng.fetch(0)), tuple(t.clone() for t in eng.fetch(1))]
    outs.append(tuple(t.clone() for t in eng.run(raws[2], slot=0)))
    assert eng.fallbacks == 0
    for raw, (pts, d, s) in zip(raws, outs):
        ep, ed, es = eng.run_eager(raw)
        assert pts.shape == ep.shape and torch.equal(pts, ep)
        _close(d, ed, 2e-6)
        _close(s, es, 2e-6)
    # against the oracle (pyramid + network) for the first fragment
    raw = raws[0].cpu().numpy()
    sub = coracle.grid_subsampling(raw, 0.03)
    pts, d, s = outs[0]
    assert np.array_equal(pts.cpu().numpy()[: len(sub)].view(np.uint32), sub.view(np.uint32))
    inp = onp.descriptor_input(cfg, np.concatenate([sub, sub]), np.ones((2 * len(sub), 1), np.float32),
                               np.asarray([len(sub)] * 2, np.int32), limits,
                               lambda q, s_, ql, sl, r: coracle.batch_neighbors(q, s_, ql, sl, r),
                               lambda p, l, dl: coracle.batch_grid_subsampling(p, l, dl))
    want_d, want_s = onp.forward(cfg, W, inp)
    assert np.abs(d.cpu().numpy() - want_d).max() <= 1e-4
    assert np.abs(s.cpu().numpy() - want_s).max() <= 1e-4 * max(1.0, np.abs(want_s).max())


def test_engine_capacity_overflow_falls_back(device, setup):
    from d3feat_amd.engine import FragmentEngine
    cfg, W, limits = setup
    eng = FragmentEngine(cfg, W, limits, raw_cap=45000, n0_cap=3000, slots=1, device=device)
    raw = torch.from_numpy(_frag(11, 40000)).to(device)          # ~10k voxels > n0_cap
    pts, d, s = eng.run(raw)
    assert eng.fallbacks == 1
    ep, ed, es = eng.run_eager(raw)
    assert torch.equal(pts, ep) and torch.equal(d, ed) and torch.equal(s, es)
    big = torch.from_numpy(_frag(12, 50000)).to(device)          # more raw points than raw_cap
    pts, d, s = eng.run(big)
    assert eng.fallbacks == 2 and pts.shape[0] == d.shape[0] == s.shape[0]
    small = torch.from_numpy(_frag(13, 1000)).to(device)         # fits: the graph is still healthy after fallbacks
    pts, d, s = eng.run(small)
    assert eng.fallbacks == 2
    ep, ed, es = eng.run_eager(small)
    assert torch.equal(pts, ep)
    _close(d, ed, 2e-6)


def test_capacity_mode_ops_match_sync_ops(device, coracle):
    """The async subsample and the capacity-mode searches, op by op, against the oracle (bit-exact)."""
    from d3feat_amd import ops
    raw = _frag(21, 30000)
    n = len(raw)
    cap = 40000
    buf = torch.zeros((cap, 3), dtype=torch.float32, device=device)
    buf[:n] = torch.from_numpy(raw).to(device)
    lens = torch.tensor([n], dtype=torch.int32, device=device)
    sub, sub_l, st = ops.batch_grid_subsample_async(buf, lens, 0.03, 12000)
    want = coracle.grid_subsampling(raw, 0.03)
    m, flags = st.tolist()
    assert flags == 0 and m == len(want) and sub_l.tolist() == [m]
    assert np.array_equal(sub[:m].cpu().numpy().view(np.uint32), want.view(np.uint32))
    # too small an output buffer: flagged, nothing written out of bounds
    guard = torch.full((100 + 8, 3), 7.0, device=device)
    sub2, _, st2 = ops.batch_grid_subsample_async(buf, lens, 0.03, 100)
    assert st2.tolist()[1] & 16
    # per-cloud capacity of a two-cloud stack: the stack fits, one cloud does not -> flagged, empty result; with a
    # sufficient per-cloud capacity the stack matches the oracle cloud by cloud (bit-exact, order included)
    raw_b = _frag(22, 9000)
    nb_ = len(raw_b)
    buf2 = torch.zeros((cap + 12000, 3), dtype=torch.float32, device=device)
    buf2[:n] = torch.from_numpy(raw).to(device)
    buf2[n:n + nb_] = torch.from_numpy(raw_b).to(device)
    lens2 = torch.tensor([n, nb_], dtype=torch.int32, device=device)
    want_b = coracle.grid_subsampling(raw_b, 0.03)
    _, sl3, st3 = ops.batch_grid_subsample_async(buf2, lens2, 0.03, 24000, elem_cap=len(want) - 1)
    assert st3.tolist()[1] & 16 and st3.tolist()[0] == 0 and sl3.tolist() == [0, 0]
    sub4, sl4, st4 = ops.batch_grid_subsample_async(buf2, lens2, 0.03, 24000, elem_cap=len(want))
    assert st4.tolist() == [len(want) + len(want_b), 0] and sl4.tolist() == [len(want), len(want_b)]
    assert np.array_equal(sub4[:len(want) + len(want_b)].cpu().numpy().view(np.uint32),
                          np.concatenate([want, want_b]).view(np.uint32))
    pair, plens = ops.stack_self_pair(sub)
    assert plens.tolist() == [m, m] and int(pair.n_dev.item()) == 2 * m
    assert torch.equal(pair[:m], sub[:m]) and torch.equal(pair[m:2 * m], sub[:m])
    grid = ops.NeighborGrid(pair, plens, 0.075)
    out, status = grid.search(pair, plens, 37)
    L = np.asarray([m, m], np.int32)
    wn = coracle.batch_neighbors(np.concatenate([want, want]), np.concatenate([want, want]), L, L, np.float32(0.075))
    got = out[: 2 * m].cpu().numpy()
    k = min(37, wn.shape[1])
    assert np.array_equal(got[:, :k], wn[:, :k])
    assert status.tolist()[0] == wn.shape[1]


def test_engine_mirror_mode_equals_stacked_pair(device, setup):
    """mirror_self_pair=True computes the cloud once and mirrors it into the stacked layout: same points bit for bit, same
    descriptors / scores up to fp32 summation order (the K split of the contractions depends on the row count)."""
    from d3feat_amd.engine import FragmentEngine
    cfg, W, limits = setup
    eng = FragmentEngine(cfg, W, limits, raw_cap=60000, n0_cap=14000, slots=2, device=device, mirror_self_pair=True)
    for seed, n in ((31, 40000), (32, 25000)):
        raw = torch.from_numpy(_frag(seed, n)).to(device)
        pts, d, s = eng.run(raw, slot=seed % 2)
        ep, ed, es = eng.run_eager(raw)
        assert eng.fallbacks == 0
        assert pts.shape == ep.shape and torch.equal(pts, ep)
        _close(d, ed, 5e-6)
        _close(s, es, 5e-6)
        m = pts.shape[0] // 2
        assert torch.equal(d[:m], d[m:]) and torch.equal(s[:m], s[m:])


def test_engine_two_different_clouds_kitti_like(device, coracle):
    """two_clouds=True: the KITTI test generator's stack of two DIFFERENT frames (datasets/KITTI.py:94-106), through the
    graph engine; against the eager path and the oracle pyramid + network."""
    from d3feat_amd.engine import FragmentEngine
    from d3feat_amd.models.variables import build_variables
    from d3feat_amd.utils.config import kitti_config
    from d3feat_amd.utils.synthetic import lidar_sweep
    from oracle import network_np as onp
    cfg = kitti_config()
    W = build_variables(cfg, seed=7, randomize_bn=True).values
    limits = np.asarray([25, 25, 25, 25, 25], np.int32)
    raws = [lidar_sweep(s, 120000) for s in (3, 103)]
    eng = FragmentEngine(cfg, W, limits, raw_cap=250000, n0_cap=30000, level_ratio=0.6, slots=1, device=device, two_clouds=True)
    pts, d, s = eng.run(tuple(torch.from_numpy(r).to(device) for r in raws))
    assert eng.fallbacks == 0
    subs = [coracle.grid_subsampling(r, 0.3) for r in raws]
    assert np.array_equal(pts.cpu().numpy().view(np.uint32), np.concatenate(subs).view(np.uint32))
    inp = onp.descriptor_input(cfg, np.concatenate(subs), np.ones((len(pts), 1), np.float32),
                               np.asarray([len(x) for x in subs], np.int32), limits,
                               lambda q, s_, ql, sl, r: coracle.batch_neighbors(q, s_, ql, sl, r),
                               lambda p, l, dl: coracle.batch_grid_subsampling(p, l, dl))
    want_d, want_s = onp.forward(cfg, W, inp)
    assert np.abs(d.cpu().numpy() - want_d).max() <= 1e-4
    assert np.abs(s.cpu().numpy() - want_s).max() <= 1e-4 * max(1.0, np.abs(want_s).max())


def test_engine_four_slots_stay_correct_under_load(device, setup):
    """Four graphs in flight for many iterations (a replayed graph with runtime memset / memcpy nodes hung or faulted under
    exactly this load; the library launches fill / copy kernels instead)."""
    from d3feat_amd.engine import FragmentEngine
    cfg, W, limits = setup
    eng = FragmentEngine(cfg, W, limits, raw_cap=45000, n0_cap=14000, slots=4, device=device)
    raws = [torch.from_numpy(_frag(70 + i, n)).to(device) for i, n in enumerate((30000, 40000, 25000, 35000, 20000))]
    refs = [tuple(t.clone() for t in eng.run_eager(r)) for r in raws]
    inflight, bad = [None] * 4, 0

    def check(i, out):
        p, d, s = out
        rp, rd, rs = refs[i]
        return p.shape == rp.shape and torch.equal(p, rp) and (d - rd).abs().max().item() < 5e-6 and \
            (s - rs).abs().max().item() < 5e-6
    for it in range(40):
        k = it % 4
        if inflight[k] is not None:
            bad += 0 if check(inflight[k], eng.fetch(k)) else 1
        eng.submit(k, raws[it % len(raws)])
        inflight[k] = it % len(raws)
    for k in range(4):
        bad += 0 if check(inflight[k], eng.fetch(k)) else 1
    assert bad == 0 and eng.fallbacks == 0


@pytest.mark.parametrize("mirror", [False, True])
def test_engine_batched_fragments(device, setup, mirror):
    """batch=3: three fragments stacked into one replay (and a partial batch of two): every fragment's result equals its own
    eager run -- per-cloud searches / subsampling / head normalisation make stack mates invisible to each other."""
    from d3feat_amd.engine import FragmentEngine
    cfg, W, limits = setup
    eng = FragmentEngine(cfg, W, limits, raw_cap=45000, n0_cap=14000, slots=2, device=device, batch=3, mirror_self_pair=mirror)
    raws = [torch.from_numpy(_frag(80 + i, n)).to(device) for i, n in enumerate((30000, 40000, 25000, 35000, 20000))]
    refs = [tuple(t.clone() for t in eng.run_eager(r)) for r in raws]
    eng.submit(0, raws[:3])
    eng.submit(1, raws[3:])                      # partial batch
    outs = [tuple(t.clone() for t in o) for o in eng.fetch(0)] + [tuple(t.clone() for t in o) for o in eng.fetch(1)]
    assert len(outs) == 5 and eng.fallbacks == 0
    for (p, d, s), (rp, rd, rs) in zip(outs, refs):
        assert p.shape == rp.shape and torch.equal(p, rp)
        _close(d, rd, 5e-6)
        _close(s, rs, 5e-6)
    # single-fragment call on a batched engine keeps the single-tuple API
    p, d, s = eng.run(raws[1])
    assert torch.equal(p, refs[1][0])


def test_engine_flagged_batch_isolates_the_outlier(device, setup):
    """A replay of four fragments of which ONE exceeds the per-fragment voxel capacity, and one with more raw points than
    raw_cap: the flags are per stacked call, so the engine replays the fragments one by one and only the outliers take the
    eager path -- fallbacks count 1 + 1, not 4 + 4; every result equals the fragment's own eager run."""
    from d3feat_amd.engine import FragmentEngine
    cfg, W, limits = setup
    eng = FragmentEngine(cfg, W, limits, raw_cap=45000, n0_cap=9000, slots=1, device=device, batch=4)
    sizes = (6000, 40000, 4000, 8000)                        # fragment 1: ~10 k voxels > n0_cap, the others 3-6 k
    raws = [torch.from_numpy(_frag(90 + i, n)).to(device) for i, n in enumerate(sizes)]
    refs = [tuple(t.clone() for t in eng.run_eager(r)) for r in raws]
    assert refs[1][0].shape[0] // 2 > 9000 and max(refs[i][0].shape[0] // 2 for i in (0, 2, 3)) < 9000
    outs = eng.run(raws)
    assert eng.isolated == 1 and eng.fallbacks == 1 and eng.fragments == 4
    for (p, d, s), (rp, rd, rs) in zip(outs, refs):
        assert torch.equal(p, rp)
        _close(d, rd, 5e-6)
        _close(s, rs, 5e-6)
    big = torch.from_numpy(_frag(95, 50000)).to(device)      # more raw points than raw_cap: never enters a replay
    outs = eng.run([raws[0], big, raws[2]])
    assert eng.isolated == 2 and eng.fallbacks == 2 and eng.fragments == 7
    assert torch.equal(outs[0][0], refs[0][0]) and torch.equal(outs[2][0], refs[2][0])
    packed = eng.run([raws[3], raws[2]])                     # and the graph is healthy afterwards
    assert eng.fallbacks == 2 and torch.equal(packed[0][0], refs[3][0])


def test_engine_isolates_a_flagged_batch_of_in_place_producers(device, setup):
    """submit() accepts fragments that a producer wrote straight into the slot's raw buffer (no copy: the data_ptr shortcut).  When
    such a replay is flagged, the isolation re-submits the fragments one by one THROUGH THE SAME BUFFER: fragment 0's re-run
    would overwrite fragments 1.. before they are re-submitted -- the engine clones aliased sources first."""
    from d3feat_amd.engine import FragmentEngine
    cfg, W, limits = setup
    eng = FragmentEngine(cfg, W, limits, raw_cap=45000, n0_cap=9000, slots=1, device=device, batch=3)
    sizes = (6000, 40000, 8000)                              # fragment 1 exceeds the voxel capacity: the replay is flagged
    hosts = [_frag(190 + i, n) for i, n in enumerate(sizes)]
    refs = [tuple(t.clone() for t in eng.run_eager(torch.from_numpy(h).to(device))) for h in hosts]
    raw = eng.slots[0].raw
    views, o = [], 0
    for h in hosts:                                          # the producer: decode / copy straight into the slot's buffer
        raw[o:o + len(h)].copy_(torch.from_numpy(h).to(device))
        views.append(raw[o:o + len(h)])
        o += len(h)
    outs = eng.run(views)
    assert eng.isolated == 1 and eng.fallbacks == 1
    for (p, d, s), (rp, rd, rs) in zip(outs, refs):
        assert torch.equal(p, rp)
        _close(d, rd, 5e-6)
        _close(s, rs, 5e-6)


def test_packed_weight_copies_outlive_any_cache_traffic(device):
    """A captured graph reads the PACKED copy of its weights (transposed fp32 for the LDS-DMA contraction, bf16 for configs[4]) by
    raw pointer.  The copies therefore ride on the weight tensor itself: no amount of other models' weights going through the
    library afterwards may free or recycle them (round 4: a global 512-entry cache did, and an engine replayed with another
    model's weights)."""
    from d3feat_amd import ops
    g = torch.Generator(device="cpu").manual_seed(3)
    A = torch.randn((3000, 128), generator=g).to(device)
    W = (torch.randn((128, 64), generator=g) * 0.1).to(device)
    want = (A.double() @ W.double()).float()
    ops.gemm(A, W)                                         # eager warm-up: packs W outside the capture
    stream = torch.cuda.Stream(device=device)
    graph = torch.cuda.CUDAGraph()
    with ops.private_workspace() as pw:
        with torch.cuda.graph(graph, stream=stream):
            out = ops.gemm(A, W)
    keep = pw.kept
    for i in range(700):                                   # 700 other weight tensors come and go
        Wi = torch.full((128, 64), float(i), device=device)
        ops.gemm(A[:64], Wi)
        del Wi
    torch.cuda.synchronize(device)
    graph.replay()
    torch.cuda.synchronize(device)
    assert (out - want).abs().max().item() <= 1e-3
    # a view of a bigger weight tensor (K_values.reshape in kernels/convolution_ops.py) gets its own copy on the base tensor
    K_values = (torch.randn((15, 8, 16), generator=g) * 0.1).to(device)
    A2 = torch.randn((500, 120), generator=g).to(device)
    o1 = ops.gemm(A2, K_values.reshape(120, 16))
    o2 = ops.gemm(A2, K_values.reshape(120, 16))
    assert torch.equal(o1, o2) and hasattr(K_values, "_d3f_f32t") and len(K_values._d3f_f32t) == 1
    assert (o1.double() - A2.double() @ K_values.reshape(120, 16).double()).abs().max().item() <= 1e-3
    del keep


def test_pack_status_packs_in_one_launch_and_clears_the_flag_words_only(device, setup):
    """d3f_pack_status (round 5): the replay's last node packs [n_total | status0 | statuses | lens] into one block and zeroes the
    sticky FLAG column of the searches' status words -- the size / kmax column stays readable after the replay -- and a flagged
    replay does not poison the next one (the flags of replay n are gone when replay n + 1 reads its own)."""
    from d3feat_amd import ops
    from d3feat_amd.engine import FragmentEngine
    a = torch.tensor([7], dtype=torch.int32, device=device)
    b = torch.tensor([1, 2], dtype=torch.int32, device=device)
    st = torch.tensor([[10, 4], [20, 0], [30, 8]], dtype=torch.int32, device=device)
    lens = torch.tensor([5, 6, 7, 8], dtype=torch.int32, device=device)
    dst = torch.full((16,), -1, dtype=torch.int32, device=device)
    ops.pack_status(dst, [a, b, st, lens], clear=st)
    torch.cuda.synchronize()
    assert dst.tolist() == [7, 1, 2, 10, 4, 20, 0, 30, 8, 5, 6, 7, 8, -1, -1, -1]
    assert st.tolist() == [[10, 0], [20, 0], [30, 0]]
    # engine level: a replay whose capacity is exceeded raises flags (fallback), the next replay of the same slot with a small cloud
    # must come back clean, from the graph
    cfg, W, limits = setup
    big, small = _frag(3, n_raw=40000), _frag(4, n_raw=6000)
    eng = FragmentEngine(cfg, W, limits, raw_cap=50000, n0_cap=8192, slots=1, device=device)
    eng.run(torch.from_numpy(big).to(device))
    assert eng.fallbacks == 1
    eng.run(torch.from_numpy(small).to(device))
    assert eng.fallbacks == 1 and eng.fragments == 2
    assert int(eng.slots[0].status[:, 1].abs().sum().item()) == 0


def test_in_place_weight_update_keeps_the_packed_copy_at_its_address(device):
    """an in-place update of a weight tensor re-packs INTO THE SAME BUFFER -- a captured graph that holds the
    packed copy's address then computes with the new weights instead of reading freed memory."""
    from d3feat_amd import ops
    g = torch.Generator(device="cpu").manual_seed(5)
    A = torch.randn((3000, 128), generator=g).to(device)
    W = (torch.randn((128, 64), generator=g) * 0.1).to(device)
    ops.gemm(A, W)                                         # eager warm-up: packs W outside the capture
    slots = [s for s in ("_d3f_x3", "_d3f_f32t", "_d3f_bf16t") if hasattr(W, s)]
    assert slots
    before = {s: [h[1].data_ptr() for h in getattr(W, s).values()] for s in slots}
    stream = torch.cuda.Stream(device=device)
    graph = torch.cuda.CUDAGraph()
    with ops.private_workspace() as pw:
        with torch.cuda.graph(graph, stream=stream):
            out = ops.gemm(A, W)
    keep = pw.kept
    W.mul_(-0.5).add_(0.01)                                # in place: same storage, new version
    want = (A.double() @ W.double()).float()
    assert ops.refresh_packed_weights([W]) == len(slots)
    after = {s: [h[1].data_ptr() for h in getattr(W, s).values()] for s in slots}
    assert before == after
    torch.cuda.synchronize(device)
    graph.replay()
    torch.cuda.synchronize(device)
    assert (out - want).abs().max().item() <= 1e-3
    # an eager call after a further update takes the same route
    W.add_(0.02)
    o2 = ops.gemm(A, W)
    assert (o2.double() - A.double() @ W.double()).abs().max().item() <= 1e-3
    assert {s: [h[1].data_ptr() for h in getattr(W, s).values()] for s in slots} == before
    del keep
    # a strided status block must be refused, not silently copied (the graph would capture a temporary)
    st = torch.zeros((4, 4), dtype=torch.int32, device=device)
    dst = torch.zeros((16,), dtype=torch.int32, device=device)
    with pytest.raises(AssertionError):
        ops.pack_status(dst, [st[:, :2]])


def test_in_place_weight_update_reaches_a_captured_engine(device, setup):
    """FragmentEngine.refresh_weights: new weight values under live graphs -- every device tensor and packed copy is rewritten in
    place, the next replay equals an engine built from the new weights (same capacities: same launch plans, bit for bit)."""
    from d3feat_amd.engine import FragmentEngine
    cfg, W, limits = setup
    raw = torch.from_numpy(_frag(11, 30000)).to(device)
    eng = FragmentEngine(cfg, W, limits, raw_cap=40000, n0_cap=10000, slots=1, device=device)
    p0, d0, s0 = (t.clone() for t in eng.run(raw))
    rng = np.random.default_rng(5)
    W2 = {k: (v * (1.0 + 0.1 * rng.standard_normal(v.shape))).astype(np.float32) if not k.endswith("kernel_points") else v
          for k, v in W.items()}
    n = eng.refresh_weights(W2)
    assert n > 0
    p1, d1, s1 = (t.clone() for t in eng.run(raw))
    assert eng.fallbacks == 0
    ref = FragmentEngine(cfg, W2, limits, raw_cap=40000, n0_cap=10000, slots=1, device=device)
    p2, d2, s2 = ref.run(raw)
    assert torch.equal(p1, p2) and torch.equal(d1, d2) and torch.equal(s1, s2)
    assert (d1 - d0).abs().max().item() > 1e-3            # the update was really seen
    with pytest.raises(ValueError):
        k = next(k for k in W if k.endswith("kernel_points"))
        eng.refresh_weights({k: W[k] + 1.0})


def test_records_written_in_place_equal_the_copied_ones(device, setup):
    """FragmentEngine.submit(out=...): the replay's last kernel writes the [xyz | desc | score] records of a fragment's KEPT cloud
    (the first of the stacked self-pair, utils/tester.py:208-229) straight into the caller's buffer (d3f_pack_descriptors_to) and
    the raw clouds are read in place (d3f_batch_grid_subsample_async_inplace): bit-identical to the slot's own record block, for
    one fragment and for a batch, also when a fragment of the batch takes the eager fallback."""
    from d3feat_amd import ops
    from d3feat_amd.engine import FragmentEngine
    from d3feat_amd.parallel import ShardCollector
    cfg, W, limits = setup
    raws = [torch.from_numpy(_frag(s, n)).to(device) for s, n in ((21, 30000), (22, 26000), (23, 34000))]
    eng = FragmentEngine(cfg, W, limits, raw_cap=40000, n0_cap=10000, slots=2, device=device, batch=3)
    eng.submit(0, raws)
    want = [r.clone() for r in eng.fetch(0, packed=True)]
    col = ShardCollector(rows_cap=8 * 10240, width=36, device=device, chunk_frags=4, frag_rows=10240)
    dst = col.slots(3)
    assert dst is not None and len(dst) == 3
    eng.submit(1, raws, out=dst)
    got = eng.fetch(1, packed=True)
    for w, g, d in zip(want, got, dst):
        half = w.shape[0] // 2
        assert g.shape[0] == half and g.data_ptr() == d.data_ptr()
        assert torch.equal(g, w[:half]) and torch.equal(w[:half, :3], w[half:, :3])
        col.add(g)
    assert col.frag_rows == [int(w.shape[0] // 2) for w in want]
    parts, rows = col.gather(compact=False)[0]
    assert all(torch.equal(p, w[: w.shape[0] // 2]) for p, w in zip(parts, want))
    # the raw clouds were read where they are: no copy into the slot's staging buffer happened for device tensors
    assert int(eng.slots[1].host_ptrs[0]) == raws[0].data_ptr()
    # a batch with an oversize fragment: that one comes from the eager path (kept half only, a fresh tensor), the others in place
    big = torch.from_numpy(_frag(24, 60000)).to(device)
    col.reset()
    dst = col.slots(2)
    eng.submit(0, [raws[0], big], out=dst)
    got = eng.fetch(0, packed=True)
    assert eng.fallbacks >= 1 and torch.equal(got[0], want[0][: want[0].shape[0] // 2])
    ep, ed, es = eng.run_eager(big)
    rec = ops.pack_descriptors(ep, ed, es)
    assert torch.equal(got[1], rec[: rec.shape[0] // 2])
