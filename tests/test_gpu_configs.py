"""Parity cases for the BASELINE.json configurations other than the bench line (#2):
  #1 demo pair            -> tests/test_gpu_golden.py (golden vectors of the reference's demo cloud)
  #3 8-way sharding       -> tests/test_parallel_gloo.py
  #4 KITTI-like frames    -> here: two DIFFERENT frames stacked (datasets/KITTI.py:94-106), dl = 0.30, deep sparse lists;
                             and a large cloud (~100k points after subsampling) for the big-N code paths
  #5 batched fragments    -> here: 8 fragments in one stack (B = 8), fp32
Same bar as everywhere: indices / points bit-exact, descriptors / scores within 1e-4."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _oracle_inputs(coracle, cfg, clouds, limits):
    from oracle import network_np as onp
    pts = np.concatenate(clouds)
    lens = np.asarray([len(c) for c in clouds], np.int32)
    return onp.descriptor_input(cfg, pts, np.ones((len(pts), 1), np.float32), lens, limits,
                                lambda q, s, ql, sl, r: coracle.batch_neighbors(q, s, ql, sl, r),
                                lambda p, l, dl: coracle.batch_grid_subsampling(p, l, dl))


def _gpu_flat(cfg, clouds, limits, device, fast):
    """tf_descriptor_input for an arbitrary stack of clouds (the dataset tail of datasets/KITTI.py / ThreeDMatch.py)."""
    from d3feat_amd import ops
    from d3feat_amd.datasets.common import Dataset
    ds = Dataset("stack")
    ds.device = device
    ds.neighborhood_limits = np.asarray(limits, np.int32)
    pts = _t(np.concatenate(clouds), device)
    lens = ops.as_lens([len(c) for c in clouds], device)
    feats = torch.ones((pts.shape[0], 1), dtype=torch.float32, device=device)
    batch_inds = None if fast else ds.tf_get_batch_inds(lens)
    li = ds.tf_descriptor_input(cfg, pts, feats, lens, batch_inds, exact_shapes=not fast, up_first_column_only=fast)
    return li + [lens, None, None, ("a", "b"), pts]


def _check_pyramid(flat, want, limits, L, fast):
    for l in range(L):
        assert np.array_equal(flat[l].cpu().numpy().view(np.uint32), want["points"][l].view(np.uint32)), l
        for name, off in (("neighbors", L), ("pools", 2 * L), ("upsamples", 3 * L)):
            g, w = flat[off + l].cpu().numpy(), want[name][l]
            if w.shape[0] == 0:
                continue
            if not fast:
                assert np.array_equal(g, w), (name, l)
            elif name == "upsamples":
                assert np.array_equal(g[:, 0], w[:, 0]), (name, l)
            else:
                assert np.array_equal(g[:, :w.shape[1]], w), (name, l)


def test_config4_kitti_like_two_frames(device, coracle):
    from d3feat_amd.models.KPFCNN_model import KernelPointFCNN
    from d3feat_amd.models.variables import build_variables
    from d3feat_amd.utils.config import kitti_config
    from d3feat_amd.utils.synthetic import lidar_sweep
    from oracle import network_np as onp
    cfg = kitti_config()
    frames = [coracle.grid_subsampling(lidar_sweep(s, 120000), 0.3) for s in (3, 103)]   # two different frames
    assert len(frames[0]) != len(frames[1])
    limits = [25, 25, 25, 25, 25]
    want = _oracle_inputs(coracle, cfg, frames, limits)
    W = build_variables(cfg, seed=7, randomize_bn=True).values
    want_d, want_s = onp.forward(cfg, W, want)
    for fast in (False, True):
        flat = _gpu_flat(cfg, frames, limits, device, fast)
        _check_pyramid(flat, want, limits, cfg.num_layers, fast)
        model = KernelPointFCNN(flat, cfg, weights=W, device=device)
        d, s = model.out_features.cpu().numpy(), model.out_scores.cpu().numpy()
        assert np.abs(d - want_d).max() <= 1e-4
        assert np.abs(s - want_s).max() <= 1e-4 * max(1.0, np.abs(want_s).max())


def test_config4b_large_cloud(device, coracle):
    """~100k points after subsampling, stacked with itself (N0 ~ 200k): bit-exact pyramid against the oracle and the
    forward within 1e-4 (the reference's KITTI cell size, a room scaled 10x)."""
    from d3feat_amd.models.KPFCNN_model import KernelPointFCNN
    from d3feat_amd.models.variables import build_variables
    from d3feat_amd.utils.config import kitti_config
    from d3feat_amd.utils.synthetic import room_fragment
    from d3feat_amd import tf_custom_ops as tfo
    from oracle import network_np as onp
    cfg = kitti_config()
    raw = room_fragment(5, n_raw=1500000, edge=33.0, jitter=0.02)
    sub = coracle.grid_subsampling(raw, 0.3)
    got_sub = tfo.grid_subsampling(_t(raw, device), 0.3).cpu().numpy()
    assert np.array_equal(got_sub.view(np.uint32), sub.view(np.uint32))
    assert 60000 < len(sub) < 140000
    limits = [30, 30, 30, 30, 30]
    want = _oracle_inputs(coracle, cfg, [sub, sub], limits)
    flat = _gpu_flat(cfg, [sub, sub], limits, device, True)
    _check_pyramid(flat, want, limits, cfg.num_layers, True)
    W = build_variables(cfg, seed=9, randomize_bn=True).values
    want_d, want_s = onp.forward(cfg, W, want)
    model = KernelPointFCNN(flat, cfg, weights=W, device=device)
    d, s = model.out_features.cpu().numpy(), model.out_scores.cpu().numpy()
    assert np.abs(d - want_d).max() <= 1e-4
    assert np.abs(s - want_s).max() <= 1e-4 * max(1.0, np.abs(want_s).max())


def test_config5_eight_fragments_one_stack(device, coracle):
    """B = 8 clouds of different sizes in one stack.  The reference's detection head is written for two clouds
    (models/D3Feat.py:70-74), so descriptors are checked against the oracle for all eight and the per-cloud score
    normalisation is checked by comparing each cloud's scores with the same cloud run in a 2-stack."""
    from d3feat_amd.models.KPFCNN_model import KernelPointFCNN
    from d3feat_amd.models.variables import build_variables
    from d3feat_amd.utils.config import threedmatch_config
    from d3feat_amd.utils.synthetic import room_fragment
    from oracle import network_np as onp
    cfg = threedmatch_config()
    clouds = [coracle.grid_subsampling(room_fragment(60 + i, n_raw=12000 + 3000 * i, edge=0.8 + 0.05 * i), 0.03) for i in range(8)]
    limits = [37, 35, 36, 38, 38]
    want = _oracle_inputs(coracle, cfg, clouds, limits)
    flat = _gpu_flat(cfg, clouds, limits, device, True)
    _check_pyramid(flat, want, limits, cfg.num_layers, True)
    W = build_variables(cfg, seed=11, randomize_bn=True).values
    trace = {}
    with torch.no_grad():
        inp = dict(want)
        inp["points"] = [torch.as_tensor(p) for p in want["points"]]
        try:
            onp.assemble_FCNN_blocks(inp, cfg, W, trace)   # fills `trace`; its 2-cloud head result is not used
        except Exception:
            pass
    x = trace["uplayer_0/last_unary_1"]
    want_d = (x * torch.rsqrt(torch.clamp((x ** 2).sum(1, keepdim=True), min=1e-10))).numpy()
    model = KernelPointFCNN(flat, cfg, weights=W, device=device)
    d = model.out_features.cpu().numpy()
    s = model.out_scores.cpu().numpy()
    assert d.shape == want_d.shape and np.abs(d - want_d).max() <= 1e-4
    assert np.isfinite(s).all() and (s >= 0).all()
    # descriptors of a cloud do not depend on its stack mates
    lens = [len(c) for c in clouds]
    off = int(np.sum(lens[:3]))
    pair = _gpu_flat(cfg, [clouds[3], clouds[3]], limits, device, True)
    m2 = KernelPointFCNN(pair, cfg, weights=W, device=device)
    d2 = m2.out_features.cpu().numpy()[: lens[3]]
    s2 = m2.out_scores.cpu().numpy()[: lens[3]]
    assert np.abs(d2 - d[off: off + lens[3]]).max() <= 2e-5
    # ... and neither do its scores: the head normalises per cloud (D3Feat.py:84-90), for any number of clouds
    assert np.abs(s2 - s[off: off + lens[3]]).max() <= 2e-5 * max(1.0, np.abs(s2).max())


def test_engine_without_stage0_on_the_demo_pair(device, coracle):
    """BASELINE configs[0] as the reference's script feeds it (demo_registration.py:24,30-95): the clouds arrive ALREADY at
    0.03 m, the engine stacks each with itself without a stage-0 pass (FragmentEngine(stage0=False)); both demo clouds in one
    replay vs the oracle on the same clouds."""
    import os
    from conftest import GOLDEN
    from d3feat_amd.engine import FragmentEngine
    from d3feat_amd.models.variables import build_variables
    from d3feat_amd.utils.config import threedmatch_config
    from oracle import parity as par
    cfg = threedmatch_config()
    W = build_variables(cfg, seed=42, randomize_bn=True).values
    limits = np.asarray([37, 35, 36, 38, 38], np.int32)
    clouds = [np.load(os.path.join(GOLDEN, "demo_bin%d_sub003.npy" % i)) for i in (0, 1)]
    eng = FragmentEngine(cfg, W, limits, n0_cap=15360, level_ratio=0.32, slots=1, device=device, batch=2, stage0=False)
    eng.submit(0, [_t(c, device) for c in clouds])
    outs = eng.fetch(0, packed=True)
    assert eng.fallbacks == 0 and len(outs) == 2
    for c, rec in zip(clouds, outs):
        ref = par.fragment_reference(cfg, W, None, limits, co=coracle, clouds=[c, c])
        r = rec.cpu().numpy()
        cmp = par.compare_fragment(ref, r[:, :3], r[:, 3:35], r[:, 35:36])
        assert cmp["points_equal"] and cmp["desc_max_abs"] <= 1e-4 and cmp["score_max_abs"] <= 1e-4, cmp
    # a cloud beyond the capacity takes the eager path and still equals the oracle
    small = FragmentEngine(cfg, W, limits, n0_cap=8192, level_ratio=0.32, slots=1, device=device, batch=1, stage0=False)
    r = small.run(_t(clouds[0], device))
    assert small.fallbacks == 1
    ref = par.fragment_reference(cfg, W, None, limits, co=coracle, clouds=[clouds[0], clouds[0]])
    cmp = par.compare_fragment(ref, r[0].cpu().numpy(), r[1].cpu().numpy(), r[2].cpu().numpy())
    assert cmp["points_equal"] and cmp["desc_max_abs"] <= 1e-4 and cmp["score_max_abs"] <= 1e-4, cmp


@pytest.mark.timeout(900)
@pytest.mark.parametrize("flag,frames", [("--config4", 2), ("--demo", 1), ("--bf16-features", 1)])
def test_bench_lines_of_the_other_configurations(device, flag, frames):
    """bench.py --config4 (KITTI-like stacks of two different sweeps, real trained weights), --demo (the reference's demo
    pair) and --bf16-features (configs[4]: batched fragments, bf16 features + bf16 contraction): a full line each -- parity object against the oracle from the timed execution, cpu_baseline, latency, no fallback."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, "bench.py", flag, "--steps", "8", "--warmup", "2", "--windows", "3", "--cpu-fragments", "2",
                        "--no-cpu-1thread"], capture_output=True, text=True, cwd=ROOT, timeout=850,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert len(r.stdout.encode()) < 4096, len(r.stdout)           # stdout = ONE compact line (the driver keeps an 8 KB tail)
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    if flag == "--bf16-features":       # configs[4]: bf16 features + bf16 contraction, documented tolerance (tests/test_gpu_bf16.py)
        assert line["dtype"] == "bf16" and line["parity"]["tolerance"] == 1.5e-2
        assert line["parity"]["desc_max_abs"] <= 1.0e-2 and line["parity"]["score_max_abs"] <= 1.5e-2, line["parity"]
    else:
        assert line["dtype"] == "f32" and line["parity"]["tolerance"] == 1e-4
    assert line["parity"]["ok"] and line["parity"]["points_equal"] and line["parity"]["idx_equal"], line["parity"]
    assert line["config"]["engine_fallbacks"] == 0 and line["parity"]["engine_fallbacks"] == 0
    assert line["cpu_baseline"]["value"] > 0 and line["latency_ms"]["median"] > 0 and line["latency_ms"]["engine_fallbacks"] == 0
    assert len(line["timing"]["window_ms"]) == 3 and line["roofline"]["frac"] > 0
    assert abs(line["value"] - 8 * frames / (line["ms_per_step"] * 8 * 1e-3)) <= 1e-2 * line["value"]
