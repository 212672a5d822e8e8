"""The drop-in callers (SURVEY.md §8b last row, §8f row 2): the reference's inference scripts run UNCHANGED through
`python -m d3feat_amd.compat_run`, their `tensorflow` / `open3d` / `utils` / `datasets` / `models` imports resolving to the
compat/ tree.

  * CPU: every tensorflow / open3d symbol the reference's inference callers touch exists in the compat modules (the scripts
    are parsed where /root/reference is present -- this container -- and compared with the frozen list of SURVEY.md §8b
    everywhere); the compat tree shadows nothing else.
  * GPU: tests/compat_caller_demo.py (the same API calls as the reference demo, in the same order) runs end to end on
    synthetic clouds with a real TensorFlow checkpoint bundle on disk; its .npz outputs equal the direct engine path.
    The reference's own demo_registration.py and test_3dmatch.py are executed AS THEY ARE (byte for byte; sha256 logged) from
    /root/reference or from the scratch mirror .ref_scratch/ (tools/make_ref_scratch.py) and their outputs are compared with
    the direct engine path; stdout of both runs is kept under gpurun_out/ref_scripts/ (copied to profiles/ as evidence).
"""
import ast
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, write_tf_bundle

# the reference's caller files: the checkout itself (build container) or the scratch mirror staged by tools/make_ref_scratch.py
# right before a GPU visit (the GPU box has no /root/reference; .ref_scratch/ travels with gpurun's snapshot, is git-ignored)
REF = next((d for d in ("/root/reference", os.path.join(ROOT, ".ref_scratch"))
            if os.path.isfile(os.path.join(d, "demo_registration.py"))), "/root/reference")
EVIDENCE = os.path.join(ROOT, "gpurun_out", "ref_scripts")
COMPAT = os.path.join(ROOT, "compat")

# the exhaustive symbol list of the two drop-in scripts (+ utils/tester.py's ModelTester, evaluate.py's RANSAC call)
SURVEY_TF = ["float32", "int32", "string", "ones", "shape", "get_collection", "GraphKeys.GLOBAL_VARIABLES", "train.Saver",
             "ConfigProto", "Session", "global_variables_initializer"]
SURVEY_O3D = ["read_point_cloud", "voxel_down_sample", "PointCloud", "Vector3dVector", "registration.Feature",
              "registration_ransac_based_on_feature_matching", "TransformationEstimationPointToPoint",
              "CorrespondenceCheckerBasedOnEdgeLength", "CorrespondenceCheckerBasedOnDistance", "RANSACConvergenceCriteria",
              "estimate_normals", "draw_geometries", "geometry.create_mesh_sphere", "set_verbosity_level", "VerbosityLevel.Error",
              "utility.Vector3dVector"]


def _chains(path, roots, only_functions=None):
    """Dotted attribute chains `root.a.b` used in a source file (optionally only inside the named functions / classes)."""
    tree = ast.parse(open(path).read())
    nodes = [tree]
    if only_functions:
        nodes = [n for n in ast.walk(tree) if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in only_functions]
    out = set()
    for top in nodes:
        for n in ast.walk(top):
            if isinstance(n, ast.Attribute):
                parts, cur = [], n
                while isinstance(cur, ast.Attribute):
                    parts.append(cur.attr)
                    cur = cur.value
                if isinstance(cur, ast.Name) and cur.id in roots:
                    out.add((cur.id, ".".join(reversed(parts))))
    return out


def _resolves(mod, chain):
    obj = mod
    for part in chain.split("."):
        if not hasattr(obj, part):
            return False
        obj = getattr(obj, part)
    return True


def _import_compat(name):
    import importlib.util
    spec = importlib.util.spec_from_file_location("_compat_" + name, os.path.join(COMPAT, name, "__init__.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_compat_modules_expose_the_survey_symbol_list():
    tf, o3d = _import_compat("tensorflow"), _import_compat("open3d")
    assert [c for c in SURVEY_TF if not _resolves(tf, c)] == []
    assert [c for c in SURVEY_O3D if not _resolves(o3d, c)] == []


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")
def test_compat_modules_cover_what_the_reference_callers_touch():
    tf, o3d = _import_compat("tensorflow"), _import_compat("open3d")
    used = set()
    used |= _chains(os.path.join(REF, "demo_registration.py"), {"tf", "open3d"})
    used |= _chains(os.path.join(REF, "test_3dmatch.py"), {"tf", "open3d"})
    used |= _chains(os.path.join(REF, "utils", "tester.py"), {"tf", "open3d"}, only_functions={"__init__", "generate_descriptor"})
    used |= _chains(os.path.join(REF, "datasets", "ThreeDMatch.py"), {"tf", "open3d"},
                    only_functions={"prepare_geometry_registration", "get_tf_mapping"})
    used |= _chains(os.path.join(REF, "geometric_registration", "evaluate.py"), {"open3d"}, only_functions={"register2Fragments"})
    # a chain may continue into attributes of returned objects (tf.ConfigProto(...).gpu_options is a call result, not a
    # chain); only pure module-attribute chains are collected by _chains, so every one must resolve
    missing = sorted((r, c) for r, c in used if not _resolves(tf if r == "tf" else o3d, c))
    assert missing == [], missing
    assert ("tf", "train.Saver") in used and ("open3d", "registration_ransac_based_on_feature_matching") in used


def test_compat_tree_is_what_the_launcher_puts_first(tmp_path):
    """In a fresh interpreter compat_run.install_paths() makes the reference's module names resolve to compat/ -- and the
    reference's own directory is never on sys.path."""
    code = ("import sys, os; sys.path.insert(0, %r)\n"
            "from d3feat_amd import compat_run\n"
            "compat_run.install_paths()\n"
            "import tensorflow, open3d, utils.config, datasets.common, models.KPFCNN_model, kernels.convolution_ops\n"
            "mods = [tensorflow, open3d, utils.config, datasets.common, models.KPFCNN_model, kernels.convolution_ops]\n"
            "assert all(os.path.abspath(m.__file__).startswith(%r) for m in mods), [m.__file__ for m in mods]\n"
            "assert utils.config.Config is __import__('d3feat_amd.utils.config', fromlist=['Config']).Config\n"
            "print('ok')\n" % (ROOT, COMPAT))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(tmp_path), timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr


def test_session_and_saver_semantics_without_a_device(tmp_path):
    """Host half of the tensorflow stand-in: Saver.restore validates names / shapes against the index and refuses a checkpoint
    without tensor data unless told otherwise; Session.run resolves init ops and callables."""
    code = ("import sys, os, warnings; sys.path.insert(0, %r)\n"
            "from d3feat_amd import compat_run\n"
            "compat_run.install_paths()\n"
            "import tensorflow as tf\n"
            "calls = []\n"
            "sess = tf.Session(config=tf.ConfigProto(log_device_placement=False, allow_soft_placement=True))\n"
            "assert sess.run(tf.global_variables_initializer()) is None\n"
            "sess.run(lambda: calls.append(1)); assert calls == [1]\n"
            "cp = tf.ConfigProto(); cp.gpu_options.allow_growth = True\n"
            "saver = tf.train.Saver(tf.get_collection(tf.GraphKeys.GLOBAL_VARIABLES, scope='KernelPointNetwork'), max_to_keep=100)\n"
            "try:\n"
            "    saver.restore(sess, %r)\n"
            "    raise SystemExit('restore of a data-less checkpoint must fail')\n"
            "except FileNotFoundError as e:\n"
            "    assert 'D3FEAT_COMPAT_ALLOW_MISSING_CHECKPOINT' in str(e)\n"
            "print('ok')\n" % (ROOT, str(tmp_path / "snap-1")))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(tmp_path), timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr


# ---- GPU ------------------------------------------------------------------------------------------------------------
def _scratch_checkout(tmp_path, seed=3):
    """A working directory shaped like the reference checkout: demo_data/*.ply (two overlapping synthetic fragments), a log
    folder with parameters.txt and a REAL checkpoint bundle (random-init weights written in TensorFlow's on-disk format)."""
    from d3feat_amd.models.variables import build_variables
    from d3feat_amd.utils.config import threedmatch_config
    from d3feat_amd.utils.ply import write_ply
    from d3feat_amd.utils.synthetic import room_fragment
    root = tmp_path / "checkout"
    (root / "demo_data").mkdir(parents=True)
    snaps = root / "results" / "Log_contraloss" / "snapshots"
    snaps.mkdir(parents=True)
    raw = room_fragment(seed, n_raw=60000, edge=1.2)
    ang = 0.4
    R = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]], np.float32)
    a = raw[raw[:, 0] < 0.75]
    b = (raw[raw[:, 0] > 0.35] @ R.T + np.float32([0.3, -0.2, 0.1])).astype(np.float32)
    for name, pts in (("cloud_bin_0.ply", a), ("cloud_bin_1.ply", b)):
        assert write_ply(str(root / "demo_data" / name), [pts], ["x", "y", "z"])
    open(root / "results" / "Log_contraloss" / "parameters.txt", "w").write(open(os.path.join(GOLDEN, "parameters_3dmatch.txt")).read())
    W = build_variables(threedmatch_config(), seed=seed, randomize_bn=True).values
    write_tf_bundle(str(snaps / "snap-54"), {"KernelPointNetwork/" + k: v for k, v in W.items()}, crc=False)
    (snaps / "snap-54.meta").write_bytes(b"")
    return root, W, (a, b)


def _run_script(script, cwd, extra_env=None):
    env = dict(os.environ, PYTHONPATH=ROOT, **(extra_env or {}))
    return subprocess.run([sys.executable, "-m", "d3feat_amd.compat_run", script, "--cwd", str(cwd)], capture_output=True,
                          text=True, cwd=str(cwd), env=env, timeout=900)


def _check_npz_against_direct_path(root, W, clouds, device):
    """The .npz files a caller wrote == the same clouds through the library directly (stage-0 subsample, exact-shape
    pyramid, model), rows in ascending score order (demo_registration.py:158-170)."""
    import torch
    from d3feat_amd import tf_custom_ops as tfo
    from d3feat_amd.datasets.common import FragmentDataset
    from d3feat_amd.models.KPFCNN_model import KernelPointFCNN
    from d3feat_amd.utils.config import threedmatch_config
    cfg = threedmatch_config()
    limits = json.load(open(root / "registration.json"))["limits"] if (root / "registration.json").exists() else None
    subs = [tfo.grid_subsampling(torch.from_numpy(c).to(device), 0.03).cpu().numpy() for c in clouds]
    ds = FragmentDataset(subs, fast=False)
    ds.device = device
    ds.init_test_input_pipeline(cfg)
    if limits is not None:
        assert [int(x) for x in ds.neighborhood_limits] == limits
    model = KernelPointFCNN(ds.flat_inputs, cfg, weights=W, device=device)
    ds.test_init_op()
    for i, sub in enumerate(subs):
        d, s = model.run()
        d, s = d.cpu().numpy(), s.cpu().numpy()
        n = len(sub)
        got = np.load(root / "demo_data" / ("cloud_bin_%d.npz" % i))
        assert got["keypts"].shape == (n, 3) and got["features"].shape == (n, 32) and got["scores"].shape == (n, 1)
        order = np.argsort(s[:n], axis=0).squeeze()
        assert np.all(np.diff(got["scores"][:, 0]) >= 0)
        assert np.array_equal(got["keypts"], sub[order])
        assert np.array_equal(got["features"], d[:n][order]) and np.array_equal(got["scores"], s[:n][order])


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_demo_like_caller_through_compat(device, tmp_path):
    root, W, clouds = _scratch_checkout(tmp_path)
    r = _run_script(os.path.join(ROOT, "tests", "compat_caller_demo.py"), root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "Model restored" in r.stdout or os.path.exists(root / "demo_data" / "cloud_bin_0.npz")
    _check_npz_against_direct_path(root, W, clouds, device)
    reg = json.load(open(root / "registration.json"))
    assert 0.0 <= reg["fitness"] <= 1.0 and np.asarray(reg["transformation"]).shape == (4, 4)


def _keep_evidence(name, script, r, extra=""):
    import hashlib
    os.makedirs(EVIDENCE, exist_ok=True)
    with open(os.path.join(EVIDENCE, name + ".log"), "w") as f:
        f.write("script %s\nsha256 %s\nreturncode %d\n%s\n---- stdout ----\n%s\n---- stderr (tail) ----\n%s\n"
                % (script, hashlib.sha256(open(script, "rb").read()).hexdigest(), r.returncode, extra, r.stdout, r.stderr[-4000:]))


needs_ref_scripts = pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "demo_registration.py")),
                                       reason="neither /root/reference nor .ref_scratch/ (tools/make_ref_scratch.py) present")


@pytest.mark.gpu
@pytest.mark.timeout(900)
@needs_ref_scripts
def test_reference_demo_script_runs_unchanged(device, tmp_path):
    """The reference's demo_registration.py itself, byte for byte, in a scratch mirror of its checkout (inputs symlinked,
    outputs local), on the reference's own demo clouds (BASELINE configs[0]: 258 342 / 268 967 raw points).  The public checkout
    lacks the checkpoint's tensor data, so the initialised weights (reference initialisers, seed 42) are kept -- names / shapes
    are still validated against the real snap-54.index.  Checked: stage 0 of BOTH clouds == the reference's own
    grid_subsampling (golden fixtures), the .npz files == the direct engine path, the script reaches its last line."""
    from d3feat_amd import compat_run
    from d3feat_amd.models.variables import build_variables
    from d3feat_amd.utils.config import threedmatch_config
    from d3feat_amd.utils.ply import read_ply_xyz
    cwd = tmp_path / "mirror"
    compat_run._mirror(REF, str(cwd))
    script = os.path.join(REF, "demo_registration.py")
    r = _run_script(script, cwd, {"D3FEAT_COMPAT_ALLOW_MISSING_CHECKPOINT": "1"})
    _keep_evidence("demo_registration", script, r)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "Model restored from results/Log_contraloss/snapshots/snap-54" in r.stdout and "RegistrationResult" in r.stdout, r.stdout[-2000:]
    assert r.stdout.count("draw_geometries") == 4          # the script ran to its last statement (demo_registration.py:246-270)
    clouds = [read_ply_xyz(os.path.join(REF, "demo_data", "cloud_bin_%d.ply" % i)) for i in (0, 1)]
    for i in (0, 1):
        z = np.load(cwd / "demo_data" / ("cloud_bin_%d.npz" % i))
        want = np.load(os.path.join(GOLDEN, "demo_bin%d_sub003.npy" % i))
        assert z["features"].shape == (len(want), 32) and np.allclose(np.linalg.norm(z["features"], axis=1), 1.0, atol=1e-4)
        # same point SET as the reference's own subsampler produced (rows are in ascending-score order in the file)
        assert np.array_equal(np.sort(z["keypts"].view([("", np.float32)] * 3), axis=0), np.sort(want.view([("", np.float32)] * 3), axis=0))
    # the real 258 k / 269 k-point stage-0 calls, values AND row order (libstdc++ iteration order) against the reference's output
    import torch
    from d3feat_amd import tf_custom_ops as tfo
    for i in (0, 1):
        got = tfo.grid_subsampling(torch.from_numpy(clouds[i]).to(device), 0.03).cpu().numpy()
        want = np.load(os.path.join(GOLDEN, "demo_bin%d_sub003.npy" % i))
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "stage 0 of cloud_bin_%d" % i
    W = build_variables(threedmatch_config(), seed=42).values
    _check_npz_against_direct_path(cwd, W, clouds, device)


def _scene_checkout(tmp_path, seed=5):
    """A working directory shaped like the reference checkout for test_3dmatch.py: data/3DMatch/fragments/<8 scenes>/ with three
    synthetic fragments in two of them (fragment numbers out of lexical order: 2, 10 -> sorted by int, ThreeDMatch.py:346), a log
    folder with the reference's parameters.txt and a REAL checkpoint bundle, an older second log that must not be chosen."""
    from d3feat_amd.models.variables import build_variables
    from d3feat_amd.utils.config import threedmatch_config
    from d3feat_amd.utils.ply import write_ply
    from d3feat_amd.utils.synthetic import room_fragment
    scenes = ['7-scenes-redkitchen', 'sun3d-home_at-home_at_scan1_2013_jan_1', 'sun3d-home_md-home_md_scan9_2012_sep_30',
              'sun3d-hotel_uc-scan3', 'sun3d-hotel_umd-maryland_hotel1', 'sun3d-hotel_umd-maryland_hotel3',
              'sun3d-mit_76_studyroom-76-1studyroom2', 'sun3d-mit_lab_hj-lab_hj_tea_nov_2_2012_scan1_erika']
    root = tmp_path / "checkout3dm"
    frags = {}
    for sc in scenes:
        (root / "data" / "3DMatch" / "fragments" / sc).mkdir(parents=True)
    for k, (sc, num, edge) in enumerate(((scenes[0], 10, 1.1), (scenes[0], 2, 0.9), (scenes[3], 0, 1.0))):
        raw = room_fragment(seed + k, n_raw=50000, edge=edge)
        assert write_ply(str(root / "data" / "3DMatch" / "fragments" / sc / ("cloud_bin_%d.ply" % num)), [raw], ["x", "y", "z"])
        frags[(sc, num)] = raw
    (root / "data" / "3DMatch" / "fragments" / scenes[0] / "cloud_bin_2.info.txt").write_text("not a ply\n")
    params = open(os.path.join(GOLDEN, "parameters_3dmatch.txt")).read()
    W = build_variables(threedmatch_config(), seed=seed, randomize_bn=True).values
    for log in ("Log_contraloss",):
        snaps = root / "results" / log / "snapshots"
        snaps.mkdir(parents=True)
        (root / "results" / log / "parameters.txt").write_text(params)
        write_tf_bundle(str(snaps / "snap-54"), {"KernelPointNetwork/" + k: v for k, v in W.items()}, crc=False)
        (snaps / "snap-54.meta").write_bytes(b"")
    (root / "geometric_registration").mkdir()
    return root, W, frags, scenes


@pytest.mark.gpu
@pytest.mark.timeout(900)
@needs_ref_scripts
def test_reference_test_3dmatch_script_runs_unchanged(device, tmp_path):
    """The reference's test_3dmatch.py itself (test_3dmatch.py:22-94 -> ModelTester / ThreeDMatchDataset of the compat tree) on a
    synthetic 3-fragment scene folder with a real TensorFlow checkpoint bundle: the three .npy files per fragment of
    utils/tester.py:215-229 -- rows of the FIRST cloud, ascending score, the `[:-1]` quirk of `in_batches[0]` -- equal the
    direct library path (stage-0 subsample, calibration over the three self-pairs, exact-shape pyramid, model)."""
    import torch
    from d3feat_amd import tf_custom_ops as tfo
    from d3feat_amd.datasets.common import FragmentDataset
    from d3feat_amd.models.KPFCNN_model import KernelPointFCNN
    from d3feat_amd.utils.config import threedmatch_config
    root, W, frags, scenes = _scene_checkout(tmp_path)
    script = os.path.join(REF, "test_3dmatch.py")
    r = _run_script(script, root)
    _keep_evidence("test_3dmatch", script, r)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "Model restored from results/Log_contraloss/snapshots/snap-54" in r.stdout
    out = root / "geometric_registration" / "D3Feat_contralo-54-pred"      # utils/tester.py:163,173
    order = [(scenes[0], 2), (scenes[0], 10), (scenes[3], 0)]               # scene list order, fragments by number
    gen_lines = [l for l in r.stdout.splitlines() if l.startswith("Generate cloud_bin_")]
    assert gen_lines == ["Generate cloud_bin_%d for %s" % (n, sc) for sc, n in order], gen_lines
    cfg = threedmatch_config()
    subs = [tfo.grid_subsampling(torch.from_numpy(frags[k]).to(device), 0.03).cpu().numpy() for k in order]
    ds = FragmentDataset(subs, fast=False)
    ds.device = device
    ds.init_test_input_pipeline(cfg)
    model = KernelPointFCNN(ds.flat_inputs, cfg, weights=W, device=device)
    ds.test_init_op()
    for (sc, num), sub in zip(order, subs):
        d, s = (t.cpu().numpy() for t in model.run())
        n = len(sub)
        desc = np.load(out / "descriptors" / sc / ("cloud_bin_%d.D3Feat.npy" % num))
        kp = np.load(out / "keypoints" / sc / ("cloud_bin_%d.npy" % num))
        score = np.load(out / "scores" / sc / ("cloud_bin_%d.npy" % num))
        assert desc.dtype == kp.dtype == score.dtype == np.float32
        assert desc.shape == (n, 32) and kp.shape == (n, 3) and score.shape == (n, 1)
        o = np.argsort(s[:n, 0], kind="stable")
        assert np.all(np.diff(score[:, 0]) >= 0)
        assert np.array_equal(kp, sub[o]) and np.array_equal(desc, d[:n][o]) and np.array_equal(score, s[:n][o])
    files = sorted(str(p.relative_to(out)) for p in out.rglob("*.npy"))
    assert len(files) == 9, files
