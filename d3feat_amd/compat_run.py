"""Run one of the reference's inference scripts UNCHANGED on the MI355X path:

    python -m d3feat_amd.compat_run /path/to/D3Feat/demo_registration.py
    python -m d3feat_amd.compat_run /path/to/D3Feat/test_3dmatch.py

The script file is executed as it is (runpy, __name__ == '__main__').  What changes is what its imports resolve to: the
repository's compat/ tree is put first on sys.path, so that `tensorflow` / `open3d` (SURVEY.md §8b's symbol list) and the
reference's own package names `utils` / `datasets` / `models` / `kernels` are the d3feat_amd-backed modules -- the script's
directory is NOT on sys.path, the reference's TF-1 implementation is never imported.

Working directory: the scripts read and write paths relative to the checkout (demo_data/*.ply, results/Log_*/..., and they
WRITE demo_data/*.npz / geometric_registration/...).  By default a scratch directory is created that mirrors the checkout's
top level with symlinks (directories that receive outputs are real directories holding symlinks to the inputs), so a
read-only checkout stays untouched; --cwd DIR uses DIR as it is.
"""
import argparse
import os
import runpy
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMPAT = os.path.join(ROOT, "compat")


def _mirror(src_root, dst_root, writable=("demo_data", "geometric_registration", "geometric_registration_kitti")):
    os.makedirs(dst_root, exist_ok=True)
    for name in os.listdir(src_root):
        s, d = os.path.join(src_root, name), os.path.join(dst_root, name)
        if os.path.lexists(d):
            continue
        if name in writable and os.path.isdir(s):
            os.makedirs(d)
            for f in os.listdir(s):
                os.symlink(os.path.join(s, f), os.path.join(d, f))
        else:
            os.symlink(s, d)
    for name in writable:
        os.makedirs(os.path.join(dst_root, name), exist_ok=True)


def install_paths():
    """Put compat/ (and the repository) first on sys.path; matplotlib's stand-in only when the real one is missing."""
    for p in (ROOT, COMPAT):
        while p in sys.path:
            sys.path.remove(p)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, COMPAT)
    try:
        import matplotlib  # noqa: F401
    except ImportError:
        sys.path.append(os.path.join(COMPAT, "_optional"))
    for mod in ("tensorflow", "open3d", "utils", "datasets", "models", "kernels"):
        m = sys.modules.get(mod)
        if m is not None and not os.path.abspath(getattr(m, "__file__", "") or "").startswith(COMPAT):
            raise RuntimeError("module %r is already imported from %s: start from a fresh interpreter" % (mod, m.__file__))


def run(script, argv=(), cwd=None, allow_missing_checkpoint=False):
    script = os.path.abspath(script)
    if cwd is None:
        cwd = tempfile.mkdtemp(prefix="d3feat_compat_")
        _mirror(os.path.dirname(script), cwd)
    if allow_missing_checkpoint:
        os.environ["D3FEAT_COMPAT_ALLOW_MISSING_CHECKPOINT"] = "1"
    install_paths()
    old_cwd, old_argv = os.getcwd(), sys.argv
    os.chdir(cwd)
    sys.argv = [script] + list(argv)
    try:
        return runpy.run_path(script, run_name="__main__"), cwd
    finally:
        os.chdir(old_cwd)
        sys.argv = old_argv


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("script")
    ap.add_argument("--cwd", default=None, help="run in this directory instead of a scratch mirror of the script's checkout")
    ap.add_argument("--allow-missing-checkpoint", action="store_true",
                    help="tf.train.Saver.restore keeps the initial weights when snap-*.data-* is absent (public checkout)")
    a, rest = ap.parse_known_args()          # anything unknown goes to the script
    _, cwd = run(a.script, rest, a.cwd, a.allow_missing_checkpoint)
    print("[compat_run] working directory: %s" % cwd)


if __name__ == "__main__":
    main()
