#!/usr/bin/env python3
"""Stage a scratch mirror of what the reference's two inference callers need, for a GPU-box visit.

`/root/reference` does not exist on the GPU box, `gpurun` ships the working tree: this script copies the CALLER FILES
(never the reference's implementation modules) into `.ref_scratch/` -- git-ignored, never committed -- so that
`tests/test_compat_scripts.py` can execute the reference's scripts byte for byte on an MI355X:

    .ref_scratch/demo_registration.py, test_3dmatch.py                       (the two scripts, unchanged)
    .ref_scratch/demo_data/cloud_bin_{0,1}.ply                               (config #1's clouds)
    .ref_scratch/results/Log_contraloss/{parameters.txt, snapshots/snap-54.{index,meta}}

    python tools/make_ref_scratch.py        # run in the build container right before `gpurun`

sha256 of every staged file is written to .ref_scratch/MANIFEST.json and compared with the reference by the tests'
evidence log (profiles/r03_ref_scripts_*.log).
"""
import hashlib
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, ".ref_scratch")
FILES = ["demo_registration.py", "test_3dmatch.py", "demo_data/cloud_bin_0.ply", "demo_data/cloud_bin_1.ply",
         "results/Log_contraloss/parameters.txt", "results/Log_contraloss/snapshots/snap-54.index",
         "results/Log_contraloss/snapshots/snap-54.meta"]


def main():
    if not os.path.isdir(REF):
        sys.exit("no %s here: run in the build container" % REF)
    if os.path.isdir(OUT):
        shutil.rmtree(OUT)
    man = {}
    for f in FILES:
        dst = os.path.join(OUT, f)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        shutil.copyfile(os.path.join(REF, f), dst)
        os.chmod(dst, 0o644)
        man[f] = hashlib.sha256(open(dst, "rb").read()).hexdigest()
    json.dump(man, open(os.path.join(OUT, "MANIFEST.json"), "w"), indent=1)
    print("staged %d files under %s" % (len(FILES), OUT))


if __name__ == "__main__":
    main()
