"""TEST INFRASTRUCTURE ONLY (oracle).  Never imported by d3feat_amd/.

Loader for tests/golden/network_*.npz -- what the reference's own Python (kernels/convolution_ops.py,
models/network_blocks.py, models/D3Feat.py, datasets/common.py, executed unmodified under oracle/tf_eager by
tools/make_golden_network.py) computed: inputs, block outputs (a row subset each), descriptors, scores -- in the
shapes oracle/network_np.py and the HIP model take.
"""
import json
import os

import numpy as np

from . import seeded_variables as sv

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


class GoldenNetwork:
    def __init__(self, name):
        """name: '3dmatch', 'kitti', or '3dmatch_4k' (a 4000-point crop: row sums of EVERY row of every block, two blocks whole)."""
        self.name = name
        self.z = np.load(os.path.join(GOLDEN, "network_%s.npz" % name))
        ext = ()
        if name == "kitti":
            ext = (np.load(os.path.join(GOLDEN, "kitti_epoch61_weights.npz")), np.load(os.path.join(GOLDEN, "kitti_kernel_points.npz")))
        self.W = sv.resolve(self.z, ext)
        z = self.z
        self.L = L = sum(1 for k in z.files if k.startswith("points_"))
        self.inputs = dict(points=[z["points_%d" % l] for l in range(L)], neighbors=[z["neighbors_%d" % l] for l in range(L)],
                           pools=[z["pools_%d" % l] for l in range(L)], upsamples=[z["upsamples_%d" % l] for l in range(L)],
                           features=z["features"], batch_weights=z["batch_weights"], in_batches=z["in_batches"],
                           out_batches=z["out_batches"], stack_lengths=z["stack_lengths"])
        self.limits = z["limits"]
        self.block_order = json.loads(str(z["block_order"]))
        self.descriptors, self.scores = z["descriptors"], z["scores"]

    def config(self):
        from d3feat_amd.utils.config import kitti_config, threedmatch_config
        return kitti_config() if self.name == "kitti" else threedmatch_config()

    def block(self, scope):
        """-> (rows, values[rows]) of the block's output as the reference computed it."""
        return self.z["rows/block/" + scope], self.z["block/" + scope]

    def rowsum(self, scope):
        """(the '3dmatch_4k' fixture) float64 [rows, 2]: per row of the block's output, (sum over channels, sum of magnitudes)."""
        return self.z["rowsum/" + scope]

    def whole_scopes(self):
        return [k[len("whole/"):] for k in self.z.files if k.startswith("whole/")]

    def whole(self, scope):
        return self.z["whole/" + scope]

    def kpconv(self, scope):
        return self.z["rows/kpconv/" + scope], self.z["kpconv/" + scope]

    def kpconv_scopes(self):
        return [k[len("kpconv/"):] for k in self.z.files if k.startswith("kpconv/")]

    def clouds(self):
        """The stage-0 clouds of the stack."""
        p0, out, o = self.inputs["points"][0], [], 0
        for n in self.inputs["stack_lengths"]:
            out.append(p0[o:o + int(n)])
            o += int(n)
        return out


def ops_fixture():
    return np.load(os.path.join(GOLDEN, "network_ops.npz"))
