"""KernelPointFCNN: the model class of the reference (models/KPFCNN_model.py:49-203), inference part.

    model = KernelPointFCNN(flat_inputs, config)          # same constructor signature
    model.anchor_inputs / out_features / out_scores / anc_id / pos_id / dropout_prob

`flat_inputs` is the positional list built by datasets/common.py:1410-1413 + the dataset tail
(datasets/ThreeDMatch.py:322; unpacked at KPFCNN_model.py:86-121), either as a list of tensors (the network is
evaluated immediately) or an iterator of such lists (dataset.flat_inputs; `run()` pulls the next element, which is
what one sess.run does in the reference).  Weights: a dict keyed by the checkpoint variable names (without the
`KernelPointNetwork/` root); missing variables are created like the reference's initialisers.  The loss graph
(:143-191) is training-only and not part of this package.
"""
import numpy as np
import torch

from .D3Feat import assemble_FCNN_blocks
from .network_blocks import use_variables
from .variables import VariableStore


class KernelPointFCNN:
    def __init__(self, flat_inputs, config, weights=None, seed=42, device=None):
        self.config = config
        if device is None:
            device = torch.device('cuda', torch.cuda.current_device())
        self.device = device
        self.variables = VariableStore(weights, seed=seed, device=device)
        self.dropout_prob = 1.0          # inference (a tf.placeholder fed with 1.0 in the reference)
        self.flat_inputs = flat_inputs
        self.anchor_inputs = None
        self.out_features = self.out_scores = None
        self.anc_id = self.pos_id = None
        if isinstance(flat_inputs, (list, tuple)):
            self.run(flat_inputs)

    def unpack(self, flat_inputs):
        """KPFCNN_model.py:86-121."""
        L = self.config.num_layers
        a = dict()
        a['points'] = flat_inputs[:L]
        a['neighbors'] = flat_inputs[L:2 * L]
        a['pools'] = flat_inputs[2 * L:3 * L]
        a['upsamples'] = flat_inputs[3 * L:4 * L]
        ind = 4 * L
        a['features'] = flat_inputs[ind]; ind += 1
        a['batch_weights'] = flat_inputs[ind]; ind += 1
        a['in_batches'] = flat_inputs[ind]; ind += 1
        a['out_batches'] = flat_inputs[ind]; ind += 1
        a['stack_lengths'] = flat_inputs[ind]; ind += 1
        self.anc_keypts_inds = flat_inputs[ind]; ind += 1
        self.pos_keypts_inds = flat_inputs[ind]; ind += 1
        ids = flat_inputs[ind]
        self.anc_id, self.pos_id = ids[0], ids[1]
        ind += 1
        a['backup_points'] = flat_inputs[ind]
        if self.config.dataset == 'KITTI' and len(flat_inputs) > ind + 1:
            ind += 1
            a['trans'] = flat_inputs[ind]
        return a

    def run(self, flat_inputs=None):
        """One forward pass = sess.run([out_features, out_scores], {dropout_prob: 1.0}) (utils/tester.py:198-199)."""
        if flat_inputs is None:
            flat_inputs = next(self.flat_inputs) if not isinstance(self.flat_inputs, (list, tuple)) else self.flat_inputs
        self.anchor_inputs = self.unpack(flat_inputs)
        with use_variables(self.variables):
            with self.variables.variable_scope('KernelPointNetwork'):
                pass
            self.out_features, self.out_scores = assemble_FCNN_blocks(self.anchor_inputs, self.config, self.dropout_prob)
        return self.out_features, self.out_scores

    __call__ = run

    def weights(self):
        """All variables (numpy), keyed by checkpoint name."""
        return self.variables.values
