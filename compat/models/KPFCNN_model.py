"""models.KPFCNN_model for the reference's unchanged callers (demo_registration.py:207, test_3dmatch.py:69): the same
constructor, the tensors of models/KPFCNN_model.py:86-132 exposed as graph handles that `tensorflow.Session.run` resolves."""
import numpy as np

import tensorflow as tf
from d3feat_amd.models.KPFCNN_model import KernelPointFCNN as _Model
from d3feat_amd.models.variables import build_variables


class KernelPointFCNN:
    def __init__(self, flat_inputs, config):
        self.config = config
        self.inner = _Model(flat_inputs if not isinstance(flat_inputs, (list, tuple)) else iter([flat_inputs]), config)
        self.dropout_prob = tf.Placeholder("dropout_prob", 1.0)
        for name in ("anchor_inputs", "out_features", "out_scores", "anc_id", "pos_id", "accuracy"):
            setattr(self, name, tf.Fetch(self, name))
        self._have_vars = False
        tf._MODELS.append(self)

    # ---- variables (the role of the TF variable collection) ----------------------------------------------------------
    def ensure_variables(self):
        if not self._have_vars:
            # same initialisers / creation order as running the graph builder (models/network_blocks.py:37-41)
            vals = build_variables(self.config, seed=42).values
            for k, v in vals.items():
                self.inner.variables.values.setdefault(k, v)
            self._have_vars = True

    def variable_names(self):
        self.ensure_variables()
        return list(self.inner.variables.values.keys())

    def load_weights(self, weights):
        self.ensure_variables()
        vals = self.inner.variables.values
        missing = [k for k in vals if k not in weights]
        if missing:
            raise KeyError("checkpoint lacks %d model variables, e.g. %s" % (len(missing), missing[:3]))
        for k in vals:
            if tuple(weights[k].shape) != tuple(vals[k].shape):
                raise ValueError("variable %s: checkpoint %s, model %s" % (k, weights[k].shape, vals[k].shape))
            vals[k] = np.ascontiguousarray(weights[k], np.float32)
        self.inner.variables.invalidate_device()

    # ---- one sess.run -------------------------------------------------------------------------------------------------
    def evaluate(self, feed_dict):
        p = feed_dict.get(self.dropout_prob, 1.0)
        if p < 0.99:
            raise NotImplementedError("dropout_prob < 0.99 selects the training graph (models/network_blocks.py:1071): "
                                      "inference only here")
        self.ensure_variables()
        desc, score = self.inner.run()
        a = self.inner.anchor_inputs

        def host(x):
            if x is None:
                return None
            if isinstance(x, (list, tuple)):
                return [host(v) for v in x]
            return x.cpu().numpy() if hasattr(x, "cpu") else np.asarray(x)
        inputs = {k: host(v) for k, v in a.items()}

        def as_bytes(s):
            return s if isinstance(s, bytes) else str(s).encode("utf-8")
        return {"anchor_inputs": inputs, "out_features": desc.cpu().numpy(), "out_scores": score.cpu().numpy(),
                "anc_id": as_bytes(self.inner.anc_id), "pos_id": as_bytes(self.inner.pos_id),
                "accuracy": np.float32(0.0)}       # the loss graph (KPFCNN_model.py:143-191) is training-only
