"""`tensorflow` as the reference's inference callers see it -- exactly the symbols demo_registration.py and
utils/tester.py touch (last row), backed by d3feat_amd on the MI355X.  NOT TensorFlow: an eager stand-in so
that the reference's scripts run unchanged (python -m d3feat_amd.compat_run /path/to/demo_registration.py).

    tf.float32 / tf.int32 / tf.string              dtype tags (demo_registration.py:94)
    tf.ones, tf.shape                              demo_registration.py:102 (device tensors)
    tf.get_collection, tf.GraphKeys                demo_registration.py:117, utils/tester.py:143
    tf.train.Saver(vars, max_to_keep).restore      :118,135 / tester.py:144,161 -> utils.tf_checkpoint reader
    tf.ConfigProto(...).gpu_options.allow_growth   :123-127
    tf.Session(config).run(fetches, feed_dict)     :128,131,150,155-156 -> one forward pass of the HIP path per run
    tf.global_variables_initializer, tf.multiply   :131, (:146, commented out in the reference)

What a fetch means here: the model object (models.KPFCNN_model.KernelPointFCNN of the compat tree) exposes its tensors as
`Fetch` handles; Session.run evaluates the model ONCE for every distinct model among the fetches -- pulling the next element
of the dataset iterator, like `iter.get_next()` does inside the reference's graph -- and returns numpy values (strings as
bytes, like TF).
"""
import os
import warnings

import numpy as np

__version__ = "1.12.0-d3feat_amd-compat"

float32, int32, int64, string = "float32", "int32", "int64", "string"
_TORCH_DTYPES = {"float32": "float32", "int32": "int32", "int64": "int64"}

_MODELS = []          # every compat KernelPointFCNN built in this process ("the graph")


def _torch():
    import torch
    return torch


def shape(x):
    """tf.shape(x): static here (eager)."""
    return tuple(int(v) for v in x.shape)


def ones(shp, dtype=float32):
    torch = _torch()
    return torch.ones(tuple(int(v) for v in shp), dtype=getattr(torch, _TORCH_DTYPES[dtype]),
                      device=torch.device("cuda", torch.cuda.current_device()))


def multiply(a, b):
    return a * b


class GraphKeys:
    GLOBAL_VARIABLES = "variables"
    TRAINABLE_VARIABLES = "trainable_variables"


class _VariableRef:
    """One model variable, by checkpoint name (what tf.get_collection returns elements of)."""

    def __init__(self, model, name):
        self.model, self.name = model, "KernelPointNetwork/" + name + ":0"
        self._key = name

    def assign(self, value):
        return _Assign(self, value)

    def value(self):
        return self.model.inner.variables.values[self._key]

    def __mul__(self, other):
        return self.value() * other

    __rmul__ = __mul__


class _Assign:
    def __init__(self, ref, value):
        self.ref, self.new = ref, value

    def _run(self):
        vs = self.ref.model.inner.variables
        vs.values[self.ref._key] = np.ascontiguousarray(self.new, dtype=np.float32)
        vs.invalidate_device()


def get_collection(key, scope=None):
    out = []
    for m in _MODELS:
        for name in sorted(m.variable_names()):
            if scope is None or ("KernelPointNetwork/" + name).startswith(scope):
                out.append(_VariableRef(m, name))
    return out


class _InitOp:
    """tf.global_variables_initializer(): the variables are created (reference initialisers) when the model is built."""

    def _run(self):
        for m in _MODELS:
            m.ensure_variables()


def global_variables_initializer():
    return _InitOp()


class _GpuOptions:
    allow_growth = False
    per_process_gpu_memory_fraction = 1.0


class ConfigProto:
    def __init__(self, device_count=None, log_device_placement=False, allow_soft_placement=False, **kw):
        self.device_count = device_count or {}
        self.log_device_placement, self.allow_soft_placement = log_device_placement, allow_soft_placement
        self.gpu_options = _GpuOptions()


class Fetch:
    """A tensor of the model graph: resolved by Session.run."""

    def __init__(self, model, what):
        self.model, self.what = model, what

    def __hash__(self):
        return id(self)


class Placeholder:
    def __init__(self, name, default=None):
        self.name, self.default = name, default

    def __hash__(self):
        return id(self)


def placeholder(dtype=float32, shape=None, name="placeholder"):
    return Placeholder(name)


class Session:
    def __init__(self, target="", graph=None, config=None):
        self.config = config
        if config is not None and config.device_count.get("GPU", 1) == 0:
            # the reference demo pins TF to the CPU (demo_registration.py:121-123); this path has no CPU implementation
            warnings.warn("ConfigProto(device_count={'GPU': 0}) ignored: d3feat_amd computes on the MI355X only")

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def run(self, fetches, feed_dict=None, options=None, run_metadata=None):
        feed_dict = feed_dict or {}
        done = {}

        def resolve(f):
            if f is None:
                return None
            if isinstance(f, (list, tuple)):
                return type(f)(resolve(x) for x in f) if isinstance(f, tuple) else [resolve(x) for x in f]
            if isinstance(f, dict):
                return {k: resolve(v) for k, v in f.items()}
            if isinstance(f, Fetch):
                if id(f.model) not in done:
                    done[id(f.model)] = f.model.evaluate(feed_dict)
                return done[id(f.model)][f.what]
            if hasattr(f, "_run"):
                return f._run()
            if callable(f):            # dataset.test_init_op
                return f()
            raise TypeError("Session.run: cannot fetch %r" % (f,))
        return resolve(fetches)


class _Saver:
    """tf.train.Saver(var_list).restore(sess, prefix): TensorBundle reader of d3feat_amd.utils.tf_checkpoint."""

    def __init__(self, var_list=None, max_to_keep=5, **kw):
        self.var_list = var_list

    def restore(self, sess, save_path):
        from d3feat_amd.utils import tf_checkpoint
        data = save_path + ".data-00000-of-00001"
        if not os.path.exists(data):
            if os.environ.get("D3FEAT_COMPAT_ALLOW_MISSING_CHECKPOINT", "0") != "1":
                raise FileNotFoundError(
                    "tf.train.Saver.restore: %s is missing (the public checkout of the reference ships snap-*.index/.meta "
                    "without the tensor data).  Supply the released blob, or set D3FEAT_COMPAT_ALLOW_MISSING_CHECKPOINT=1 "
                    "to keep the initialised weights (variable names / shapes are still validated against the index)." % data)
            index = tf_checkpoint.read_index(save_path + ".index")
            for m in _MODELS:
                m.ensure_variables()
                vals = m.inner.variables.values
                for name, e in index.items():
                    if tf_checkpoint.is_model_variable(name):
                        key = name[len(tf_checkpoint.ROOT_SCOPE):]
                        if key not in vals or tuple(vals[key].shape) != tuple(e.shape):
                            raise ValueError("checkpoint variable %s %s does not match the model" % (name, tuple(e.shape)))
            warnings.warn("checkpoint data missing: weights keep their initial values (%s)" % save_path)
            return
        weights = tf_checkpoint.load_checkpoint(save_path)
        for m in _MODELS:
            m.load_weights(weights)

    def save(self, *a, **kw):
        raise NotImplementedError("the compat layer covers the inference callers only")


class train:
    Saver = _Saver


class errors:
    class OutOfRangeError(Exception):
        pass


def load_op_library(path):
    raise NotImplementedError("custom ops are built into libd3feat_amd.so (d3feat_amd.tf_custom_ops); nothing to load")
