"""Variable store: the role tf.variable_scope / tf.Variable / tf.train.Saver play in the reference.

Names are the reference's checkpoint names without the `KernelPointNetwork/` root, e.g.
  layer_0/simple_0/weights, layer_0/simple_0/kernel_points, layer_0/simple_0/batch_normalization/gamma,
  layer_1/resnetb_0/conv2/weights, uplayer_0/last_unary_1/weights.
Host master copies are numpy float32; device copies (and the folded inference batch-norm scale/shift vectors)
are created lazily on the model's GPU.
"""
import contextlib

import numpy as np
import torch


def truncated_normal(rng, shape, stddev):
    """tf.truncated_normal: N(0, stddev) re-drawn while |x| > 2 stddev."""
    out = rng.standard_normal(shape)
    bad = np.abs(out) > 2.0
    while np.any(bad):
        out[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(out) > 2.0
    return out * stddev


class VariableStore:
    def __init__(self, values=None, seed=42, device=None, create=True):
        self.values = dict(values) if values is not None else {}
        self.rng = np.random.default_rng(seed)
        self.device = device
        self.create = create
        self._scope = []
        self._dev = {}

    # ---- scopes -------------------------------------------------------------------------------------
    @contextlib.contextmanager
    def variable_scope(self, name):
        self._scope.append(name)
        try:
            yield
        finally:
            self._scope.pop()

    def full_name(self, name):
        return '/'.join(self._scope + [name])

    # ---- host values -----------------------------------------------------------------------------------
    def get(self, name, shape, init):
        full = self.full_name(name)
        if full not in self.values:
            if not self.create:
                raise KeyError('variable %s missing from the weight set' % full)
            self.values[full] = np.ascontiguousarray(init(), dtype=np.float32)
        v = self.values[full]
        if tuple(v.shape) != tuple(shape):
            raise ValueError('variable %s has shape %s, expected %s' % (full, tuple(v.shape), tuple(shape)))
        return full

    def weight_variable(self, shape):
        """models/network_blocks.py:37-41: truncated normal, stddev sqrt(2/shape[-1]), rounded to 3 decimals."""
        def init():
            w = truncated_normal(self.rng, tuple(shape), np.sqrt(2 / shape[-1]))
            return np.round(w.astype(np.float32) * np.float32(1000)) / np.float32(1000)
        return self.get('weights', shape, init)

    def batch_norm_variables(self, channels):
        """tf.layers.batch_normalization variables: gamma=1, beta=0, moving_mean=0, moving_variance=1."""
        with self.variable_scope('batch_normalization'):
            names = [self.get('gamma', (channels,), lambda: np.ones(channels)),
                     self.get('beta', (channels,), lambda: np.zeros(channels)),
                     self.get('moving_mean', (channels,), lambda: np.zeros(channels)),
                     self.get('moving_variance', (channels,), lambda: np.ones(channels))]
        return names

    # ---- device copies ------------------------------------------------------------------------------------
    def tensor(self, full):
        t = self._dev.get(full)
        if t is None:
            t = torch.from_numpy(np.ascontiguousarray(self.values[full], dtype=np.float32)).to(self.device)
            self._dev[full] = t
        return t

    def folded_bn(self, names, eps=1e-6):
        """Inference batch norm (models/network_blocks.py:149-160, epsilon 1e-6) as y = x*scale + shift with
        scale = gamma*rsqrt(var+eps), shift = beta - mean*scale  (tf.nn.batch_normalization's own factoring)."""
        key = ('bn',) + tuple(names)
        t = self._dev.get(key)
        if t is None:
            g, b, m, v = (self.values[n].astype(np.float32) for n in names)
            scale = (g / np.sqrt(v + np.float32(eps))).astype(np.float32)
            shift = (b - m * scale).astype(np.float32)
            t = (torch.from_numpy(scale).to(self.device), torch.from_numpy(shift).to(self.device))
            self._dev[key] = t
        return t

    def stacked_branches(self, w1, bn1, w2, bn2, eps=1e-6):
        """The two branches of a resnet block that meet in an add -- leaky(bn(x @ W1) + bn(f @ W2)), models/network_blocks.py:
        321-368 -- as ONE contraction: [x | f] @ [W1 * s1 ; W2 * s2] + (t1 + t2), the inference batch-norm scales folded into
        the stacked weights (s = gamma * rsqrt(var + eps), t = beta - mean * s).  -> (W [C1+C2, N], shift [N]) on the device."""
        key = ('stack', w1, w2)
        t = self._dev.get(key)
        if t is None:
            def fold(names):
                g, b, m, v = (self.values[n].astype(np.float32) for n in names)
                sc = (g / np.sqrt(v + np.float32(eps))).astype(np.float32)
                return sc, (b - m * sc).astype(np.float32)
            s1, t1 = fold(bn1)
            s2, t2 = fold(bn2)
            W = np.concatenate([self.values[w1] * s1[None, :], self.values[w2] * s2[None, :]], 0).astype(np.float32)
            t = (torch.from_numpy(np.ascontiguousarray(W)).to(self.device), torch.from_numpy(t1 + t2).to(self.device))
            self._dev[key] = t
        return t

    def invalidate_device(self):
        self._dev = {}


def build_variables(config, seed=42, in_features_dim=None, randomize_bn=False, device=None):
    """Create every variable of the network on the host without running it: a shape-only walk of
    models/network_blocks.py:1052-1118 (encoder) and models/D3Feat.py:19-63 (decoder).  The result has exactly the
    variable names / shapes of the reference's checkpoints (tests/test_host_logic.py checks
    it against the table decoded from results/Log_contraloss/snapshots/snap-54.index).

    randomize_bn: draw non-trivial batch-norm statistics (for parity tests; identity statistics hide bugs)."""
    from ..kernels.kernel_points import create_kernel_points
    vs = VariableStore(seed=seed, device=device)
    K = config.num_kernel_points
    cin = config.in_features_dim if in_features_dim is None else in_features_dim

    def bn(ch):
        names = vs.batch_norm_variables(ch)
        if randomize_bn:
            vs.values[names[0]] = (1.0 + 0.2 * vs.rng.standard_normal(ch)).astype(np.float32)
            vs.values[names[1]] = (0.1 * vs.rng.standard_normal(ch)).astype(np.float32)
            vs.values[names[2]] = (0.1 * vs.rng.standard_normal(ch)).astype(np.float32)
            vs.values[names[3]] = (0.5 + vs.rng.random(ch)).astype(np.float32)

    def kpconv(ci, co, layer):
        vs.weight_variable([K, ci, co])
        extent = config.KP_extent * (config.first_subsampling_dl * config.density_parameter * 2 ** layer) / config.density_parameter
        vs.get('kernel_points', (K, 3), lambda: create_kernel_points(1.5 * extent, K, 1, 3, config.fixed_kernel_points,
                                                                     rng=vs.rng).reshape(K, 3))

    layer, fdim, F, bil = 0, config.first_features_dim, [], 0
    start_i = len(config.architecture)
    for block_i, block in enumerate(config.architecture):
        if any(t in block for t in ('pool', 'strided', 'upsample', 'global')):
            F.append(cin)
        if 'upsample' in block:
            start_i = block_i
            break
        with vs.variable_scope('layer_{:d}/{:s}_{:d}'.format(layer, block.replace('_deformable', ''), bil)):
            if block == 'simple':
                kpconv(cin, fdim, layer)
                bn(fdim)
                cin = fdim
            elif block in ('resnetb', 'resnetb_strided'):
                with vs.variable_scope('conv1'):
                    vs.weight_variable([cin, fdim // 2]); bn(fdim // 2)
                with vs.variable_scope('conv2'):
                    kpconv(fdim // 2, fdim // 2, layer); bn(fdim // 2)
                with vs.variable_scope('conv3'):
                    vs.weight_variable([fdim // 2, 2 * fdim]); bn(2 * fdim)
                if cin != 2 * fdim:
                    with vs.variable_scope('shortcut'):
                        vs.weight_variable([cin, 2 * fdim]); bn(2 * fdim)
                cin = 2 * fdim
            elif block == 'unary':
                vs.weight_variable([cin, fdim]); bn(fdim)
                cin = fdim
            else:
                raise NotImplementedError(block)
        bil += 1
        if 'pool' in block or 'strided' in block:
            layer += 1
            fdim *= 2
            bil = 0
    layer = config.num_layers - 1
    fdim = config.first_features_dim * 2 ** layer
    bil = 0
    for block in config.architecture[start_i:]:
        with vs.variable_scope('uplayer_{:d}/{:s}_{:d}'.format(layer, block, bil)):
            if block == 'unary':
                vs.weight_variable([cin, fdim]); bn(fdim)
                cin = fdim
            elif block == 'last_unary':
                vs.weight_variable([cin, 32])
                cin = 32
            elif block != 'nearest_upsample':
                raise NotImplementedError(block)
        bil += 1
        if 'upsample' in block:
            layer -= 1
            fdim //= 2
            bil = 0
            cin += F[layer]
    return vs
