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
        self._recipes = {}      # derived device tensors -> how to recompute them on the host (update_in_place)

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

    def _bn_host(self, names, eps):
        g, b, m, v = (self.values[n].astype(np.float32) for n in names)
        scale = (g / np.sqrt(v + np.float32(eps))).astype(np.float32)
        return scale, (b - m * scale).astype(np.float32)

    def folded_bn(self, names, eps=1e-6):
        """Inference batch norm (models/network_blocks.py:149-160, epsilon 1e-6) as y = x*scale + shift with
        scale = gamma*rsqrt(var+eps), shift = beta - mean*scale  (tf.nn.batch_normalization's own factoring)."""
        key = ('bn',) + tuple(names)
        t = self._dev.get(key)
        if t is None:
            scale, shift = self._bn_host(names, eps)
            t = (torch.from_numpy(scale).to(self.device), torch.from_numpy(shift).to(self.device))
            self._dev[key] = t
            self._recipes[key] = (lambda: self._bn_host(names, eps))
        return t

    def _stack_host(self, w1, bn1, w2, bn2, eps):
        s1, t1 = self._bn_host(bn1, eps)
        s2, t2 = self._bn_host(bn2, eps)
        W = np.concatenate([self.values[w1] * s1[None, :], self.values[w2] * s2[None, :]], 0).astype(np.float32)
        return np.ascontiguousarray(W), (t1 + t2).astype(np.float32)

    def stacked_branches(self, w1, bn1, w2, bn2, eps=1e-6):
        """The two branches of a resnet block that meet in an add -- leaky(bn(x @ W1) + bn(f @ W2)), models/network_blocks.py:
        321-368 -- as ONE contraction: [x | f] @ [W1 * s1 ; W2 * s2] + (t1 + t2), the inference batch-norm scales folded into
        the stacked weights (s = gamma * rsqrt(var + eps), t = beta - mean * s).  -> (W [C1+C2, N], shift [N]) on the device."""
        key = ('stack', w1, w2)
        t = self._dev.get(key)
        if t is None:
            W, shift = self._stack_host(w1, bn1, w2, bn2, eps)
            t = (torch.from_numpy(W).to(self.device), torch.from_numpy(shift).to(self.device))
            self._dev[key] = t
            self._recipes[key] = (lambda: self._stack_host(w1, bn1, w2, bn2, eps))
        return t

    def update_in_place(self, values):
        """New values for some variables of a model whose device copies may already be captured BY ADDRESS in HIP graphs
        (d3feat_amd.engine): the host masters are replaced and every device tensor made from them -- the plain copies, the folded
        batch-norm vectors, the stacked branch weights -- is rewritten IN PLACE, so a replayed graph reads the new values at the
        old addresses.  -> the device tensors that were rewritten (ops.refresh_packed_weights re-packs their packed copies, also
        in place).  Shapes cannot change; kernel points ride in the launch arguments of a captured graph and cannot change."""
        for name, v in values.items():
            if name not in self.values:
                raise KeyError('variable %s is not part of the weight set' % name)
            v = np.ascontiguousarray(v, dtype=np.float32)
            if v.shape != self.values[name].shape:
                raise ValueError('variable %s has shape %s, expected %s' % (name, v.shape, self.values[name].shape))
            if name.endswith('kernel_points') and not np.array_equal(v, self.values[name]):
                raise ValueError('kernel points are launch arguments of the captured graphs: build a new engine to change %s' % name)
            self.values[name] = v
        touched = []
        for key, t in self._dev.items():
            if isinstance(key, str):
                t.copy_(torch.from_numpy(self.values[key]))
                touched.append(t)
            else:
                host = self._recipes[key]()
                for dst, src in zip(t, host):
                    dst.copy_(torch.from_numpy(src))
                    touched.append(dst)
        return touched

    def invalidate_device(self):
        self._dev = {}
        self._recipes = {}


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
