"""Block library of the KPFCNN encoder/decoder, mirroring models/network_blocks.py of the reference for the
block types of the shipped architectures.  Every block keeps the reference signature
    block(layer_ind, inputs, features, radius, fdim, config, training)
and runs eagerly on the MI355X; variables live in the VariableStore set by `use_variables(...)` (the role of
the TF variable scopes).  Inference only: `training` must be False (the reference derives it from
dropout_prob < 0.99, :1071), batch norm uses the moving statistics and is fused, together with LeakyReLU(0.2)
and the residual add, into the epilogue of the producing contraction.
"""
import contextlib

import numpy as np

from .. import ops
from ..kernels import convolution_ops as conv_ops
from ..kernels.kernel_points import create_kernel_points

_STORE = None


@contextlib.contextmanager
def use_variables(store):
    global _STORE
    prev, _STORE = _STORE, store
    try:
        yield store
    finally:
        _STORE = prev


def _vs():
    if _STORE is None:
        raise RuntimeError('no VariableStore active: wrap the call in network_blocks.use_variables(store)')
    return _STORE


def variable_scope(name):
    return _vs().variable_scope(name)


# ---- utilities (models/network_blocks.py:37-83, :149-186) ----------------------------------------------------------

def weight_variable(shape):
    """:37-41 -> device tensor of the (created or loaded) `weights` variable of the current scope."""
    vs = _vs()
    return vs.tensor(vs.weight_variable(shape))


def ind_max_pool(x, inds):
    """:51-66."""
    return ops.ind_max_pool(x, inds)


def closest_pool(x, inds):
    """:69-83."""
    return ops.closest_pool_cat(x, inds)


def _bn(channels, use_batch_norm=True):
    """(col_scale, col_shift) of the current scope's inference batch norm (:149-165); without batch norm the
    reference adds a learnt `offset` vector (:162-165)."""
    vs = _vs()
    if use_batch_norm:
        return vs.folded_bn(vs.batch_norm_variables(channels))
    name = vs.get('offset', (channels,), lambda: np.zeros(channels))
    return None, vs.tensor(name)


def batch_norm(x, use_batch_norm=True, momentum=0.99, training=False):
    """:149-165, inference mode, as a stand-alone op."""
    if training:
        raise NotImplementedError('d3feat_amd implements the inference path only (training=False)')
    scale, shift = _bn(int(x.shape[1]), use_batch_norm)
    return ops.affine_act(x, scale, shift)


def leaky_relu(features, alpha=0.2):
    """:185-186."""
    return ops.affine_act(features, leaky=True, alpha=alpha)


def _epilogue(channels, config, leaky, residual=None):
    scale, shift = _bn(channels, config.use_batch_norm)
    return dict(col_scale=scale, col_shift=shift, residual=residual, leaky=leaky, alpha=0.2)


def _kernel_points(config, extent):
    """`kernel_points` variable of the current scope (kernels/convolution_ops.py:128-148): created with radius
    1.5*extent when absent."""
    vs = _vs()
    k = config.num_kernel_points

    def init():
        pts = create_kernel_points(1.5 * extent, k, num_kernels=1, dimension=3, fixed=config.fixed_kernel_points,
                                   rng=vs.rng)
        return pts.reshape((k, 3))
    name = vs.get('kernel_points', (k, 3), init)
    return vs.values[name]


def KPConv(query_points, support_points, neighbors_indices, features, K_values, radius, config, epilogue=None):
    """:86-103: extent = KP_extent * radius / density_parameter, then kernels.convolution_ops.KPConv."""
    extent = config.KP_extent * radius / config.density_parameter
    return conv_ops.KPConv(query_points, support_points, neighbors_indices, features, K_values,
                           fixed=config.fixed_kernel_points, KP_extent=extent, KP_influence=config.KP_influence,
                           aggregation_mode=config.convolution_mode, K_points=_kernel_points(config, extent),
                           epilogue=epilogue)


def _check_inference(training):
    if training:
        raise NotImplementedError('d3feat_amd implements the inference path only (training=False)')


# ---- blocks --------------------------------------------------------------------------------------------------------

def last_unary_block(layer_ind, inputs, features, radius, fdim, config, training):
    """:194-205: 1x1 convolution to 32 channels, no batch norm / activation."""
    w = weight_variable([int(features.shape[1]), 32])
    with ops.f32_output():      # descriptors and scores are computed from fp32 values whatever the feature storage
        return conv_ops.unary_convolution(features, w)


def unary_block(layer_ind, inputs, features, radius, fdim, config, training):
    """:207-219."""
    _check_inference(training)
    w = weight_variable([int(features.shape[1]), fdim])
    if isinstance(features, ops.UpsampleCat):
        # decoder: nearest upsampling + skip concatenation (models/D3Feat.py:55-63) fused into this contraction
        e = _epilogue(fdim, config, True)
        return ops.gemm_upsample_cat(features, w, col_scale=e['col_scale'], col_shift=e['col_shift'], leaky=True, alpha=0.2)
    return conv_ops.unary_convolution(features, w, epilogue=_epilogue(fdim, config, True))


def simple_block(layer_ind, inputs, features, radius, fdim, config, training):
    """:222-244."""
    _check_inference(training)
    w = weight_variable([config.num_kernel_points, int(features.shape[1]), fdim])
    return KPConv(inputs['points'][layer_ind], inputs['points'][layer_ind], inputs['neighbors'][layer_ind], features, w,
                  radius, config, epilogue=_epilogue(fdim, config, True))


def _resnetb(layer_ind, inputs, features, radius, fdim, config, strided):
    cin = int(features.shape[1])
    with variable_scope('conv1'):
        w = weight_variable([cin, fdim // 2])
        x = conv_ops.unary_convolution(features, w, epilogue=_epilogue(fdim // 2, config, True))
    with variable_scope('conv2'):
        w = weight_variable([config.num_kernel_points, fdim // 2, fdim // 2])
        if strided:
            q, s, nb = inputs['points'][layer_ind + 1], inputs['points'][layer_ind], inputs['pools'][layer_ind]
        else:
            q, s, nb = inputs['points'][layer_ind], inputs['points'][layer_ind], inputs['neighbors'][layer_ind]
        x = KPConv(q, s, nb, x, w, radius, config, epilogue=_epilogue(fdim // 2, config, True))
    # variables are created in the reference's order (conv3 before shortcut, :342-356 / :585-600) so that lazily created
    # random weights equal build_variables(seed)'s; the compute order below is free
    with variable_scope('conv3'):
        vs = _vs()
        w3 = vs.weight_variable([fdim // 2, 2 * fdim])
        bn3 = vs.batch_norm_variables(2 * fdim) if config.use_batch_norm else None
    with variable_scope('shortcut'):
        shortcut = ind_max_pool(features, inputs['pools'][layer_ind]) if strided else features
        need_sc = int(shortcut.shape[1]) != 2 * fdim
        # leaky_relu(bn(conv3(x)) + bn(shortcut(f))): both unary branches as ONE contraction over [x | f], the batch-norm
        # scales folded into the stacked weights -- no shortcut tensor, no residual re-read, one launch less
        fuse = need_sc and config.use_batch_norm and (fdim // 2) % 4 == 0
        sc_w = sc_bn = None
        if need_sc:
            vs = _vs()
            sc_w = vs.weight_variable([int(shortcut.shape[1]), 2 * fdim])
            if fuse:
                sc_bn = vs.batch_norm_variables(2 * fdim)
            else:
                shortcut = conv_ops.unary_convolution(shortcut, vs.tensor(sc_w), epilogue=_epilogue(2 * fdim, config, False))
    with variable_scope('conv3'):
        vs = _vs()
        if fuse:
            W, shift = vs.stacked_branches(w3, bn3, sc_w, sc_bn)
            return ops.gemm_cat2(x, shortcut, W, col_shift=shift, leaky=True, alpha=0.2)
        # leaky_relu(batch_norm(conv3) + shortcut): the add and the activation ride in the contraction's epilogue
        return conv_ops.unary_convolution(x, vs.tensor(w3), epilogue=_epilogue(2 * fdim, config, True, residual=shortcut))


def resnetb_block(layer_ind, inputs, features, radius, fdim, config, training):
    """:321-368: unary -> KPConv -> unary, shortcut (unary + BN iff the width changes), LeakyReLU(sum)."""
    _check_inference(training)
    return _resnetb(layer_ind, inputs, features, radius, fdim, config, False)


def resnetb_strided_block(layer_ind, inputs, features, radius, fdim, config, training):
    """:561-612: as resnetb but the KPConv queries the next layer's points through `pools`, and the shortcut is
    the max pooling of the input features (+ unary + BN iff the width changes)."""
    _check_inference(training)
    return _resnetb(layer_ind, inputs, features, radius, fdim, config, True)


def nearest_upsample_block(layer_ind, inputs, features, radius, fdim, config, training):
    """:971-979."""
    with variable_scope('nearest_upsample'):
        return closest_pool(features, inputs['upsamples'][layer_ind - 1])


def _materialized(block_fn):
    """Blocks other than `unary` take a real tensor: materialise a pending nearest-upsample concatenation first."""
    def wrapped(layer_ind, inputs, features, radius, fdim, config, training):
        if isinstance(features, ops.UpsampleCat):
            features = features.materialize()
        return block_fn(layer_ind, inputs, features, radius, fdim, config, training)
    wrapped.__name__ = block_fn.__name__
    wrapped.__doc__ = block_fn.__doc__
    return wrapped


def get_block_ops(block_name):
    """:982-1042 for the block types of the shipped architectures (results/*/parameters.txt:19)."""
    table = {'unary': unary_block, 'last_unary': last_unary_block, 'simple': simple_block, 'resnetb': resnetb_block,
             'resnetb_strided': resnetb_strided_block, 'nearest_upsample': nearest_upsample_block}
    if block_name in table:
        return table[block_name] if block_name == 'unary' else _materialized(table[block_name])
    known_unsupported = ('simple_strided', 'resnet', 'resnetb_light', 'resnetb_deformable', 'inception_deformable',
                         'resnetb_light_strided', 'resnetb_deformable_strided', 'inception_deformable_strided', 'vgg',
                         'max_pool', 'max_pool_wide', 'global_average', 'simple_upsample', 'resnetb_upsample')
    if block_name in known_unsupported:
        raise NotImplementedError('block "%s" is not used by any released D3Feat model and is not implemented'
                                  % block_name)
    raise ValueError('Unknown block name in the architecture definition : ' + block_name)


def assemble_CNN_blocks(inputs, config, dropout_prob):
    """:1052-1118: encoder.  Returns the skip list F (features before every strided block, and the last ones)."""
    r = config.first_subsampling_dl * config.density_parameter
    layer = 0
    fdim = config.first_features_dim
    features = inputs['features']
    F = []
    training = dropout_prob < 0.99
    block_in_layer = 0
    for block in config.architecture:
        if any(tmp in block for tmp in ('pool', 'strided', 'upsample', 'global')):
            F += [features]
        if 'upsample' in block:
            break
        with variable_scope('layer_{:d}/{:s}_{:d}'.format(layer, block.replace('_deformable', ''), block_in_layer)):
            features = get_block_ops(block)(layer, inputs, features, r, fdim, config, training)
        block_in_layer += 1
        if 'pool' in block or 'strided' in block:
            layer += 1
            r *= 2
            fdim *= 2
            block_in_layer = 0
        if 'global' in block:
            F += [features]
    return F
