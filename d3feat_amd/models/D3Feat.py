"""Decoder + descriptor / detection head: models/D3Feat.py of the reference.

    assemble_FCNN_blocks(inputs, config, dropout_prob) -> (features f32[N0,32] l2-normalised, scores f32[N0,1])

The nearest-upsample gather and the skip concatenation (D3Feat.py:55-63) are one kernel; the l2 normalisation
(:65) and the whole soft detection module (:67-115) are one kernel after a per-cloud max reduction.
"""
import numpy as np
import torch

from .. import ops
from .network_blocks import assemble_CNN_blocks, get_block_ops, variable_scope


def assemble_FCNN_blocks(inputs, config, dropout_prob):
    F = assemble_CNN_blocks(inputs, config, dropout_prob)
    features = F[-1]
    layer = config.num_layers - 1
    r = config.first_subsampling_dl * config.density_parameter * 2 ** layer
    fdim = config.first_features_dim * 2 ** layer
    training = dropout_prob < 0.99
    start_i = 0
    for block_i, block in enumerate(config.architecture):
        if 'upsample' in block:
            start_i = block_i
            break
    block_in_layer = 0
    for block in config.architecture[start_i:]:
        with variable_scope('uplayer_{:d}/{:s}_{:d}'.format(layer, block, block_in_layer)):
            if block == 'nearest_upsample':
                # D3Feat.py:39-63 for this block type: closest_pool, then concat with the encoder skip F[layer-1]
                # ... kept lazy: the unary block that follows contracts [gathered | skip] directly (ops.UpsampleCat)
                with variable_scope('nearest_upsample'):
                    if isinstance(features, ops.UpsampleCat):
                        features = features.materialize()
                    features = ops.UpsampleCat(features, inputs['upsamples'][layer - 1], F[layer - 1])
            else:
                features = get_block_ops(block)(layer, inputs, features, r, fdim, config, training)
        block_in_layer += 1
        if 'upsample' in block:
            layer -= 1
            r *= 0.5
            fdim = fdim // 2
            block_in_layer = 0
            if block != 'nearest_upsample':
                raise NotImplementedError('only nearest_upsample decoders are implemented')
    if isinstance(features, ops.UpsampleCat):
        features = features.materialize()
    return detection_head(features, inputs)


def detection_head(features, inputs):
    """D3Feat.py:65-115 on the un-normalised `features`; needs inputs['neighbors'][0] and inputs['stack_lengths']."""
    lens = inputs['stack_lengths']
    dev = features.device
    lens_dev = ops.as_lens(lens, dev)
    # datasets/common.py:453-496: a row of in_batches holds the shadow index iff the cloud is shorter than the longest one,
    # or all clouds have the same length (extra pad column).  The head kernel derives that from the lengths on the device;
    # pass inputs['in_batches_padded'] (int32[B] on the device) to override.
    include_zero = inputs.get('in_batches_padded')
    ib = inputs.get('in_batches')
    group = ib.group if isinstance(ib, ops.StackGroups) else 0
    return ops.detect_head(features, inputs['neighbors'][0], lens_dev, include_zero, stack_group=group)
