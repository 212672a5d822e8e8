"""TEST INFRASTRUCTURE ONLY (oracle).  Never imported by d3feat_amd/.

CPU restatement (torch-CPU float32 tensors used as a numpy with threads; no autograd, no GPU) of the
reference's TensorFlow-1 graph for the inference hot path.  TensorFlow is not installable here
(SURVEY.md §8c), so this file follows the reference's Python line by line instead; every function cites
the lines it restates (paths relative to /root/reference).  Stock TF kernels (gather, matmul,
batch_normalization, softplus, l2_normalize) have standard semantics but unspecified summation order,
hence the 1e-4 fp32 tolerance the parity tests use for floating-point outputs.

PARITY STATUS: the integer / geometric inputs this graph consumes (pyramid) are pinned bit-exactly to the
reference's own C++ (tests/test_oracle_vs_ref.py); the floating-point graph itself is *unpinned* by any
reference test or golden vector (the reference ships none) -- it is pinned only by this
restatement, cross-checked against an independent float64 evaluation in tests/test_oracle_network.py.
"""
import math

import numpy as np
import torch

# --------------------------------------------------------------------------------------------------
# kernels/convolution_ops.py
# --------------------------------------------------------------------------------------------------


def unary_convolution(features, K_values):
    """kernels/convolution_ops.py:90-99 -- tf.matmul(features, K_values)."""
    return features @ K_values


def radius_gaussian(sq_r, sig, eps=1e-9):
    """kernels/convolution_ops.py:36-44."""
    return torch.exp(-sq_r / (2 * sig ** 2 + eps))


def KPConv_ops(query_points, support_points, neighbors_indices, features, K_points, K_values, KP_extent,
               KP_influence="linear", aggregation_mode="sum", chunk=4096):
    """kernels/convolution_ops.py:161-255, evaluated in row chunks (the TF graph materialises the same
    [n, K, 15, 3] / [n, K, Cin] intermediates for all rows at once; chunking does not change any value)."""
    query_points = torch.as_tensor(query_points, dtype=torch.float32)
    support_points = torch.as_tensor(support_points, dtype=torch.float32)
    features = torch.as_tensor(features, dtype=torch.float32)
    K_points = torch.as_tensor(K_points, dtype=torch.float32)
    K_values = torch.as_tensor(K_values, dtype=torch.float32)
    idx = torch.as_tensor(neighbors_indices).long()
    n_kp = K_points.shape[0]
    # :190-191 shadow support point at 1e6
    support = torch.cat([support_points, torch.ones_like(support_points[:1]) * 1e6], 0)
    # :234 zero feature row for shadow neighbours
    feats = torch.cat([features, torch.zeros_like(features[:1])], 0)
    out = torch.empty((query_points.shape[0], K_values.shape[2]), dtype=torch.float32)
    for a in range(0, query_points.shape[0], chunk):
        b = min(a + chunk, query_points.shape[0])
        ind = idx[a:b]
        neighbors = support[ind]                                   # :194
        neighbors = neighbors - query_points[a:b, None, :]         # :197
        differences = neighbors[:, :, None, :] - K_points          # :200-202
        sq_distances = (differences ** 2).sum(3)                   # :205
        if KP_influence == "constant":                             # :208-211
            all_weights = torch.ones_like(sq_distances).transpose(1, 2)
        elif KP_influence == "linear":                             # :213-216
            all_weights = torch.clamp(1 - torch.sqrt(sq_distances + 1e-10) / (2 * KP_extent), min=0.0)
            all_weights = all_weights.transpose(1, 2)
        elif KP_influence == "gaussian":                           # :218-222
            all_weights = radius_gaussian(sq_distances, KP_extent * 0.3).transpose(1, 2)
        else:
            raise ValueError("Unknown influence function type (config.KP_influence)")
        if aggregation_mode == "closest":                          # :227-229
            nn1 = sq_distances.argmin(2)
            all_weights = all_weights * torch.nn.functional.one_hot(nn1, n_kp).transpose(1, 2).float()
        elif aggregation_mode != "sum":
            raise ValueError("Unknown convolution mode. Should be 'closest' or 'sum'")
        nf = feats[ind]                                            # :237
        weighted = torch.matmul(all_weights, nf)                   # :240  [n, 15, Cin]
        kernel_outputs = torch.matmul(weighted.transpose(0, 1), K_values)   # :243-244 [15, n, Cout]
        o = kernel_outputs.sum(0)                                  # :247
        nsum = nf.sum(-1)                                          # :250
        num = (nsum > 0.0).float().sum(-1)                         # :251
        num = torch.clamp(num, min=1.0)                            # :252
        out[a:b] = o / num[:, None]                                # :253
    return out


# --------------------------------------------------------------------------------------------------
# models/network_blocks.py
# --------------------------------------------------------------------------------------------------


def ind_max_pool(x, inds):
    """models/network_blocks.py:51-66 (shadow row = per-channel minimum)."""
    x = torch.cat([x, x.min(0, keepdim=True)[0]], 0)
    return x[torch.as_tensor(inds).long()].max(1)[0]


def closest_pool(x, inds):
    """models/network_blocks.py:69-83 (shadow row = zeros; first column only)."""
    x = torch.cat([x, torch.zeros((1, x.shape[1]), dtype=x.dtype)], 0)
    return x[torch.as_tensor(inds).long()[:, 0]]


def batch_norm(x, W, scope, eps=1e-6):
    """models/network_blocks.py:149-160 in inference mode (training = dropout_prob < 0.99 is False at
    test time, :1071): tf.layers.batch_normalization with moving statistics, epsilon 1e-6."""
    g = torch.as_tensor(W[scope + "/batch_normalization/gamma"])
    b = torch.as_tensor(W[scope + "/batch_normalization/beta"])
    m = torch.as_tensor(W[scope + "/batch_normalization/moving_mean"])
    v = torch.as_tensor(W[scope + "/batch_normalization/moving_variance"])
    inv = torch.rsqrt(v + eps) * g           # tf.nn.batch_normalization: inv = rsqrt(var+eps)*scale
    return x * inv + (b - m * inv)           #                            x*inv + (offset - mean*inv)


def leaky_relu(x, alpha=0.2):
    """models/network_blocks.py:185-186."""
    return torch.where(x > 0, x, x * alpha)


class _Cfg:
    pass


def KPConv(query_points, support_points, neighbors_indices, features, K_values, K_points, radius, config):
    """models/network_blocks.py:86-103: extent = KP_extent * radius / density_parameter; the kernel points
    come from the `kernel_points` variable (kernels/convolution_ops.py:145-148)."""
    extent = config.KP_extent * radius / config.density_parameter
    return KPConv_ops(query_points, support_points, neighbors_indices, features, K_points, K_values, extent,
                      config.KP_influence, config.convolution_mode)


def _w(W, name):
    return torch.as_tensor(W[name], dtype=torch.float32)


def unary_block(layer_ind, inputs, features, radius, fdim, config, W, scope):
    """:207-219."""
    x = unary_convolution(features, _w(W, scope + "/weights"))
    return leaky_relu(batch_norm(x, W, scope))


def last_unary_block(layer_ind, inputs, features, radius, fdim, config, W, scope):
    """:194-205 (no BN, no activation)."""
    return unary_convolution(features, _w(W, scope + "/weights"))


def simple_block(layer_ind, inputs, features, radius, fdim, config, W, scope):
    """:222-244."""
    x = KPConv(inputs["points"][layer_ind], inputs["points"][layer_ind], inputs["neighbors"][layer_ind], features,
               _w(W, scope + "/weights"), _w(W, scope + "/kernel_points"), radius, config)
    return leaky_relu(batch_norm(x, W, scope))


def resnetb_block(layer_ind, inputs, features, radius, fdim, config, W, scope):
    """:321-368."""
    x = unary_convolution(features, _w(W, scope + "/conv1/weights"))
    x = leaky_relu(batch_norm(x, W, scope + "/conv1"))
    x = KPConv(inputs["points"][layer_ind], inputs["points"][layer_ind], inputs["neighbors"][layer_ind], x,
               _w(W, scope + "/conv2/weights"), _w(W, scope + "/conv2/kernel_points"), radius, config)
    x = leaky_relu(batch_norm(x, W, scope + "/conv2"))
    x = unary_convolution(x, _w(W, scope + "/conv3/weights"))
    x = batch_norm(x, W, scope + "/conv3")
    if features.shape[1] != 2 * fdim:
        shortcut = unary_convolution(features, _w(W, scope + "/shortcut/weights"))
        shortcut = batch_norm(shortcut, W, scope + "/shortcut")
    else:
        shortcut = features
    return leaky_relu(x + shortcut)


def resnetb_strided_block(layer_ind, inputs, features, radius, fdim, config, W, scope):
    """:561-612."""
    x = unary_convolution(features, _w(W, scope + "/conv1/weights"))
    x = leaky_relu(batch_norm(x, W, scope + "/conv1"))
    x = KPConv(inputs["points"][layer_ind + 1], inputs["points"][layer_ind], inputs["pools"][layer_ind], x,
               _w(W, scope + "/conv2/weights"), _w(W, scope + "/conv2/kernel_points"), radius, config)
    x = leaky_relu(batch_norm(x, W, scope + "/conv2"))
    x = unary_convolution(x, _w(W, scope + "/conv3/weights"))
    x = batch_norm(x, W, scope + "/conv3")
    shortcut = ind_max_pool(features, inputs["pools"][layer_ind])
    if shortcut.shape[1] != 2 * fdim:
        shortcut = unary_convolution(shortcut, _w(W, scope + "/shortcut/weights"))
        shortcut = batch_norm(shortcut, W, scope + "/shortcut")
    return leaky_relu(x + shortcut)


def nearest_upsample_block(layer_ind, inputs, features, radius, fdim, config, W, scope):
    """:971-979."""
    return closest_pool(features, inputs["upsamples"][layer_ind - 1])


_BLOCKS = {"unary": unary_block, "last_unary": last_unary_block, "simple": simple_block, "resnetb": resnetb_block,
           "resnetb_strided": resnetb_strided_block, "nearest_upsample": nearest_upsample_block}


def get_block_ops(name):
    """:982-1042 restricted to the block types of the shipped architectures."""
    if name not in _BLOCKS:
        raise ValueError("Unknown block name in the architecture definition : " + name)
    return _BLOCKS[name]


def assemble_CNN_blocks(inputs, config, W, trace=None):
    """models/network_blocks.py:1052-1118 (encoder)."""
    r = config.first_subsampling_dl * config.density_parameter
    layer = 0
    fdim = config.first_features_dim
    features = torch.as_tensor(inputs["features"], dtype=torch.float32)
    F = []
    block_in_layer = 0
    for block in config.architecture:
        if any(t in block for t in ("pool", "strided", "upsample", "global")):
            F.append(features)
        if "upsample" in block:
            break
        scope = "layer_{:d}/{:s}_{:d}".format(layer, block.replace("_deformable", ""), block_in_layer)
        features = get_block_ops(block)(layer, inputs, features, r, fdim, config, W, scope)
        if trace is not None:
            trace[scope] = features
        block_in_layer += 1
        if "pool" in block or "strided" in block:
            layer += 1
            r *= 2
            fdim *= 2
            block_in_layer = 0
        if "global" in block:
            F.append(features)
    return F


def assemble_FCNN_blocks(inputs, config, W, trace=None):
    """models/D3Feat.py:5-115: decoder, l2-normalised descriptors, detection scores."""
    F = assemble_CNN_blocks(inputs, config, W, trace)
    features = F[-1]
    layer = config.num_layers - 1
    r = config.first_subsampling_dl * config.density_parameter * 2 ** layer
    fdim = config.first_features_dim * 2 ** layer
    start_i = 0
    for block_i, block in enumerate(config.architecture):
        if "upsample" in block:
            start_i = block_i
            break
    block_in_layer = 0
    for block in config.architecture[start_i:]:
        scope = "uplayer_{:d}/{:s}_{:d}".format(layer, block, block_in_layer)
        features = get_block_ops(block)(layer, inputs, features, r, fdim, config, W, scope)
        if trace is not None:
            trace[scope] = features
        block_in_layer += 1
        if "upsample" in block:
            layer -= 1
            r *= 0.5
            fdim = fdim // 2
            block_in_layer = 0
            features = torch.cat((features, F[layer]), 1)
    # :65  tf.nn.l2_normalize(x, axis=1, epsilon=1e-10) = x * rsqrt(max(sum(x^2), eps))
    sq = (features ** 2).sum(1, keepdim=True)
    backup_features = features * torch.rsqrt(torch.clamp(sq, min=1e-10))
    scores = detection_head(features, inputs["neighbors"][0], inputs["in_batches"], inputs["stack_lengths"])
    return backup_features, scores


def detection_head(features, neighbor, in_batches, stack_lengths):
    """models/D3Feat.py:67-115."""
    features = torch.as_tensor(features, dtype=torch.float32)
    neighbor = torch.as_tensor(neighbor).long()
    in_batches = torch.as_tensor(in_batches).long()
    l0, l1 = int(stack_lengths[0]), int(stack_lengths[1])
    features = torch.cat([features, torch.zeros_like(features[:1])], 0)                  # :77-78
    neighbor = torch.cat([neighbor, torch.ones_like(neighbor[:1]) * (l0 + l1)], 0)       # :79-80
    m0 = features[in_batches[0]].max()                                                   # :84
    m1 = features[in_batches[1]].max()                                                   # :85
    max_per_sample = torch.cat([torch.ones((l0, 1)) * m0, torch.ones((l1 + 1, 1)) * m1], 0)   # :86-89
    features = features / (max_per_sample + 1e-6)                                        # :90
    nf = features[neighbor]                                                              # :93
    nsum = nf.sum(-1)                                                                    # :94
    num = (nsum != 0).sum(-1, keepdim=True)                                              # :95
    num = torch.clamp(num, min=1)                                                        # :96
    mean_features = nf.sum(1) / num.float()                                              # :97
    local_max_score = torch.nn.functional.softplus(features - mean_features)             # :98
    depth_wise_max = features.max(1, keepdim=True)[0]                                    # :101
    depth_wise_max_score = features / (1e-6 + depth_wise_max)                            # :102
    all_score = local_max_score * depth_wise_max_score                                   # :104
    score = all_score.max(1, keepdim=True)[0]                                            # :106
    return score[:-1]                                                                    # :115


# --------------------------------------------------------------------------------------------------
# datasets/common.py
# --------------------------------------------------------------------------------------------------


def get_batch_inds(stack_lengths):
    """datasets/common.py:408-451: [3, 2, 5] -> [0,0,0,1,1,2,2,2,2,2]."""
    return np.concatenate([np.full(int(n), b, np.int32) for b, n in enumerate(stack_lengths)]) \
        if len(stack_lengths) else np.zeros((0,), np.int32)


def stack_batch_inds(stack_lengths):
    """datasets/common.py:453-496: rows arange padded with N; an extra pad column if all lengths are equal."""
    lens = [int(x) for x in stack_lengths]
    n, mx = sum(lens), max(lens)
    rows, p = [], 0
    for l in lens:
        rows.append(np.concatenate([np.arange(p, p + l), np.full(mx - l, n)]).astype(np.int32))
        p += l
    out = np.stack(rows, 0)
    if n == mx * len(lens):
        out = np.concatenate([out, np.full((len(lens), 1), n, np.int32)], 1)
    return out


def pyramid_constants(config):
    """The exact fp32 radii / cell sizes the ops receive: python double arithmetic of
    datasets/common.py:1312,1355,1370,1396, then one cast to float32 by the TF op."""
    r_normal = config.first_subsampling_dl * config.KP_extent * 2.5
    out = []
    for _ in range(config.num_layers):
        dl = 2 * r_normal / (config.KP_extent * 2.5)
        out.append(dict(r=np.float32(r_normal), r_up=np.float32(2 * r_normal), dl_pool=np.float32(dl)))
        r_normal *= 2
    return out


def descriptor_input(config, stacked_points, stacked_features, stack_lengths, neighborhood_limits,
                     batch_neighbors, batch_subsampling):
    """datasets/common.py:1301-1413 (tf_descriptor_input) for the non-deformable architectures.
    `batch_neighbors(q, s, q_lens, s_lens, r)` and `batch_subsampling(points, lens, dl)` are the two CPU ops
    (pass the oracle's or the reference's)."""
    stacked_points = np.ascontiguousarray(stacked_points, np.float32)
    stack_lengths = np.asarray(stack_lengths, np.int32)
    batch_inds = get_batch_inds(stack_lengths)
    min_len = stack_lengths.min()
    batch_weights = np.float32(min_len) / stack_lengths.astype(np.float32)                 # :1309
    stacked_weights = batch_weights[batch_inds]                                            # :1310
    r_normal = config.first_subsampling_dl * config.KP_extent * 2.5                        # :1312
    layer_blocks = []
    input_points, input_neighbors, input_pools, input_upsamples, input_batches_len = [], [], [], [], []
    arch = config.architecture
    for block_i, block in enumerate(arch):
        if "global" in block or "upsample" in block:
            break
        if not ("pool" in block or "strided" in block):
            layer_blocks += [block]
            if block_i < len(arch) - 1 and not ("upsample" in arch[block_i + 1]):
                continue
        if layer_blocks:
            r = r_normal
            conv_i = batch_neighbors(stacked_points, stacked_points, stack_lengths, stack_lengths, np.float32(r))
        else:
            conv_i = np.zeros((0, 1), np.int32)
        if "pool" in block or "strided" in block:
            dl = 2 * r_normal / (config.KP_extent * 2.5)                                   # :1355
            pool_p, pool_b = batch_subsampling(stacked_points, stack_lengths, np.float32(dl))
            r = r_normal
            pool_i = batch_neighbors(pool_p, stacked_points, pool_b, stack_lengths, np.float32(r))       # :1367
            up_i = batch_neighbors(stacked_points, pool_p, stack_lengths, pool_b, np.float32(2 * r))     # :1370
        else:
            pool_i = np.zeros((0, 1), np.int32)
            pool_p = np.zeros((0, 3), np.float32)
            pool_b = np.zeros((0,), np.int32)
            up_i = np.zeros((0, 1), np.int32)
        lim = int(neighborhood_limits[len(input_points)])                                  # :399-406
        conv_i, pool_i, up_i = conv_i[:, :lim], pool_i[:, :lim], up_i[:, :lim]
        input_points += [stacked_points]
        input_neighbors += [conv_i]
        input_pools += [pool_i]
        input_upsamples += [up_i]
        input_batches_len += [stack_lengths]
        stacked_points, stack_lengths = pool_p, pool_b
        r_normal *= 2
        layer_blocks = []
    in_b = stack_batch_inds(input_batches_len[0])
    out_b = stack_batch_inds(input_batches_len[-1])
    return dict(points=input_points, neighbors=input_neighbors, pools=input_pools, upsamples=input_upsamples,
                features=np.ascontiguousarray(stacked_features, np.float32), batch_weights=stacked_weights,
                in_batches=in_b, out_batches=out_b, stack_lengths=np.asarray(input_batches_len[0], np.int32),
                batches_len=input_batches_len)


def neighbor_histograms(neighbors_list, hist_n):
    """datasets/common.py:645-647: per-layer histogram of valid-neighbour counts (valid = index < rows)."""
    hists = []
    for mat in neighbors_list:
        mat = np.asarray(mat)
        counts = np.sum(mat < mat.shape[0], axis=1)
        hists.append(np.bincount(counts, minlength=hist_n)[:hist_n])
    return np.vstack(hists).astype(np.int64)


def limits_from_histograms(neighb_hists, keep_ratio=0.8):
    """datasets/common.py:667-668."""
    hist_n = neighb_hists.shape[1]
    cumsum = np.cumsum(neighb_hists.T, axis=0)
    return np.sum(cumsum < (keep_ratio * cumsum[hist_n - 1, :]), axis=0).astype(np.int32)


def hist_size(config):
    """datasets/common.py:613 / :808."""
    return int(np.ceil(4 / 3 * np.pi * (config.density_parameter + 1) ** 3))


def forward(config, W, inputs, trace=None):
    """One `sess.run([out_features, out_scores])` of models/KPFCNN_model.py:129-132."""
    with torch.no_grad():
        inp = dict(inputs)
        inp["points"] = [torch.as_tensor(p, dtype=torch.float32) for p in inputs["points"]]
        desc, score = assemble_FCNN_blocks(inp, config, W, trace)
    return desc.numpy(), score.numpy()
