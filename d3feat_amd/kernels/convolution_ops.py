"""KPConv operators, same names / argument order as the reference's kernels/convolution_ops.py, running as two
HIP kernels on the MI355X: the fused gather + kernel-point-influence + aggregation (csrc/kpconv.hip) and the
(num_kp*Cin) x Cout contraction on the matrix cores (csrc/gemm_f32.hip).

    unary_convolution(features, K_values)                                       convolution_ops.py:90-99
    KPConv(query_points, support_points, neighbors_indices, features, K_values,
           fixed='center', KP_extent=1.0, KP_influence='linear', aggregation_mode='sum')     :102-158
    KPConv_ops(query_points, support_points, neighbors_indices, features, K_points, K_values,
               KP_extent, KP_influence, aggregation_mode)                                     :161-255

Extra keyword (not in the reference, defaults keep its behaviour): `epilogue`, a dict of
{col_scale, col_shift, residual, leaky, alpha} fused into the contraction's epilogue -- the inference
batch-norm / LeakyReLU / shortcut-add that always follow a KPConv in models/network_blocks.py.
The deformable variants (:258-627) are not part of any shipped architecture and are not provided.
"""
import numpy as np
import torch

from .. import ops
from .kernel_points import create_kernel_points



def unary_convolution(features, K_values, epilogue=None):
    """features f32[n, Cin] @ K_values f32[Cin, Cout]."""
    return ops.gemm(features, K_values, **(epilogue or {}))


def KPConv(query_points, support_points, neighbors_indices, features, K_values, fixed='center', KP_extent=1.0,
           KP_influence='linear', aggregation_mode='sum', K_points=None, epilogue=None):
    """convolution_ops.py:102-158.  In the reference the kernel point disposition is created here
    (radius 1.5*KP_extent) and kept in the `kernel_points` variable; pass `K_points` to use stored ones."""
    K_radius = 1.5 * KP_extent
    num_kpoints = int(K_values.shape[0])
    points_dim = int(query_points.shape[1])
    if K_points is None:
        K_points = create_kernel_points(K_radius, num_kpoints, num_kernels=1, dimension=points_dim, fixed=fixed)
        K_points = K_points.reshape((num_kpoints, points_dim)).astype(np.float32)
    return KPConv_ops(query_points, support_points, neighbors_indices, features, K_points, K_values, KP_extent,
                      KP_influence, aggregation_mode, epilogue=epilogue)


def KPConv_ops(query_points, support_points, neighbors_indices, features, K_points, K_values, KP_extent, KP_influence,
               aggregation_mode, epilogue=None):
    """convolution_ops.py:161-255.
    query_points [n,3]; support_points [n0,3]; neighbors_indices int32 [n,K] (pad = n0);
    features [n0,Cin]; K_points [num_kp,3] (numpy or tensor); K_values [num_kp,Cin,Cout] -> [n,Cout]."""
    if KP_influence not in ('constant', 'linear', 'gaussian'):
        raise ValueError('Unknown influence function type (config.KP_influence)')
    if aggregation_mode not in ('closest', 'sum'):
        raise ValueError("Unknown convolution mode. Should be 'closest' or 'sum'")
    num_kp, cin, cout = K_values.shape
    if features.shape[1] != cin:
        raise ValueError('KPConv: features have %d channels, K_values expects %d' % (features.shape[1], cin))
    # the fused forms address rows with one 24-bit multiply (csrc/common.h d3f_fits_u24: rows and leading dimensions < 2^24, rows x
    # leading dimension < 2^31) and return D3F_ERR_ARG beyond it; such stacks take aggregation + contraction, which has a general form
    # -- the same predicate as csrc/kpconv.hip kp_fits_u24: the feature matrix is also a buffer resource, rows x ld < 2^30
    def _u24(rows, ld, lim=31):
        return rows < (1 << 24) and ld < (1 << 24) and rows * ld < (1 << lim)
    fused_ok = _u24(int(query_points.shape[0]), int(neighbors_indices.stride(0))) and \
        _u24(int(features.shape[0]), int(features.stride(0)), 30)
    if cin == 1 and cout <= 256:
        # input layer: one fused kernel (gather + influences + 15-term contraction + epilogue)
        return ops.kpconv_fused_c1(query_points, support_points, neighbors_indices, features, K_points, K_values,
                                   KP_extent, KP_influence, aggregation_mode, **(epilogue or {}))
    if fused_ok and cin == 32 and cout == 32 and features.stride(0) % 4 == 0 and features.data_ptr() % 16 == 0:
        # level-0 convolutions: aggregation + contraction + epilogue in one kernel, the 113 MB wf tensor stays in LDS
        return ops.kpconv_fused32(query_points, support_points, neighbors_indices, features, K_points, K_values, KP_extent,
                                  KP_influence, aggregation_mode, **(epilogue or {}))
    if fused_ok and features.stride(0) % 4 == 0 and features.data_ptr() % 16 == 0 and \
            ops.kpconv_fused_supported(cin, cout, num_kp, KP_influence, aggregation_mode):
        # levels 1 and 2 (Cin = Cout = 64 / 128): the same, in tiles of 16 queries, the contraction fed from LDS
        return ops.kpconv_fused(query_points, support_points, neighbors_indices, features, K_points, K_values, KP_extent,
                                KP_influence, aggregation_mode, **(epilogue or {}))
    wf, inv_cnt = ops.kpconv_aggregate(query_points, support_points, neighbors_indices, features, K_points, KP_extent,
                                       KP_influence, aggregation_mode)
    return ops.gemm(wf, K_values.reshape(num_kp * cin, cout), row_scale=inv_cnt, **(epilogue or {}))
