from d3feat_amd.kernels.convolution_ops import KPConv, KPConv_ops, unary_convolution  # noqa: F401
