"""d3feat_amd -- MI355X-native D3Feat inference hot path.

Layout (mirrors the reference's module paths for the hot path only):
  csrc/                        HIP kernels for gfx950 + the C ABI (include/d3feat_amd.h) -> lib/libd3feat_amd.so
  ops.py                       torch-tensor front end of the C ABI (device memory + streams only)
  tf_custom_ops.py             batch_ordered_neighbors / batch_grid_subsampling / ...  (reference: tf_custom_ops/)
  cpp_wrappers/cpp_subsampling grid_subsampling.compute(...)                            (reference: cpp_wrappers/)
  kernels/convolution_ops.py   KPConv, KPConv_ops, unary_convolution                    (reference: kernels/)
  kernels/kernel_points.py     kernel point dispositions
  models/                      network_blocks, D3Feat, KPFCNN_model                     (reference: models/)
  datasets/common.py           Dataset: descriptor pyramid, neighbour calibration       (reference: datasets/common.py)
  utils/config.py              Config (parameters.txt compatible)                       (reference: utils/config.py)
  parallel.py                  fragment sharding across GPUs + final RCCL gather
"""
__version__ = "0.1.0"
