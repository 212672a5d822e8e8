from d3feat_amd.kernels.kernel_points import *  # noqa: F401,F403
