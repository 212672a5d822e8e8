from d3feat_amd.models.network_blocks import *  # noqa: F401,F403
from d3feat_amd.models.network_blocks import (KPConv, assemble_CNN_blocks, get_block_ops, ind_max_pool, closest_pool,  # noqa: F401
                                              weight_variable, batch_norm, leaky_relu)
