from d3feat_amd.models.D3Feat import *  # noqa: F401,F403
from d3feat_amd.models.D3Feat import assemble_FCNN_blocks, detection_head  # noqa: F401
