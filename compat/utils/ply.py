from d3feat_amd.utils.ply import *  # noqa: F401,F403
from d3feat_amd.utils.ply import read_ply, write_ply  # noqa: F401
