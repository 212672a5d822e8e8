from d3feat_amd.utils.config import Config  # noqa: F401
