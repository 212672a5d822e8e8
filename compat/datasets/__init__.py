"""`datasets` as the reference's callers import it: d3feat_amd.datasets behind the reference's module names."""
