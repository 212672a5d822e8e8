"""`utils` as the reference's callers import it: d3feat_amd.utils behind the reference's module names."""
