"""`kernels` as the reference's callers import it: aliases of d3feat_amd.kernels."""
