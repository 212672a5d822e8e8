"""`models` as the reference's callers import it: thin aliases of d3feat_amd.models (see compat/README.md)."""
