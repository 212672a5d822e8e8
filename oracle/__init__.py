"""TEST INFRASTRUCTURE ONLY.

`oracle/` holds the CPU checker for the D3Feat hot path: a plain-C restatement of the reference's
preprocessing (d3f_oracle.c), a numpy/torch-CPU restatement of its TensorFlow graph (network_np.py),
and -- when /root/reference is present at build time -- the reference's own C++ compiled in place
(oracle/_ref).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.
"""
