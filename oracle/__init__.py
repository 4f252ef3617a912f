"""TEST INFRASTRUCTURE ONLY: CPU oracle for the nnmf()/nnlm() hot path (see nnlm_oracle.py, nnlm_ref.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
