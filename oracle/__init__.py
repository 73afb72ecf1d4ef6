"""TEST INFRASTRUCTURE: CPU oracle of the HALO acquisition-scoring path (see halo_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
