"""oracle/ -- CPU restatement of the reference's sparse-conv path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
See oracle/spconv_ref.c for provenance ("parity unpinned": spconv 1.2.1 is absent offline and the
reference holds no fixtures for this path).
"""
