"""CPU oracle package -- TEST INFRASTRUCTURE ONLY (see oracle/tf_oracle.h).

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.
"""
