"""oracle/ -- TEST INFRASTRUCTURE ONLY (CPU restatement of the reference's hot path).

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.
Nothing under usip_amd/ imports this package.
"""
