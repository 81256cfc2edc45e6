"""Drop-in replacements of the reference's two extension modules.

Either put this directory on sys.path (`import index_max` then finds index_max.py here) or call
usip_amd.install().  Function names, argument order and tensor contracts are those of the
reference's pybind11 modules (models/index_max_ext/index_max.cpp:154-159,
models/ball_query_ext/ball_query.cpp:45-48).
"""
