"""`index_max` -- same four entry points as the reference extension
(models/index_max_ext/index_max.cpp:154-159), backed by libusip_hip.so."""
try:
    from usip_amd import ops as _ops
except ImportError:          # imported as a top-level module with usip_amd/dropin on sys.path
    import os as _os
    import sys as _sys
    _sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))))
    from usip_amd import ops as _ops


def forward_cpu(data, index, K):
    """CPU single thread (index_max.cpp:73-112)."""
    return _ops.index_max_cpu(data, index, K, 1)


def forward_multi_thread_cpu(data, index, K, thread_num):
    """CPU multi-thread over channels (index_max.cpp:33-70)."""
    return _ops.index_max_cpu(data, index, K, thread_num)


def forward_cuda(data, index, K):
    """Device path (index_max.cpp:132-139). One gfx950 kernel serves both device entry points."""
    return _ops.index_max(data, index, K)


def forward_cuda_shared_mem(data, index, K):
    """Device path (index_max.cpp:141-148)."""
    return _ops.index_max(data, index, K)
