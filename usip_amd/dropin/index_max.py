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


def _all_on_host(*tensors):
    return all(hasattr(t, "is_cuda") and not t.is_cuda for t in tensors)


def _device_or_host(data, index, K):
    """The reference's own networks.py calls forward_cuda_shared_mem with whatever device its tensors live on
    (networks.py:118,131), and BASELINE configs[0] is that file on PyTorch CPU: when EVERY tensor argument is a host
    tensor the call goes to the product's own host twin (usip_index_max_f32_cpu, csrc/host_cpu.cpp -- the loop of
    index_max.cpp:98-109, read through strides as its accessor does).  Device tensors -- and any mix -- take the HIP
    kernel and its CHECK_INPUT rules (index_max.cpp:119-121): nothing on a GPU ever falls back to the host."""
    if _all_on_host(data, index):
        return _ops.index_max_cpu(data.contiguous(), index.contiguous(), K, 1)
    return _ops.index_max(data, index, K)


def forward_cuda(data, index, K):
    """Device path (index_max.cpp:132-139). One gfx950 kernel serves both device entry points."""
    return _device_or_host(data, index, K)


def forward_cuda_shared_mem(data, index, K):
    """Device path (index_max.cpp:141-148)."""
    return _device_or_host(data, index, K)
