"""`ball_query` -- same two entry points as the reference extension
(models/ball_query_ext/ball_query.cpp:45-48), backed by libusip_hip.so."""
try:
    from usip_amd import ops as _ops
except ImportError:          # imported as a top-level module with usip_amd/dropin on sys.path
    import os as _os
    import sys as _sys
    _sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))))
    from usip_amd import ops as _ops


def _device_or_host(node_to_point_dist, radius, K):
    """networks.py:698 hands over the distance matrix on whatever device the model lives on; BASELINE configs[0] is
    that file on PyTorch CPU.  A HOST matrix goes to the product's own host twin (usip_ball_query_f32_cpu,
    csrc/host_cpu.cpp: the scan of ball_query_cuda.cu:22-46, one row at a time); a device matrix takes the HIP kernel
    and its CHECK_INPUT rules (ball_query.cpp:10-12).  Nothing on a GPU ever falls back to the host."""
    if hasattr(node_to_point_dist, "is_cuda") and not node_to_point_dist.is_cuda:
        return _ops.ball_query_cpu(node_to_point_dist.detach().contiguous(), radius, K)
    return _ops.ball_query(node_to_point_dist, radius, K)


def forward_cuda_shared_mem(node_to_point_dist, radius, K):
    """ball_query.cpp:33-39. dist f32 [B,M,N] -> i32 [B,M,K], on the device the matrix is on."""
    return _device_or_host(node_to_point_dist, radius, K)


def forward_cuda(node_to_point_dist, radius, K):
    """The reference's forward_cuda is an unimplemented stub that prints and returns garbage
    (ball_query.cpp:23-31); here it is an alias of the working entry point."""
    return _device_or_host(node_to_point_dist, radius, K)


def forward_cpu(node_to_point_dist, radius, K):
    """No reference counterpart (its CPU path is the stub above): the host twin of the device kernel, for BASELINE
    configs[0] -- the reference's plumbing case on PyTorch CPU.  dist f32 [B,M,N] on the HOST -> i32 [B,M,K]."""
    return _ops.ball_query_cpu(node_to_point_dist, radius, K)
