"""oracle/native.py -- TEST INFRASTRUCTURE ONLY.

ctypes bindings of oracle/libusip_oracle.so (the plain-C restatement in
usip_oracle.c).  numpy in, numpy out.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libusip_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "usip_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libusip_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def index_max(data: np.ndarray, index: np.ndarray, K: int) -> np.ndarray:
    """index_max.cpp:73-112 semantics. data f32 [B,C,N], index i32 [B,N] -> i32 [B,C,K]."""
    data = np.ascontiguousarray(data, dtype=np.float32)
    index = np.ascontiguousarray(index, dtype=np.int32)
    B, C, N = data.shape
    out = np.empty((B, C, K), dtype=np.int32)
    scratch = np.empty((K,), dtype=np.float32)
    lib().oracle_index_max_f32(_p(data, ctypes.c_float), _p(index, ctypes.c_int32),
                               _p(out, ctypes.c_int32), _p(scratch, ctypes.c_float),
                               B, C, N, K)
    return out


def ball_query(dist: np.ndarray, radius: float, K: int, return_prefix: bool = False):
    """ball_query_cuda.cu:22-46 semantics. dist f32 [B,M,N] -> i32 [B,M,K]
    (+ per-row scanned prefix length when return_prefix)."""
    dist = np.ascontiguousarray(dist, dtype=np.float32)
    B, M, N = dist.shape
    out = np.empty((B, M, K), dtype=np.int32)
    prefix = np.empty((B, M), dtype=np.int32)
    lib().oracle_ball_query_f32(_p(dist, ctypes.c_float), _p(out, ctypes.c_int32),
                                ctypes.c_float(radius), K, B, M, N, _p(prefix, ctypes.c_int32))
    return (out, prefix) if return_prefix else out


def pairwise_dist(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """a f32 [B,3,M], b f32 [B,3,N] -> f32 [B,M,N]: sqrt(((dx^2+dy^2)+dz^2)), no FMA."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    B, _, M = a.shape
    N = b.shape[2]
    out = np.empty((B, M, N), dtype=np.float32)
    lib().oracle_pairwise_dist_f32(_p(a, ctypes.c_float), _p(b, ctypes.c_float),
                                   _p(out, ctypes.c_float), B, M, N)
    return out
