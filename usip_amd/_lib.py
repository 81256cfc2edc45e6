"""ctypes loader of libusip_hip.so -- the only way the product reaches its kernels.

There is no fallback: if the library is missing or fails to load, every operator raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# USIP_LIB=<path>: load another BUILD of the same library (same-box A/B of two builds, tools/ab_build.sh); never a fallback
LIB_PATH = os.environ.get("USIP_LIB") or os.path.join(_HERE, "libusip_hip.so")
ABI = 5                      # = "abi=<n>" of usip_version(): bumped with every incompatible change of include/usip_hip.h
_lib = None

_f32p = ctypes.c_void_p
_i32p = ctypes.c_void_p
_int = ctypes.c_int
_flt = ctypes.c_float
_stream = ctypes.c_void_p

# name -> argtypes; must list every symbol include/usip_hip.h declares (tests check this).
SIGNATURES = {
    "usip_version": ([], ctypes.c_char_p),
    "usip_set_tuning": ([ctypes.c_char_p, _int], _int),
    "usip_tuning_value": ([_int], _int),
    "usip_wgrad_defer": ([_int], _int),
    "usip_wgrad_defer_on": ([_stream], _int),
    "usip_wgrad_defer_hold": ([_int], _int),
    "usip_wgrad_flush": ([_stream], _int),
    "usip_index_max_f32": ([_f32p, _i32p, _i32p, _int, _int, _int, _int, _stream], _int),
    "usip_index_max_f32_cpu": ([_f32p, _i32p, _i32p, _int, _int, _int, _int, _int], _int),
    "usip_ball_query_f32": ([_f32p, _i32p, _flt, _int, _int, _int, _int, _stream], _int),
    "usip_ball_query_f32_cpu": ([_f32p, _i32p, _flt, _int, _int, _int, _int], _int),
    "usip_pairwise_dist_f32_cpu": ([_f32p, _f32p, _f32p, _int, _int, _int], _int),
    "usip_pairwise_dist_f32": ([_f32p, _f32p, _f32p, _int, _int, _int, _stream], _int),
    "usip_som_assign_f32": ([_f32p, _f32p, _i32p, _int, _int, _int, _stream], _int),
    "usip_som_cluster_f32": ([_f32p, _i32p, _f32p, _i32p, _f32p, _int, _int, _int, _stream], _int),
    "usip_som_cluster_csr_f32": ([_f32p, _i32p, _i32p, _i32p, _f32p, _i32p, _f32p, _int, _int, _int, _stream], _int),
    "usip_index_max_values_f32": ([_f32p, _i32p, _i32p, _i32p, _f32p, _int, _int, _int, _int, _int, _stream], _int),
    "usip_index_max_values_backward_add_f32": ([_f32p, _i32p, _i32p, _f32p, _int, _int, _int, _int, _int, _int,
                                                _stream], _int),
    "usip_index_max_values_backward_f32": ([_f32p, _i32p, _i32p, _i32p, _f32p, _int, _int, _f32p, _int, _int, _int, _int,
                                            _stream], _int),
    "usip_mlp_narrow_forward_blocks": ([_int, _int, _int, _int], _int),
    "usip_mlp_narrow_forward_f32": ([_f32p, _int, _f32p, _f32p, _int, _f32p, _f32p, _int, _f32p, _int, _f32p, _int, _int,
                                     _int, _int, _stream], _int),
    "usip_detector_head_f32": ([_f32p, _f32p, _flt, _f32p, _f32p, _int, _int, _stream], _int),
    "usip_detector_head_backward_f32": ([_f32p, _f32p, _f32p, _f32p, _int, _int, _stream], _int),
    "usip_rigid_transform_f32": ([_f32p, _f32p, _f32p, _f32p, _f32p, _int, _int, _int, _stream], _int),
    "usip_detector_loss_combine_f32": ([_f32p, _f32p, _flt, _f32p, ctypes.c_longlong, _stream], _int),
    "usip_fill_scaled_f32": ([_f32p, _flt, _f32p, ctypes.c_longlong, _stream], _int),
    "usip_mlp_split3_blocks": ([_int, _int, _int], _int),
    "usip_mlp_split3_multi_f32": ([ctypes.c_void_p, _int, _int, _stream], _int),
    "usip_bn_group_dy_sum_f32": ([_f32p, _f32p, _f32p, _f32p, _int, _int, _int, _int, _stream], _int),
    "usip_csr_by_index_i32": ([_i32p, _i32p, _i32p, _int, _int, _int, _stream], _int),
    "usip_segment_sum_supported": ([_int, _int], _int),
    "usip_segment_sum_f32": ([_f32p, _i32p, _i32p, _f32p, _int, _int, _int, _int, _int, _int, _stream], _int),
    "usip_nearest_backward_f32": ([_f32p, _f32p, _f32p, _i32p, _f32p, _f32p, _f32p, _int, _int, _int, _int, _stream],
                                  _int),
    "usip_nearest_nd_f32": ([_f32p, _f32p, _f32p, _i32p, _int, _int, _int, _int, _stream], _int),
    "usip_chamfer_prob_f32": ([_f32p, _i32p, _f32p, _i32p, _f32p, _f32p, _f32p, _int, _int, _int, _stream], _int),
    "usip_chamfer_prob_backward_f32": ([_f32p, _f32p, _i32p, _f32p, _i32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p,
                                        _int, _int, _int, _stream], _int),
    "usip_nearest_workspace": ([_int, _int, _int], ctypes.c_longlong),
    "usip_nearest_f32": ([_f32p, _f32p, _f32p, _i32p, _f32p, _i32p, _int, _int, _int, _stream], _int),
    "usip_mlp_gemm_tiles": ([_int, _int, _int], _int),
    "usip_mlp_gemm_f32": ([_f32p, _int, _f32p, _f32p, _f32p, _int, _f32p, _f32p, _int, _f32p, _i32p, _int,
                           _f32p, _int, _f32p, _int, _int, _int, _int, _stream], _int),
    "usip_mlp_gemm_bf16": ([_f32p, _int, _f32p, _f32p, _f32p, _int, _f32p, _f32p, _int, _f32p, _i32p, _int,
                            _f32p, _int, _f32p, _int, _int, _int, _int, _stream], _int),
    "usip_mlp_gemm_f32x3": ([_f32p, _int, _f32p, _f32p, _f32p, _int, _f32p, _f32p, _int, _f32p, _i32p, _int,
                             _f32p, _int, _f32p, _int, _int, _int, _int, _stream], _int),
    "usip_mlp_gemm_f32x3_used": ([_int, _int, _int, _int], _int),
    "usip_mlp_x3p_tile_rows": ([_int, _int, _int], _int),
    "usip_mlp_x3p_tile_cols": ([_int, _int, _int, _int, _int], _int),
    "usip_mlp_split3_bytes": ([_int, _int], ctypes.c_longlong),
    "usip_mlp_split3_f32": ([_f32p, _int, _int, _int, _int, ctypes.c_void_p, _stream], _int),
    "usip_mlp_gemm_x3p_f32": ([ctypes.c_void_p, _f32p, _f32p, _f32p, _int, _f32p, _f32p, _int, _f32p, _i32p, _int,
                               _f32p, _int, _f32p, _int, _int, _int, _int, _stream], _int),
    "usip_mlp_gemm_x2r_tiles": ([_int, _int], _int),
    "usip_mlp_gemm_x2r_f32": ([ctypes.c_void_p, _f32p, _f32p, _f32p, _int, _f32p, _f32p, _int, _f32p, _i32p, _int, _f32p,
                               _int, _f32p, _int, _int, _int, _int, _stream], _int),
    "usip_mlp_split2h_f32": ([_f32p, _int, _int, _int, _int, ctypes.c_void_p, _stream], _int),
    "usip_mlp_gemm_x2h_f32": ([ctypes.c_void_p, _f32p, _f32p, _f32p, _int, _f32p, _f32p, _int, _f32p, _i32p, _int,
                               _f32p, _int, _f32p, _int, _int, _int, _int, _stream], _int),
    "usip_mlp_gemm_x2d_red_tiles": ([_int, _int, _int, _int, _int], _int),
    "usip_mlp_gemm_x2f_used": ([_int] * 10, _int),
    "usip_mlp_gemm_x2h_red_f32": ([ctypes.c_void_p, _f32p, _f32p, _f32p, _int, _f32p, _i32p, _int, _f32p, _f32p, _f32p, _f32p,
                                   _f32p, _int, _int, _int, _int, _int, _stream], _int),
    "usip_mlp_wgrad_f32x3_used": ([_int, _int, _int, _int], _int),
    "usip_mlp_wgrad_f32x3_blocks": ([_int, _int, _int, _int], _int),
    "usip_bn_pool_backward_reduce_f32": ([_f32p, _i32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _int, _f32p, _f32p,
                                          _f32p, _f32p, _int, _int, _int, _int, _int, _stream], _int),
    "usip_bn_finalize_f32": ([_f32p, _int, _int, ctypes.c_longlong, _f32p, _f32p, _flt, _flt, _f32p, _f32p, _f32p,
                              _f32p, _f32p, _stream], _int),
    "usip_bn_apply_f32": ([_f32p, _f32p, _f32p, _int, _int, _int, _int, _stream], _int),
    "usip_bn_backward_reduce_f32": ([_f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _int, _f32p, _f32p, _f32p, _f32p,
                                     _f32p, _int, _int, _int, _int, _int, _stream], _int),
    "usip_mlp_wgrad_workspace": ([_int, _int, _int, _int], ctypes.c_longlong),
    "usip_mlp_wgrad_blocks": ([_int, _int, _int, _int], _int),
    "usip_mlp_wgrad_f32": ([_f32p, _f32p, _f32p, _int, _f32p, _f32p, _f32p, _i32p, _int, _f32p, _f32p, _int, _int,
                            _int, _int, _int, _int, _stream], _int),
    "usip_mlp_wgrad_bf16": ([_f32p, _f32p, _f32p, _int, _f32p, _f32p, _f32p, _i32p, _int, _f32p, _f32p, _int, _int,
                             _int, _int, _int, _int, _stream], _int),
    "usip_mlp_wgrad_f32x3": ([_f32p, _f32p, _f32p, _int, _f32p, _f32p, _f32p, _i32p, _int, _f32p, _f32p, _int, _int,
                              _int, _int, _int, _int, _stream], _int),
    "usip_mlp_wgrad_x2h_f32": ([_f32p, _f32p, _f32p, _int, _f32p, _f32p, _f32p, _i32p, _int, _f32p, _f32p, _int, _int,
                                _int, _int, _int, _int, _stream], _int),
    "usip_mlp_narrow_backward_supported": ([_int, _int, _int], _int),
    "usip_mlp_narrow_backward_workspace": ([_int, _int, _int], ctypes.c_longlong),
    "usip_mlp_narrow_backward_blocks": ([_int, _int, _int], _int),
    "usip_mlp_layer_backward_x2h_supported": ([_int, _int, _int, _int], _int),
    "usip_mlp_layer_backward_x2h_workspace": ([_int, _int, _int, _int], ctypes.c_longlong),
    "usip_mlp_layer_backward_x2h_blocks": ([_int, _int, _int, _int], _int),
    "usip_mlp_layer_backward_x2h_f32": ([_f32p, _f32p, _f32p, _f32p, _i32p, _int, _f32p, _int, _f32p, ctypes.c_void_p,
                                         _f32p, _int, _f32p, _f32p, _int, _f32p, _f32p, _int, _int, _int, _int, _stream],
                                        _int),
    "usip_mlp_layer_backward_x2h_ws_f32": ([_f32p, _f32p, _f32p, _f32p, _int, _f32p, ctypes.c_void_p, _f32p, _int, _f32p,
                                            _f32p, _int, _f32p, _f32p, _int, _f32p, _f32p, _int, _int, _int, _int, _stream],
                                           _int),
    "usip_mlp_wsum_finalize_f32": ([_f32p, _f32p, _int, _int, _f32p, _f32p, _int, _f32p, _int, _stream], _int),
    "usip_mlp_narrow_backward_f32": ([_f32p, _f32p, _f32p, _f32p, _int, _f32p, _f32p, _int, _f32p, _int, _f32p, _f32p,
                                      _int, _f32p, _int, _int, _int, _int, _stream], _int),
    "usip_bn_backward_finalize_f32": ([_f32p, _int, _int, ctypes.c_longlong, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p,
                                       _stream], _int),
    "usip_bn_backward_finalize_max_f32": ([_f32p, _int, _int, ctypes.c_longlong, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p,
                                           _f32p, _int, _f32p, _int, _stream], _int),
    "usip_group_max_act_f32": ([_f32p, _f32p, _int, _f32p, _i32p, _f32p, _int, _int, _int, _int, _stream], _int),
    "usip_group_gather_f32": ([_f32p, _i32p, _f32p, _f32p, _int, _int, _int, _int, _int, _int, _int, _int,
                               _stream], _int),
    "usip_group_gather_backward_f32": ([_f32p, _i32p, _f32p, _int, _int, _int, _int, _int, _int, _int, _stream], _int),
    "usip_group_max_f32": ([_f32p, _f32p, _i32p, ctypes.c_longlong, _int, _stream], _int),
    "usip_group_max_backward_f32": ([_f32p, _i32p, _f32p, ctypes.c_longlong, _int, _stream], _int),
    "usip_group_max_backward_add_f32": ([_f32p, _i32p, _f32p, ctypes.c_longlong, _int, _stream], _int),
    "usip_multi_transpose_f32": ([_f32p, _f32p, _i32p, _int, _int, _stream], _int),
    "usip_adam_step_f32": ([_f32p, _f32p, _f32p, _f32p, _f32p, _flt, _flt, _flt, _flt, ctypes.c_longlong, _stream], _int),
    "usip_adam_step_hyper_f32": ([_f32p, _f32p, _f32p, _f32p, _f32p, _f32p, ctypes.c_longlong, _stream], _int),
    "usip_knn_f32": ([_f32p, _f32p, _i32p, _int, _int, _int, _int, _stream], _int),
    "usip_knn_points_f32": ([_f32p, _f32p, _i32p, _int, _int, _int, _int, _stream], _int),
    "usip_knn_layer_supported": ([_int, _int, _int], _int),
    "usip_knn_layer_forward_f32": ([_f32p, _f32p, _int, _f32p, _f32p, _i32p, _f32p, _f32p, _int, _int, _int, _int, _int,
                                    _stream], _int),
    "usip_knn_layer_backward_f32": ([_f32p, _f32p, _f32p, _int, _f32p, _i32p, _i32p, _f32p, _f32p, _int, _int, _int, _int,
                                     _int, _stream], _int),
    "usip_fps_f32": ([_f32p, _i32p, _i32p, _int, _int, _int, _stream], _int),
    "usip_nms_f32": ([_f32p, _f32p, _flt, _i32p, _i32p, _int, _int, _stream], _int),
    "usip_ball_query_coords_f32": ([_f32p, _f32p, _i32p, _flt, _int, _int, _int, _int, _stream], _int),
}


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "usip_amd: %s is missing. Build it with `python -m usip_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU/PyTorch fallback." % LIB_PATH)
        # PyTorch-ROCm ships its own libamdhip64.so (same SONAME as /opt/rocm's).  The tensors we
        # are handed live in THAT runtime, so it must be the one already loaded when our library's
        # dependency is resolved: import torch first, then check that exactly one HIP runtime is mapped.
        import torch  # noqa: F401
        try:
            l = ctypes.CDLL(LIB_PATH)
        except OSError as e:
            raise RuntimeError("usip_amd: cannot load %s: %s" % (LIB_PATH, e)) from e
        try:
            with open("/proc/self/maps") as f:
                runtimes = sorted({ln.split()[-1] for ln in f if "libamdhip64" in ln})
        except OSError:
            runtimes = []
        if len(runtimes) > 1:
            raise RuntimeError("usip_amd: two HIP runtimes are loaded (%s); import torch before anything "
                               "that links libamdhip64" % ", ".join(runtimes))
        for name, (args, res) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.argtypes = args
            fn.restype = res
        # ctypes cannot see a changed signature, and results depend on three compile flags (an SLP-vectorised build gave
        # rare wrong values on gfx950): the library says what it is, anything else is refused
        ver = l.usip_version().decode()
        if ("abi=%d" % ABI) not in ver.split() or "flags=no-contract,no-fast-math,no-slp" not in ver.split():
            raise RuntimeError("usip_amd: %s reports %r; this package needs abi=%d built by usip_amd/build.py "
                               "(-ffp-contract=off -fno-fast-math -fno-slp-vectorize)" % (LIB_PATH, ver, ABI))
        # measurement knobs for same-box A/B runs: USIP_TUNE="x3_gemm_tile=3,index_max_ch=4" (include/usip_hip.h)
        for item in filter(None, os.environ.get("USIP_TUNE", "").split(",")):
            name, _, value = item.partition("=")
            if l.usip_set_tuning(name.strip().encode(), int(value)) != 0:
                raise RuntimeError("usip_amd: USIP_TUNE names an unknown knob or value: %r" % item)
        _lib = l
    return _lib


def check(rc: int, what: str):
    if rc == 0:
        return
    if rc < 0:
        raise RuntimeError("usip_amd: %s: invalid argument (USIP_EINVAL)" % what)
    raise RuntimeError("usip_amd: %s: HIP error %d" % (what, rc))
