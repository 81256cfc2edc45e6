"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/usip_hip.h declares, and the host-side mirror of the reference interface behaves like
the reference's (names, argument meaning, error behaviour).  No device compute here."""
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "usip_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(usip_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from usip_amd import _lib
    lib = _lib.lib()
    declared = _declared_symbols()
    assert "usip_ball_query_f32" in declared and "usip_index_max_f32" in declared
    for name in declared:
        assert hasattr(lib, name), "libusip_hip.so does not export %s" % name
        assert name in _lib.SIGNATURES, "no ctypes signature for %s" % name
    assert b"gfx950" in lib.usip_version()


def test_dropin_modules_have_reference_entry_points():
    import usip_amd
    im, bq = usip_amd.install()
    import index_max
    import ball_query
    assert index_max is im and ball_query is bq
    for fn in ("forward_cpu", "forward_multi_thread_cpu", "forward_cuda", "forward_cuda_shared_mem"):
        assert callable(getattr(index_max, fn))          # index_max.cpp:154-159
    for fn in ("forward_cuda", "forward_cuda_shared_mem"):
        assert callable(getattr(ball_query, fn))         # ball_query.cpp:45-48


def test_device_entry_points_reject_host_and_noncontiguous_tensors():
    """CHECK_INPUT of the reference -> RuntimeError (index_max.cpp:119-121, ball_query.cpp:10-12).
    There is no CPU fallback behind the device entry points."""
    import usip_amd
    im, bq = usip_amd.install()
    d = torch.randn(2, 3, 16)
    i = torch.zeros(2, 16, dtype=torch.int32)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        im.forward_cuda_shared_mem(d, i, 4)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        im.forward_cuda(d, i, 4)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        bq.forward_cuda_shared_mem(torch.rand(2, 3, 16), 0.5, 4)


@pytest.mark.parametrize("tag", ["random", "ties", "floor", "empty", "nan", "tiny", "n_lt_wave"])
@pytest.mark.parametrize("threads", [1, 3])
def test_host_entry_points_match_reference_golden(tag, threads):
    """index_max.forward_cpu / forward_multi_thread_cpu twins against vectors produced by the
    reference's own C++."""
    import usip_amd
    im, _ = usip_amd.install()
    g = load_golden("index_max_cases.npz")
    d, i, K = torch.from_numpy(g[tag + "_data"]), torch.from_numpy(g[tag + "_index"]), int(g[tag + "_K"])
    out = im.forward_cpu(d, i, K) if threads == 1 else im.forward_multi_thread_cpu(d, i, K, threads)
    assert out.dtype == torch.int32
    assert np.array_equal(out.numpy(), g[tag + "_out"])
