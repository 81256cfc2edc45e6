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


def test_device_entry_points_take_host_tensors_to_the_host_twins_and_never_mix():
    """"`forward_cuda_shared_mem` of both modules accepts CPU tensors so networks.py runs on CPU unchanged"
    (BASELINE configs[0]).  ALL tensor arguments on the host -> the product's own host twins (csrc/host_cpu.cpp; never
    oracle/); a host/device mix or a wrong dtype is still the reference's CHECK_INPUT RuntimeError
    (index_max.cpp:119-121, ball_query.cpp:10-12); the tensor-level device wrappers (usip_amd.ops) refuse host tensors."""
    import usip_amd
    from usip_amd import ops
    im, bq = usip_amd.install()
    g = load_golden("index_max_cases.npz")
    d, i, K = torch.from_numpy(g["ties_data"]), torch.from_numpy(g["ties_index"]), int(g["ties_K"])
    for fn in (im.forward_cuda_shared_mem, im.forward_cuda):
        out = fn(d, i, K)
        assert out.dtype == torch.int32 and not out.is_cuda and np.array_equal(out.numpy(), g["ties_out"])
    # strided views, as networks.py may hand over: read through their strides like the reference's accessor
    wide = torch.zeros(d.shape[0], d.shape[1], 2 * d.shape[2])
    wide[:, :, ::2] = d
    assert np.array_equal(im.forward_cuda_shared_mem(wide[:, :, ::2], i, K).numpy(), g["ties_out"])
    gd = load_golden("dist_ball_cases.npz")
    dist = torch.from_numpy(gd["dist"])
    for fn in (bq.forward_cuda_shared_mem, bq.forward_cuda):
        out = fn(dist, float(gd["radius"]), int(gd["K"]))
        assert out.dtype == torch.int32 and np.array_equal(out.numpy(), gd["ball_idx_unpinned"])
    with pytest.raises(RuntimeError, match="int32|Int"):
        im.forward_cuda_shared_mem(d, i.long(), K)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        ops.index_max(d, i, K)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        ops.ball_query(dist, 0.5, 4)


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


def test_ball_front_end_host_twins_match_oracle_and_fixture():
    """BASELINE configs[0] (ModelNet40 detector, N=1024, M=64, batch 2 on PyTorch CPU): the Ball front end's two operators
    through the product's own host twins (usip_pairwise_dist_f32_cpu, usip_ball_query_f32_cpu) -- distances bit-identical
    to torch.norm on the pinned platform, rows equal to the C oracle's and to the committed fixture."""
    import usip_amd
    from oracle import native
    from usip_amd import ops, synth
    _, bq = usip_amd.install()
    g = load_golden("dist_ball_cases.npz")
    x, node = torch.from_numpy(g["x"]), torch.from_numpy(g["node"])
    dist = ops.pairwise_dist_cpu(node, x)
    assert np.array_equal(dist.numpy(), g["dist"])
    out = bq.forward_cpu(dist, float(g["radius"]), int(g["K"]))
    assert out.dtype == torch.int32 and np.array_equal(out.numpy(), g["ball_idx_unpinned"])
    b = synth.make_pair_batch(11, 2, 1024, 64, 3, "sphere")            # configs[0] shapes
    x, node = torch.from_numpy(b["src_pc"]), torch.from_numpy(b["src_node"])
    ref = torch.norm(node.unsqueeze(3) - x.unsqueeze(2), dim=1).contiguous()
    dist = ops.pairwise_dist_cpu(node, x)
    assert torch.equal(dist, ref)
    for radius, K in ((0.2, 64), (0.05, 32), (5.0, 8)):                # partially filled, empty and full balls
        assert np.array_equal(bq.forward_cpu(dist, radius, K).numpy(), native.ball_query(dist.numpy(), radius, K))
    with pytest.raises(RuntimeError):
        bq.forward_cpu(dist.double(), 0.2, 4)


def test_bn_momentum_decay_rule_matches_reference():
    """a-13 host logic: the momentum _EpochDecayBatchNorm switches to equals what the reference's MyBatchNorm1d/2d
    end up with (models/layers.py:61-71, :112-121; fixture from the reference itself): no change for epoch None / 0,
    momentum0 * decay ** (epoch // step) from epoch 1 on, clamped below at 0.01, and never reset afterwards."""
    from conftest import load_golden
    from usip_amd.layers import MyBatchNorm1d, MyBatchNorm2d
    g = load_golden("bn_decay_cases.npz")
    for cls in (MyBatchNorm1d, MyBatchNorm2d):
        for e in g["epochs"]:
            epoch = None if e < 0 else int(e)
            bn = cls(4, momentum=0.1, momentum_decay_step=2, momentum_decay=0.6)
            bn.decay_momentum(epoch)
            assert bn.momentum == float(g["momentum_%s" % ("none" if e < 0 else int(e))]), (cls, e)
            assert bn.epoch_driven == (epoch is not None)
        bn = cls(4, momentum=0.1, momentum_decay_step=2, momentum_decay=0.6)
        bn.decay_momentum(9)
        bn.decay_momentum(None)                     # the reference keeps the decayed value
        assert bn.momentum == float(g["momentum_9"])
        bn = cls(4, momentum=0.1, momentum_decay_step=None, momentum_decay=0.6)
        bn.decay_momentum(40)
        assert bn.momentum == 0.1
