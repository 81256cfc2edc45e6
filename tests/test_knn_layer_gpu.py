"""The first layer of GeneralKNNFusionModule without the gathered tensor (usip_amd/csrc/knn_layer.hip, round 6):
   Y[b,:,m,k] = W_c . (database[b,:,n] - query[b,:,m]) + (W_f . feat[b] + bias)[:, n],  n = idx[b,m,k]
against fp64 restatements of models/layers.py:422-431 + :208-216 (gather, decenter, cat, conv1x1, BatchNorm backward),
and against the gather + generic-layer form it replaces (USIP_KNN_LAYER=0), through the whole module."""
import numpy as np
import pytest
import torch

from conftest import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

SHAPES = [(3, 32, 48, 40, 8, 64), (2, 128, 512, 512, 16, 256), (1, 5, 33, 17, 4, 10)]   # B, C, N, M, K, Cout


def _inputs(B, C, N, M, K, Cout, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    feat = torch.randn(B, C, N, generator=g)
    database = torch.randn(B, 3, N, generator=g) * 10
    query = database[:, :, torch.randperm(N, generator=g)[:M]] if M <= N else torch.randn(B, 3, M, generator=g) * 10
    query = query.contiguous()
    idx = torch.randint(0, N, (B, M, K), generator=g, dtype=torch.int32)
    idx[0, 0, :] = 0                                       # a long segment and (for N > M*K/..) empty ones
    W = torch.randn(Cout, 3 + C, generator=g) * 0.2
    bias = torch.randn(Cout, generator=g) * 0.1
    return [t.to(DEV) for t in (feat, database, query, idx, W, bias)]


def _gathered64(feat, database, query, idx):
    B, C, N = feat.shape
    _, M, K = idx.shape
    flat = idx.long().view(B, 1, M * K)
    d = torch.gather(database, 2, flat.expand(B, 3, M * K)).view(B, 3, M, K) - query.unsqueeze(3)   # fp32, as layers.py:428-430
    f = torch.gather(feat, 2, flat.expand(B, C, M * K)).view(B, C, M, K)
    return torch.cat((d, f), dim=1).double().view(B, 3 + C, M * K)


@pytest.mark.parametrize("shape", SHAPES)
def test_knn_layer_forward_matches_the_gathered_convolution(shape):
    from usip_amd import ops
    B, C, N, M, K, Cout = shape
    feat, database, query, idx, W, bias = _inputs(*shape)
    assert ops.knn_layer_supported(N, M, K)
    U = (torch.einsum("oc,bcn->bon", W[:, 3:].double(), feat.double()) + bias.double().view(1, -1, 1)).float().contiguous()
    Y, stats = ops.knn_layer_forward(U, W.contiguous(), database, query, idx)
    X = _gathered64(feat, database, query, idx)
    ref = torch.einsum("oc,bcp->bop", W.double(), X) + bias.double().view(1, -1, 1)
    assert_close(Y.cpu().numpy(), ref.cpu().numpy(), name="Y")
    st = stats.view(2, Cout, B).double()
    assert_close(st[0].sum(1).cpu().numpy(), Y.double().sum(dim=(0, 2)).cpu().numpy(), name="sum")
    assert_close(st[1].sum(1).cpu().numpy(), (Y.double() ** 2).sum(dim=(0, 2)).cpu().numpy(), name="sum of squares")
    Y2, stats2 = ops.knn_layer_forward(U, W.contiguous(), database, query, idx)
    assert torch.equal(Y, Y2) and torch.equal(stats, stats2)


@pytest.mark.parametrize("shape", SHAPES)
def test_knn_layer_backward_matches_segment_sums_of_the_batchnorm_gradient(shape):
    from usip_amd import ops
    B, C, N, M, K, Cout = shape
    feat, database, query, idx, W, bias = _inputs(*shape, seed=1)
    g = torch.Generator(device="cpu").manual_seed(7)
    P = M * K
    Y = torch.randn(B, Cout, P, generator=g).to(DEV)
    dZ = torch.randn(B, Cout, P, generator=g).to(DEV)
    coef4 = (torch.randn(4, Cout, generator=g) * 0.5).to(DEV).contiguous()
    start, perm = ops.csr_by_index(idx.view(B, P), N)
    dcoord = ops.group_gather(database, idx, sub=query).view(B, 3, P)
    dU, dwc = ops.knn_layer_backward(dZ, Y, coef4, True, dcoord, start, perm, M, K)
    z = ops.bn_apply(Y, coef4[:2].contiguous(), False)     # the kernels' own fma(y, a1, a0): the same ReLU decisions
    c = coef4.double().view(4, 1, Cout, 1)
    dY = c[0] * torch.where(z > 0, dZ, torch.zeros_like(dZ)).double() + c[2] * Y.double() + c[3]
    ref_dU = torch.zeros(B, Cout, N, dtype=torch.float64, device=DEV)
    ref_dU.scatter_add_(2, idx.long().view(B, 1, P).expand(B, Cout, P), dY)
    X = _gathered64(feat, database, query, idx)
    ref_dwc = torch.einsum("bop,bjp->oj", dY, X[:, :3])
    assert_close(dU.cpu().numpy(), ref_dU.cpu().numpy(), name="dU")
    assert_close(dwc.cpu().numpy(), ref_dwc.cpu().numpy(), name="dW[:, :3]")
    dU2, dwc2 = ops.knn_layer_backward(dZ, Y, coef4, True, dcoord, start, perm, M, K)
    from usip_amd import _lib
    _lib.lib().usip_set_tuning(b"r5_forms", 128)        # the one-row-per-workgroup form: the same sums in the same order
    try:
        dU3, dwc3 = ops.knn_layer_backward(dZ, Y, coef4, True, dcoord, start, perm, M, K)
    finally:
        _lib.lib().usip_set_tuning(b"r5_forms", 0)
    assert torch.equal(dU, dU3)
    assert_close(dwc3.cpu().numpy(), ref_dwc.cpu().numpy(), name="dW[:, :3], one row per workgroup")
    assert torch.equal(dU, dU2) and torch.equal(dwc, dwc2)          # fixed summation order: the same bits


def _module_run(first_layer_form, shape, matmul_mode_name):
    from usip_amd import functional as Fh, layers, ops
    B, C, N, M, K, Cout = shape
    prev_mode = ops.set_matmul_mode(matmul_mode_name)
    prev = Fh.KNN_FIRST_LAYER
    Fh.KNN_FIRST_LAYER = first_layer_form
    try:
        torch.manual_seed(3)
        mod = layers.GeneralKNNFusionModule(3 + C, [Cout, Cout], [2 * Cout, 2 * Cout], "relu", "batch").to(DEV).train()
        feat, database, query, idx, _, _ = _inputs(*shape, seed=2)
        feat.requires_grad_(True)
        out = mod(query, database, feat, K)
        r = torch.randn(out.shape, generator=torch.Generator(device="cpu").manual_seed(5)).to(DEV)
        (out * r).sum().backward()
        torch.cuda.synchronize()
        grads = {k: p.grad.detach().clone() for k, p in mod.named_parameters()}
        bufs = {k: v.detach().clone() for k, v in mod.state_dict().items() if "running" in k}
        return out.detach(), feat.grad.detach(), grads, bufs
    finally:
        Fh.KNN_FIRST_LAYER = prev
        ops.set_matmul_mode(prev_mode)


@pytest.mark.parametrize("mode", ["f32", "f32x2"])
@pytest.mark.parametrize("shape", SHAPES[:2])
def test_knn_fusion_module_with_and_without_the_gathered_tensor(shape, mode):
    """The whole module (max-pools, four more layers) through both forms of its first layer: forward to 1e-5, BatchNorm
    buffers to 1e-5, gradients at the free-running bound (two correct fp32 forwards take a few max-pool / ReLU decisions
    differently: DESIGN.md 3)."""
    out_new, dfeat_new, g_new, b_new = _module_run(True, shape, mode)
    out_old, dfeat_old, g_old, b_old = _module_run(False, shape, mode)
    assert_close(out_new.cpu().numpy(), out_old.cpu().numpy(), name="module output")
    for k in b_old:
        assert_close(b_new[k].cpu().numpy(), b_old[k].cpu().numpy(), name=k)

    def rel(a, b):
        return float((a - b).norm() / b.norm().clamp_min(1e-30))
    biggest = max(float(v.norm()) for v in g_old.values())
    assert rel(dfeat_new, dfeat_old) <= 2e-2
    for k, v in g_old.items():
        if float(v.norm()) < 1e-5 * biggest:
            continue                                       # a bias in front of a BatchNorm: analytically zero
        assert rel(g_new[k], v) <= 2e-2, k


def test_knn_first_layer_is_what_the_training_module_runs():
    from usip_amd import functional as Fh, layers, prof
    shape = SHAPES[0]
    B, C, N, M, K, Cout = shape
    mod = layers.GeneralKNNFusionModule(3 + C, [Cout], [Cout], "relu", "batch").to(DEV).train()
    feat, database, query, idx, _, _ = _inputs(*shape)
    feat.requires_grad_(True)
    prof.reset()
    prof.enable(True)
    try:
        mod(query, database, feat, K).sum().backward()
        names = set(prof.summary())
    finally:
        prof.enable(False)
        prof.reset()
    assert "knn_layer_fwd" in names and "knn_layer_bwd" in names and "segment_sum" not in names, names
    with torch.no_grad():                                  # inference keeps the gather + layer form
        assert not Fh.knn_first_layer_supported(feat, idx, mod.layers_before[0].conv.bias, mod.layers_before[0].norm, True)


def test_knn_layer_edge_cases():
    """Empty batch: nothing is launched, empty outputs.  A shape the kernels do not take (M * K not a multiple of 4, or a
    row of M * K floats beyond 64 KiB of LDS) is refused by usip_knn_layer_supported and the module keeps the gather +
    layer form; out-of-range indices are clamped (never read outside the cloud)."""
    from usip_amd import functional as Fh, layers, ops
    assert not ops.knn_layer_supported(48, 5, 3) and not ops.knn_layer_supported(512, 2048, 16)
    assert ops.knn_layer_supported(512, 512, 16) and ops.knn_layer_supported(64, 64, 16)
    U = torch.empty(0, 8, 16, device=DEV)
    W = torch.randn(8, 3 + 4, device=DEV)
    Y, stats = ops.knn_layer_forward(U, W, torch.empty(0, 3, 16, device=DEV), torch.empty(0, 3, 4, device=DEV),
                                     torch.empty(0, 4, 4, dtype=torch.int32, device=DEV))
    assert Y.shape == (0, 8, 16)
    B, C, N, M, K, Cout = 2, 6, 20, 5, 3, 8                  # M * K = 15: the module falls back
    mod = layers.GeneralKNNFusionModule(3 + C, [Cout], [Cout], "relu", "batch").to(DEV).train()
    feat, database, query, idx, _, _ = _inputs(B, C, N, M, K, Cout)
    feat.requires_grad_(True)
    assert not Fh.knn_first_layer_supported(feat, idx, mod.layers_before[0].conv.bias, mod.layers_before[0].norm, True)
    mod(query, database, feat, K).sum().backward()
    assert feat.grad is not None and torch.isfinite(feat.grad).all()
    B, C, N, M, K, Cout = SHAPES[0]
    feat, database, query, idx, W, bias = _inputs(*SHAPES[0])
    bad = idx.clone()
    bad[0, 0, 0], bad[1, 2, 3] = -7, N + 100
    U = torch.randn(B, Cout, N, device=DEV)
    Y, _ = ops.knn_layer_forward(U, W.contiguous(), database, query, bad)
    ref, _ = ops.knn_layer_forward(U, W.contiguous(), database, query, bad.clamp(0, N - 1))
    assert torch.equal(Y, ref)
