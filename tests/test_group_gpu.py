"""GPU parity of the grouping / pooling kernels (csrc/group.hip) and of the pooled-concat shared-MLP
layer against plain PyTorch on the same inputs (float64 truth for the float paths)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    return float((a.detach().double() - b.detach().double()).abs().max() / b.detach().double().abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("shape", [(2, 7, 500, 33, 64), (3, 128, 512, 512, 16), (1, 3, 64, 10, 5)])
def test_group_gather_and_backward(shape):
    from usip_amd import ops
    B, C, N, M, K = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, C, N, generator=g).to(DEV)
    idx = torch.randint(0, N, (B, M, K), generator=g).to(DEV)
    sub = torch.randn(B, 3, M, generator=g).to(DEV)
    want = torch.gather(x, 2, idx.view(B, 1, M * K).expand(B, C, M * K)).view(B, C, M, K).clone()
    want[:, 0:3] -= sub.unsqueeze(3)
    got = ops.group_gather(x, idx.int(), sub=sub)
    assert torch.equal(got, want)
    # into a channel slice of a wider tensor
    out = torch.zeros(B, C + 5, M, K, device=DEV)
    ops.group_gather(x, idx.int(), out=out, coff=5)
    assert torch.equal(out[:, 5:], torch.gather(x, 2, idx.view(B, 1, M * K).expand(B, C, M * K)).view(B, C, M, K))
    assert float(out[:, :5].abs().max()) == 0.0
    dout = torch.randn(B, C + 5, M, K, generator=g).to(DEV)
    dx = ops.group_gather_backward(dout, idx.int(), C, N, coff=5)
    want_dx = torch.zeros(B, C, N, device=DEV, dtype=torch.float64).scatter_add_(
        2, idx.view(B, 1, M * K).expand(B, C, M * K), dout[:, 5:].reshape(B, C, M * K).double())
    assert _rel(dx, want_dx) <= 1e-5


@pytest.mark.parametrize("shape", [(2, 64, 100, 64), (2, 256, 64, 16), (1, 5, 7, 5), (1, 3, 9, 32), (1, 2, 3, 100)])
def test_group_max_fwd_bwd(shape):
    from usip_amd import functional as Fh
    B, C, M, K = shape
    g = torch.Generator().manual_seed(sum(shape))
    z = torch.randn(B, C, M, K, generator=g)
    z[:, :, ::2] = torch.relu(z[:, :, ::2])                 # exact ties at 0, as after ReLU
    z = z.to(DEV).requires_grad_(True)
    pooled = Fh.group_max(z)
    want, _ = torch.max(z.detach(), dim=3)
    assert torch.equal(pooled.detach(), want)
    gp = torch.randn(B, C, M, generator=g).to(DEV)
    pooled.backward(gp)
    # one winner per row, it attains the max, and it is the FIRST such k
    nz = (z.grad != 0) | ((gp == 0).unsqueeze(3) & False)
    first = (z.detach() == want.unsqueeze(3)).float().argmax(dim=3)
    want_grad = torch.zeros_like(z).scatter_(3, first.unsqueeze(3), gp.unsqueeze(3))
    assert torch.equal(z.grad, want_grad)
    assert int(nz.sum(dim=3).max()) <= 1


@pytest.mark.parametrize("cfg", [(2, 64, 64, 128, 40, 64, False), (2, 256, 256, 512, 24, 16, True),
                                 (1, 16, 16, 24, 10, 5, True), (2, 8, 12, 20, 6, 32, False)])
def test_pooled_concat_layer_matches_cat_reference(cfg):
    from usip_amd import functional as Fh
    B, Ch, Cp, Cout, M, K, pooled_first = cfg
    g = torch.Generator().manual_seed(sum(cfg[:6]))
    h = torch.randn(B, Ch, M, K, generator=g).to(DEV)
    pooled = torch.randn(B, Cp, M, generator=g).to(DEV)
    w = (torch.randn(Cout, Ch + Cp, generator=g) * (2.0 / (Ch + Cp)) ** 0.5).to(DEV)
    b = (0.1 * torch.randn(Cout, generator=g)).to(DEV)
    gamma = (1 + 0.1 * torch.randn(Cout, generator=g)).to(DEV)
    beta = (0.1 * torch.randn(Cout, generator=g)).to(DEV)
    gy = torch.randn(B, Cout, M, K, generator=g).to(DEV)

    def ref(dtype):
        hh = h.detach().clone().to(dtype).requires_grad_(True)
        pp = pooled.detach().clone().to(dtype).requires_grad_(True)
        ww = w.detach().clone().to(dtype).requires_grad_(True)
        ga = gamma.detach().clone().to(dtype).requires_grad_(True)
        be = beta.detach().clone().to(dtype).requires_grad_(True)
        e = pp.unsqueeze(3).expand(B, Cp, M, K)
        x = torch.cat((e, hh) if pooled_first else (hh, e), dim=1)
        y = torch.einsum("oc,bcmk->bomk", ww, x) + b.to(dtype).view(1, -1, 1, 1)
        y = torch.relu(F.batch_norm(y, None, None, ga, be, True, 0.1, 1e-5))
        y.backward(gy.to(dtype))
        return [y.detach(), hh.grad, pp.grad, ww.grad, ga.grad, be.grad]

    truth, aten = ref(torch.float64), ref(torch.float32)
    hs, ps = h.detach().clone().requires_grad_(True), pooled.detach().clone().requires_grad_(True)
    ws = w.detach().clone().view(Cout, Ch + Cp, 1, 1).requires_grad_(True)
    bs = b.detach().clone().requires_grad_(True)
    bn = torch.nn.BatchNorm2d(Cout).to(DEV).train()
    bn.weight.data.copy_(gamma)
    bn.bias.data.copy_(beta)
    y = Fh.conv1x1_bn_act_pooled(hs, ps, ws, bs, bn, True, pooled_first)
    y.backward(gy)
    got = [y.detach(), hs.grad, ps.grad, ws.grad.view(Cout, Ch + Cp), bn.weight.grad, bn.bias.grad]
    for name, a, t, f32 in zip(["y", "dh", "dpooled", "dw", "dgamma", "dbeta"], got, truth, aten):
        err, aten_err = _rel(a, t), _rel(f32, t)
        assert err <= max(1e-5, 4 * aten_err), (name, err, aten_err)


@pytest.mark.parametrize("cfg", [(2, 128, 128, 40, 64), (2, 512, 512, 24, 16), (1, 20, 12, 6, 32), (1, 8, 8, 5, 5)])
def test_layer_plus_max_fused_matches_reference(cfg):
    """conv1x1 + BN + ReLU + max over K as one node (sparse pooled backward, PRO_BN_BWD_POOL) against plain
    PyTorch in float64 / float32."""
    from usip_amd import functional as Fh
    B, Cin, Cout, M, K = cfg
    g = torch.Generator().manual_seed(sum(cfg))
    x = torch.randn(B, Cin, M, K, generator=g).to(DEV)
    w = (torch.randn(Cout, Cin, generator=g) * (2.0 / Cin) ** 0.5).to(DEV)
    b = (0.1 * torch.randn(Cout, generator=g)).to(DEV)
    gamma = (1 + 0.1 * torch.randn(Cout, generator=g)).to(DEV)
    beta = (0.1 * torch.randn(Cout, generator=g)).to(DEV)
    gp = torch.randn(B, Cout, M, generator=g).to(DEV)

    def ref(dtype):
        xx = x.detach().clone().to(dtype).requires_grad_(True)
        ww = w.detach().clone().to(dtype).requires_grad_(True)
        ga = gamma.detach().clone().to(dtype).requires_grad_(True)
        be = beta.detach().clone().to(dtype).requires_grad_(True)
        y = torch.einsum("oc,bcmk->bomk", ww, xx) + b.to(dtype).view(1, -1, 1, 1)
        z = torch.relu(F.batch_norm(y, None, None, ga, be, True, 0.1, 1e-5))
        p, _ = torch.max(z, dim=3)
        p.backward(gp.to(dtype))
        return [p.detach(), xx.grad, ww.grad, ga.grad, be.grad]

    truth, aten = ref(torch.float64), ref(torch.float32)
    xs = x.detach().clone().requires_grad_(True)
    ws = w.detach().clone().view(Cout, Cin, 1, 1).requires_grad_(True)
    bs = b.detach().clone().requires_grad_(True)
    bn = torch.nn.BatchNorm2d(Cout).to(DEV).train()
    bn.weight.data.copy_(gamma)
    bn.bias.data.copy_(beta)
    p = Fh.conv1x1_bn_relu_max(xs, ws, bs, bn)
    p.backward(gp)
    got = [p.detach(), xs.grad, ws.grad.view(Cout, Cin), bn.weight.grad, bn.bias.grad]
    for name, a, t, f32 in zip(["pooled", "dx", "dw", "dgamma", "dbeta"], got, truth, aten):
        err, aten_err = _rel(a, t), _rel(f32, t)
        assert err <= max(1e-5, 4 * aten_err), (name, err, aten_err)


@pytest.mark.parametrize("K", [16, 64, 5])
def test_pool_fork_gradient_equals_separate_pool_and_pass(K):
    """group_max_fork (pooling gradient added into the consumer's dense gradient in place) must give the
    producing layer bit-for-bit the gradient that separate pooling + autograd's dense sum gives.  K = 5 takes
    the non-fused fallback of the fork."""
    from usip_amd import functional as Fh
    B, Cin, C, M = 2, 9, 64, 40
    g = torch.Generator().manual_seed(K)
    x = torch.randn(B, Cin, M, K, generator=g).to(DEV)
    w1 = (torch.randn(C, Cin, 1, 1, generator=g) * 0.3).to(DEV)
    w2 = (torch.randn(32, 2 * C, 1, 1, generator=g) * 0.1).to(DEV)
    gy = torch.randn(B, 32, M, K, generator=g).to(DEV)

    def run(fork):
        a = w1.clone().requires_grad_(True)
        b = w2.clone().requires_grad_(True)
        bn1, bn2 = torch.nn.BatchNorm2d(C).to(DEV), torch.nn.BatchNorm2d(32).to(DEV)
        xs = x.clone().requires_grad_(True)
        h = Fh.conv1x1_bn_act(xs, a, None, bn1, True, defer=True)
        if fork:
            pooled, h2 = Fh.group_max_fork(h)
        else:
            pooled, h2 = Fh.group_max(h), h
        y = Fh.conv1x1_bn_act_pooled(h2, pooled, b, None, bn2, True, pooled_first=False)
        y.backward(gy)
        return y.detach(), xs.grad, a.grad, b.grad, bn1.weight.grad

    for got, want in zip(run(True), run(False)):
        assert torch.equal(got, want)


def test_group_max_backward_add_inplace():
    from usip_amd import ops
    B, C, M, K = 2, 3, 50, 12
    g = torch.Generator().manual_seed(1)
    dz = torch.randn(B, C, M, K, generator=g).to(DEV)
    dp = torch.randn(B, C, M, generator=g).to(DEV)
    arg = torch.randint(0, K, (B, C, M), generator=g, dtype=torch.int32).to(DEV)
    want = dz + ops.group_max_backward(dp, arg, K)
    out = ops.group_max_backward_add_(dz, dp, arg)
    assert out.data_ptr() == dz.data_ptr() and torch.equal(dz, want)


def test_gather_backward_is_reproducible_bit_for_bit():
    """The scatter-add of the gather backward runs in single-wave workgroups: repeated launches on the same
    input must give the same bits (heavy collisions: 16 neighbours drawn from 40 points)."""
    from usip_amd import ops
    B, C, N, M, K = 4, 33, 40, 512, 16
    g = torch.Generator().manual_seed(3)
    dout = torch.randn(B, C + 3, M, K, generator=g).to(DEV)
    idx = torch.randint(0, N, (B, M, K), generator=g, dtype=torch.int32).to(DEV)
    first = ops.group_gather_backward(dout, idx, C, N, coff=3)
    want = torch.zeros(B, C, N, dtype=torch.float64, device=DEV).scatter_add_(
        2, idx.long().view(B, 1, M * K).expand(B, C, M * K), dout[:, 3:].double().reshape(B, C, M * K))
    assert _rel(first, want) < 1e-6
    for _ in range(20):
        assert torch.equal(ops.group_gather_backward(dout, idx, C, N, coff=3), first)
