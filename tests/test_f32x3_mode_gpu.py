"""The f32x3 mode of the shared-MLP kernels (csrc/shared_mlp_bf16.hip, NS = 3): fp32-accurate products on the
bf16 matrix cores (three bf16 planes per operand, six plane products, fp32 accumulation).

It is held to the fp32 bar, not to a bf16 tolerance: against an fp64 product of the SAME fp32 operands (prologue
evaluated in fp32 first) the error must be at the fp32-MFMA kernel's own level, every prologue and both kernel
families (GEMM, weight gradient), forced onto the split kernel at every shape the tile supports; a whole shared-MLP
layer (forward + backward) and a whole detector step must pass the same checks the fp32 mode passes."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _fma(a, b, c):
    return (a.double() * b.double() + c.double()).float()


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


@pytest.fixture
def x3_forced():
    """f32x3 mode with the split kernel forced wherever its tile applies (default: matrix-bound launches only)."""
    from usip_amd import _lib, ops
    prev = ops.set_matmul_mode("f32x3")
    _lib.lib().usip_set_tuning(b"gemm_split3", 2)
    yield
    _lib.lib().usip_set_tuning(b"gemm_split3", 0)
    ops.set_matmul_mode(prev)


@pytest.fixture
def x3_mode():
    from usip_amd import ops
    prev = ops.set_matmul_mode("f32x3")
    yield
    ops.set_matmul_mode(prev)


GEMM_SHAPES = [  # (nb, K, M, P)
    (2, 128, 128, 1024), (1, 131, 256, 512), (1, 512, 512, 1024), (2, 256, 256, 2048), (2, 40, 130, 333),
    (1, 33, 70, 64), (1, 640, 512, 512), (2, 512, 256, 640),
]


@pytest.mark.parametrize("shape", GEMM_SHAPES)
@pytest.mark.parametrize("pro", [0, 1, 2])
def test_gemm_f32x3_is_fp32_accurate(shape, pro, x3_forced):
    from usip_amd import ops
    nb, K, M, P = shape
    g = torch.Generator().manual_seed(K * 7 + M + P + pro)
    At = (torch.randn(K, M, generator=g) * (2.0 / K) ** 0.5).to(DEV)
    X = torch.randn(nb, K, P, generator=g).to(DEV)
    bias = (0.1 * torch.randn(M, generator=g)).to(DEV)
    coef = X2 = None
    if pro == 1:
        coef = torch.stack([1 + 0.1 * torch.randn(K, generator=g), 0.1 * torch.randn(K, generator=g)]).to(DEV)
        xin = torch.relu(_fma(X, coef[0].view(1, K, 1), coef[1].view(1, K, 1)))
    elif pro == 2:
        X2 = torch.randn(nb, K, P, generator=g).to(DEV)
        coef = torch.stack([1 + 0.1 * torch.randn(K, generator=g), 0.1 * torch.randn(K, generator=g),
                            0.05 * torch.randn(K, generator=g), 0.05 * torch.randn(K, generator=g)]).to(DEV)
        c = [coef[i].view(1, K, 1) for i in range(4)]
        dyh = torch.where(_fma(X2, c[0], c[1]) > 0, X, torch.zeros_like(X))
        xin = _fma(c[0], dyh, _fma(c[2], X2, c[3]))
    else:
        xin = X
    fwd = pro < 2
    Y, stats = ops.mlp_gemm(At, X, bias=bias, want_stats=fwd, pro=pro, X2=X2, coef=coef)
    want = torch.matmul(At.double().t().unsqueeze(0), xin.double()) + bias.double().view(1, M, 1)
    prev = ops.set_matmul_mode("f32")
    Y32, _ = ops.mlp_gemm(At, X, bias=bias, want_stats=fwd, pro=pro, X2=X2, coef=coef)
    ops.set_matmul_mode(prev)
    e3, e32 = _rel(Y, want), _rel(Y32, want)
    assert e3 <= max(5e-7, 2 * e32), (e3, e32)             # at (usually below) the fp32 FMA chain's own error
    assert e3 < 2e-6
    assert not torch.equal(Y, Y32)                         # and it really is another kernel
    if fwd:
        s = stats.double().sum(-1)
        assert _rel(s[1], (Y.double() ** 2).sum((0, 2))) < 1e-5


@pytest.mark.parametrize("shape", [(2, 128, 128, 1024), (1, 256, 131, 512), (1, 512, 512, 1024), (2, 130, 40, 333),
                                   (2, 512, 256, 640), (1, 256, 256, 4096)])
@pytest.mark.parametrize("pro", [0, 2])
@pytest.mark.parametrize("xpro", [False, True])
def test_wgrad_f32x3_is_fp32_accurate(shape, pro, xpro, x3_forced):
    from usip_amd import ops
    nb, M, N, P = shape
    g = torch.Generator().manual_seed(M * 5 + N + P + pro)
    G = torch.randn(nb, M, P, generator=g).to(DEV)
    X = torch.randn(nb, N, P, generator=g).to(DEV)
    G2 = coef4 = xcoef = None
    gin, xin = G, X
    if pro == 2:
        G2 = torch.randn(nb, M, P, generator=g).to(DEV)
        coef4 = torch.stack([1 + 0.1 * torch.randn(M, generator=g), 0.1 * torch.randn(M, generator=g),
                             0.05 * torch.randn(M, generator=g), 0.05 * torch.randn(M, generator=g)]).to(DEV)
        c = [coef4[i].view(1, M, 1) for i in range(4)]
        dyh = torch.where(_fma(G2, c[0], c[1]) > 0, G, torch.zeros_like(G))
        gin = _fma(c[0], dyh, _fma(c[2], G2, c[3]))
    if xpro:
        xcoef = torch.stack([1 + 0.1 * torch.randn(N, generator=g), 0.1 * torch.randn(N, generator=g)]).to(DEV)
        xin = torch.relu(_fma(X, xcoef[0].view(1, N, 1), xcoef[1].view(1, N, 1)))
    dW = ops.mlp_wgrad(G, X, pro=pro, G2=G2, coef4=coef4, xcoef=xcoef)
    want = torch.einsum("bmp,bnp->mn", gin.double(), xin.double())
    prev = ops.set_matmul_mode("f32")
    dW32 = ops.mlp_wgrad(G, X, pro=pro, G2=G2, coef4=coef4, xcoef=xcoef)
    ops.set_matmul_mode(prev)
    e3, e32 = _rel(dW, want), _rel(dW32, want)
    # fp32 class: the fp32 kernel's many short position segments, summed pairwise, make it unusually accurate here
    # (1-2e-7); the split kernel accumulates longer segments and lands at 4-8e-7 -- a few units of fp32 epsilon
    assert e3 <= max(1e-6, 4 * e32), (e3, e32)
    assert e3 < 2e-6
    assert torch.equal(ops.mlp_wgrad(G, X, pro=pro, G2=G2, coef4=coef4, xcoef=xcoef), dW)     # deterministic


def test_split_is_exact_for_hard_values(x3_forced):
    """Operands that stress the three-plane split: powers of two, values with all 24 mantissa bits set, tiny and
    large magnitudes, exact zeros, negative zero, subnormals.  A 1 x K by K x 1 style product with one non-zero
    per row isolates single products: each must come back to within 2^-24 relative (the dropped plane pairs
    contribute at most 3 * 2^-27)."""
    from usip_amd import ops
    K, M, P = 128, 128, 128
    vals = torch.tensor([1.0, -1.0, 3.0, 1.0 + 2.0 ** -23, 2.0 - 2.0 ** -23, 16777215.0, 1e-20, -3.3e15, 0.333333343,
                         2.0 ** -126, 2.0 ** -130, 0.0, -0.0, 7.0e-5, 123456.789, -9.87654321e-3], dtype=torch.float32)
    At = torch.zeros(K, M)
    X = torch.zeros(1, K, P)
    for i in range(K):
        At[i, i] = vals[i % len(vals)]
        X[0, i, :] = vals[(i * 5 + 3) % len(vals)] * (1 + torch.arange(P) * 2.0 ** -12)
    Y, _ = ops.mlp_gemm(At.to(DEV), X.to(DEV))
    want = (At.double().t() @ X[0].double()).unsqueeze(0)
    err = (Y.double().cpu() - want).abs()
    normal = want.abs() >= 2.0 ** -100
    assert bool((err[normal] <= want.abs()[normal] * 2.0 ** -23).all()), float((err[normal] / want.abs()[normal]).max())
    # products below the normal range: the bf16 matrix pipe flushes subnormal partial products, so such a result is
    # exact only to bf16 -- 2^-8 of a number smaller than 1e-30, i.e. zero for every purpose of this path
    assert bool((err[~normal] <= want.abs()[~normal] * 2.0 ** -7 + 2.0 ** -126).all())


def test_layer_forward_backward_in_f32x3_mode(x3_forced):
    """One shared-MLP layer (conv1x1 + BatchNorm + ReLU) forward and backward through the split kernels, against
    fp64 truth, with the fp32 mode's own bar (tests/test_shared_mlp_gpu.py): <= 1e-5, or no worse than 4x ATen."""
    import torch.nn.functional as F
    from usip_amd import functional as Fh
    for (nb, Cin, Cout, P) in [(2, 128, 128, 1024), (1, 512, 512, 1024), (2, 131, 256, 512), (2, 640, 512, 512)]:
        g = torch.Generator().manual_seed(Cin + Cout)
        x = torch.randn(nb, Cin, P, generator=g).to(DEV)
        w = (torch.randn(Cout, Cin, generator=g) * (2.0 / Cin) ** 0.5).to(DEV)
        b = (0.1 * torch.randn(Cout, generator=g)).to(DEV)
        gy = torch.randn(nb, Cout, P, generator=g).to(DEV)
        gamma = (1 + 0.1 * torch.randn(Cout, generator=g)).to(DEV)
        beta = (0.1 * torch.randn(Cout, generator=g)).to(DEV)

        def ref(dtype):
            xs, ws, bs = (t.detach().to(dtype).requires_grad_(True) for t in (x, w, b))
            gs, be = (t.detach().to(dtype).requires_grad_(True) for t in (gamma, beta))
            y = torch.relu(F.batch_norm(torch.matmul(ws, xs) + bs.view(1, -1, 1), None, None, gs, be, True, 0.1, 1e-5))
            y.backward(gy.to(dtype))
            return [y.detach(), xs.grad, ws.grad, gs.grad, be.grad]
        truth, aten = ref(torch.float64), ref(torch.float32)
        xs = x.clone().requires_grad_(True)
        ws = w.clone().view(Cout, Cin, 1).requires_grad_(True)
        bs = b.clone().requires_grad_(True)
        bn = torch.nn.BatchNorm1d(Cout).to(DEV).train()
        bn.weight.data.copy_(gamma)
        bn.bias.data.copy_(beta)
        y = Fh.conv1x1_bn_act(xs, ws, bs, bn, True)
        y.backward(gy)
        got = [y.detach(), xs.grad, ws.grad.view(Cout, Cin), bn.weight.grad, bn.bias.grad]
        for name, a, t, f32 in zip(["y", "dx", "dw", "dgamma", "dbeta"], got, truth, aten):
            err, aten_err = _rel(a, t), _rel(f32, t)
            assert err <= max(1e-5, 4 * aten_err), (name, (nb, Cin, Cout, P), err, aten_err)


@pytest.mark.parametrize("fix", ["detector_ball_micro.npz", "detector_som_cfg1.npz"])
def test_detector_step_in_f32x3_mode_matches_reference(fix, x3_forced):
    """The whole step with every product the 128 x 128 tile supports on the split kernels: forward, losses, BatchNorm
    buffers at 1e-5 and all index tensors bit-exact against the fixture captured from the reference."""
    import test_modules_gpu as tm
    g, st = tm._run_step(fix)
    idx = st.detector.last_indices
    if "idx/ball_idx" in g:
        assert np.array_equal(idx["ball_idx"].cpu().numpy(), g["idx/ball_idx"])
    if "idx/min_idx" in g:
        assert np.array_equal(idx["first_idx"].cpu().numpy(), g["idx/index_max_0"])
        assert np.array_equal(idx["second_idx"].cpu().numpy(), g["idx/index_max_1"])
    assert np.array_equal(idx["knn_I"].cpu().numpy(), g["idx/knn_I"])
    for k in ("keypoints", "sigmas", "loss", "loss_chamfer", "chamfer_pure", "chamfer_weighted"):
        tm.assert_close(st.last[k].detach().cpu().numpy(), g[k], name=k)
    for k, v in st.detector.state_dict().items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            tm.assert_close(v.cpu().numpy(), g["buf/" + k], name=k)


def test_default_f32x3_selection_follows_the_library(x3_mode):
    """Without the forcing knob the library itself decides per launch (matrix-bound shapes only); narrow layers keep
    the fp32 kernel and give bit-identical results in both modes."""
    from usip_amd import _lib, ops
    lib = _lib.lib()
    assert lib.usip_mlp_gemm_f32x3_used(512, 512, 8192, 16) == 1
    assert lib.usip_mlp_gemm_f32x3_used(64, 64, 32768, 16) == 0
    assert lib.usip_mlp_wgrad_f32x3_used(512, 512, 8192, 16) == 1
    At = torch.randn(64, 64, device=DEV)
    X = torch.randn(2, 64, 4096, device=DEV)
    Y3, _ = ops.mlp_gemm(At, X)
    prev = ops.set_matmul_mode("f32")
    Y32, _ = ops.mlp_gemm(At, X)
    ops.set_matmul_mode(prev)
    assert torch.equal(Y3, Y32)
