"""The f32x2 mode of the shared-MLP kernels (csrc/shared_mlp_x3.hip, NPL = 2): fp32-accurate products from TWO fp16
planes per operand and THREE plane products, both operands scaled by exact powers of two derived from rigorous bounds.

Held to the same bar as f32x3 (tests/test_f32x3_mode_gpu.py): against an fp64 product of the SAME fp32 operands
(prologue evaluated in fp32 first) the error must be at the fp32-MFMA kernel's own level -- here additionally over
operand magnitudes from 1e-12 to 1e4 (gradient-like scales), with heavy-tailed operands (the scale comes from a
BOUND, so typical elements sit far below the top of the fp16 range), and for hard values of the split.

The kernels take their operand scale from BatchNorm statistics, so the tests build operands the way a training step
has them: X is a layer's pre-BN output with its true batch statistics, dZ / Y go through usip_bn_backward_reduce_f32."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _fma(a, b, c):
    return (a.double() * b.double() + c.double()).float()


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-300))


@pytest.fixture
def x2_forced():
    from usip_amd import _lib, ops
    prev = ops.set_matmul_mode("f32x2")
    _lib.lib().usip_set_tuning(b"gemm_split3", 2)
    yield
    _lib.lib().usip_set_tuning(b"gemm_split3", 0)
    ops.set_matmul_mode(prev)


def _bn_coef(y, gamma, beta, eps=1e-5):
    """(scale, shift, mean, invstd) of training-mode BatchNorm over y [nb,C,P], as usip_bn_finalize_f32 leaves them."""
    mu = y.double().mean(dim=(0, 2))
    var = y.double().var(dim=(0, 2), unbiased=False)
    istd = (1.0 / torch.sqrt(var + eps)).float()
    sc = gamma * istd
    return torch.stack([sc, beta - mu.float() * sc, mu.float(), istd]).contiguous()


SHAPES = [(2, 128, 128, 1024), (1, 131, 256, 512), (1, 512, 512, 1024), (2, 256, 256, 2048), (2, 40, 130, 333),
          (1, 640, 512, 512), (2, 512, 256, 640)]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("scale", [1.0, 1e-6, 3e3])
@pytest.mark.parametrize("heavy", [False, True])
def test_gemm_forward_f32x2_is_fp32_accurate(shape, scale, heavy, x2_forced):
    """pro = 1: Y = W . relu(bn(X)) + bias with statistics epilogue.  `scale`: magnitude of the pre-BN tensor (the
    BatchNorm brings it back to O(gamma), the weights carry `scale` instead so the PRODUCT spans the range)."""
    from usip_amd import ops
    nb, K, M, P = shape
    g = torch.Generator().manual_seed(K * 7 + M + P)
    At = (torch.randn(K, M, generator=g) * (2.0 / K) ** 0.5 * scale).to(DEV)
    X = torch.randn(nb, K, P, generator=g)
    if heavy:
        X = X * torch.exp(1.5 * torch.randn(nb, K, P, generator=g))
    X = (X * 7.0 + 3.0).to(DEV)
    gamma = (1 + 0.3 * torch.randn(K, generator=g)).to(DEV)
    beta = (0.3 * torch.randn(K, generator=g)).to(DEV)
    bias = (0.1 * scale * torch.randn(M, generator=g)).to(DEV)
    coef = _bn_coef(X, gamma, beta)
    xin = torch.relu(_fma(X, coef[0].view(1, K, 1), coef[1].view(1, K, 1)))
    Y, stats = ops.mlp_gemm(At, X, bias=bias, want_stats=True, pro=1, coef=coef)
    want = torch.matmul(At.double().t().unsqueeze(0), xin.double()) + bias.double().view(1, M, 1)
    prev = ops.set_matmul_mode("f32")
    Y32, _ = ops.mlp_gemm(At, X, bias=bias, want_stats=True, pro=1, coef=coef)
    ops.set_matmul_mode(prev)
    e2, e32 = _rel(Y, want), _rel(Y32, want)
    assert torch.isfinite(Y).all()
    assert e2 <= max(5e-7, 2 * e32), (e2, e32)
    assert e2 < 2e-6
    assert not torch.equal(Y, Y32)                         # it really is the other kernel
    s = stats.double().sum(-1)
    assert _rel(s[1], (Y.double() ** 2).sum((0, 2))) < 1e-5


@pytest.mark.parametrize("cfg", [(2, 64, 128, 2048, 64, False), (1, 64, 128, 1088, 32, False), (3, 128, 128, 640, 64, False),
                                 (2, 128, 128, 4096, 64, True), (16, 64, 128, 4096, 64, True)])
def test_register_resident_gemm_with_row_bias(cfg):
    """gemm_x2r_kernel with a per-neighbourhood row bias (a pooled-concat layer's feature half): values against fp64,
    statistics against the output, ragged last tile (1088 = 17 x 64); small 64-input launches reach it only with the
    split kernels forced (the streaming fp32 kernel keeps them), the step-sized one (16 x 4096 positions) on its own."""
    from usip_amd import _lib, ops
    nb, K, M, P, G, natural = cfg
    g = torch.Generator().manual_seed(K + M + P + G)
    prev = ops.set_matmul_mode("f32x2")
    prev_narrow = ops.NARROW_FWD
    if not natural:
        _lib.lib().usip_set_tuning(b"gemm_split3", 2)
        ops.NARROW_FWD = False
    try:
        At = (torch.randn(K, M, generator=g) * (2.0 / K) ** 0.5).to(DEV)
        X = (torch.randn(nb, K, P, generator=g) * 3.0 + 1.0).to(DEV)
        gamma, beta = (1 + 0.3 * torch.randn(K, generator=g)).to(DEV), (0.3 * torch.randn(K, generator=g)).to(DEV)
        bias = (0.1 * torch.randn(M, generator=g)).to(DEV)
        rb = torch.randn(nb, M, P // G, generator=g).to(DEV)
        coef = _bn_coef(X, gamma, beta)
        xin = torch.relu(_fma(X, coef[0].view(1, K, 1), coef[1].view(1, K, 1)))
        Y, stats = ops.mlp_gemm(At, X, bias=bias, want_stats=True, pro=1, coef=coef, rowbias=rb, rb_group=G)
        want = (torch.matmul(At.double().t().unsqueeze(0), xin.double()) + bias.double().view(1, M, 1)
                + rb.double().repeat_interleave(G, dim=2))
        assert _rel(Y, want) < 2e-6, _rel(Y, want)
        s = stats.double().sum(-1)
        assert _rel(s[0], Y.double().sum((0, 2))) < 1e-5 and _rel(s[1], (Y.double() ** 2).sum((0, 2))) < 1e-5
        if not natural or nb * P >= 65536:                  # it really was the register-resident kernel
            assert stats.shape[2] == _lib.lib().usip_mlp_gemm_x2r_tiles(P, nb)
        Y2, _ = ops.mlp_gemm(At, X, bias=bias, want_stats=True, pro=1, coef=coef, rowbias=rb, rb_group=G)
        assert torch.equal(Y, Y2)
    finally:
        _lib.lib().usip_set_tuning(b"gemm_split3", 0)
        ops.NARROW_FWD = prev_narrow
        ops.set_matmul_mode(prev)


@pytest.mark.parametrize("shape", [s for s in SHAPES if s[1] <= 512])
@pytest.mark.parametrize("gscale", [1.0, 1e-12, 1e-6, 1e4])
def test_gemm_backward_f32x2_is_fp32_accurate(shape, gscale, x2_forced):
    """pro = 2: dX = W^T . dY with dY = BatchNorm'(ReLU'(dZ)) rebuilt from (dZ, Y, coef4); the incoming gradient at
    magnitudes from 1e-12 to 1e4, heavy-tailed (a few entries 1e3 x the typical one)."""
    from usip_amd import ops
    nb, K, M, P = shape
    g = torch.Generator().manual_seed(K * 11 + M + P)
    W = (torch.randn(K, M, generator=g) * (2.0 / K) ** 0.5).to(DEV)          # [Cout = K][Cin = M], the dgrad's K-major operand
    Yp = (torch.randn(nb, K, P, generator=g) * 2.0 + 0.5).to(DEV)            # the layer's pre-BN output
    dZ = torch.randn(nb, K, P, generator=g)
    dZ[torch.rand(nb, K, P, generator=g) < 1e-4] *= 1e3
    dZ = (dZ * gscale).to(DEV)
    gamma = (1 + 0.3 * torch.randn(K, generator=g)).to(DEV)
    beta = (0.3 * torch.randn(K, generator=g)).to(DEV)
    cf = _bn_coef(Yp, gamma, beta)
    dgamma, dbeta, coef4, _ = ops.bn_backward_reduce(dZ, Yp, cf, cf[2].contiguous(), cf[3].contiguous(), gamma, True)
    assert coef4.shape[0] == 5
    c = [coef4[i].view(1, K, 1) for i in range(4)]
    dyh = torch.where(_fma(Yp, c[0], c[1]) > 0, dZ, torch.zeros_like(dZ))
    dy = _fma(c[0], dyh, _fma(c[2], Yp, c[3]))
    assert float(coef4[4, :(K + 63) // 64].max()) >= float(dy.abs().max())    # the bound is a bound
    dX, _ = ops.mlp_gemm(W, dZ, pro=2, X2=Yp, coef=coef4, tag="dgrad")
    want = torch.matmul(W.double().t().unsqueeze(0), dy.double())
    prev = ops.set_matmul_mode("f32")
    dX32, _ = ops.mlp_gemm(W, dZ, pro=2, X2=Yp, coef=coef4[:4].contiguous(), tag="dgrad")
    ops.set_matmul_mode(prev)
    e2, e32 = _rel(dX, want), _rel(dX32, want)
    assert torch.isfinite(dX).all()
    assert e2 <= max(5e-7, 2 * e32), (e2, e32)
    assert e2 < 2e-6
    assert not torch.equal(dX, dX32)


@pytest.mark.parametrize("shape", [(1, 512, 512, 4096), (2, 256, 512, 2048), (1, 256, 256, 8192), (2, 384, 200, 1024)])
@pytest.mark.parametrize("gscale", [1.0, 1e-12, 1e4])
def test_wgrad_f32x2_is_fp32_accurate(shape, gscale, x2_forced):
    """dW = sum_p dY[m][p] act(X)[n][p] with both operands on two fp16 planes (256 x 256 tiles: M, N > 128)."""
    from usip_amd import ops
    nb, M, N, P = shape
    g = torch.Generator().manual_seed(M * 5 + N + P)
    Yp = (torch.randn(nb, M, P, generator=g) * 2.0 + 0.5).to(DEV)
    dZ = (torch.randn(nb, M, P, generator=g) * gscale).to(DEV)
    X = (torch.randn(nb, N, P, generator=g) * 3.0 - 1.0).to(DEV)
    gm, bm = (1 + 0.3 * torch.randn(M, generator=g)).to(DEV), (0.3 * torch.randn(M, generator=g)).to(DEV)
    gn, bn = (1 + 0.3 * torch.randn(N, generator=g)).to(DEV), (0.3 * torch.randn(N, generator=g)).to(DEV)
    cf, xcoef = _bn_coef(Yp, gm, bm), _bn_coef(X, gn, bn)
    _, _, coef4, _ = ops.bn_backward_reduce(dZ, Yp, cf, cf[2].contiguous(), cf[3].contiguous(), gm, True)
    c = [coef4[i].view(1, M, 1) for i in range(4)]
    dyh = torch.where(_fma(Yp, c[0], c[1]) > 0, dZ, torch.zeros_like(dZ))
    gin = _fma(c[0], dyh, _fma(c[2], Yp, c[3]))
    xin = torch.relu(_fma(X, xcoef[0].view(1, N, 1), xcoef[1].view(1, N, 1)))
    dW = ops.mlp_wgrad(dZ, X, pro=2, G2=Yp, coef4=coef4, xcoef=xcoef)
    want = torch.einsum("bmp,bnp->mn", gin.double(), xin.double())
    prev = ops.set_matmul_mode("f32x3")
    dW3 = ops.mlp_wgrad(dZ, X, pro=2, G2=Yp, coef4=coef4, xcoef=xcoef)
    ops.set_matmul_mode("f32")
    dW32 = ops.mlp_wgrad(dZ, X, pro=2, G2=Yp, coef4=coef4[:4].contiguous(), xcoef=xcoef)
    ops.set_matmul_mode(prev)
    e2, e3, e32 = _rel(dW, want), _rel(dW3, want), _rel(dW32, want)
    assert torch.isfinite(dW).all()
    assert e2 <= max(1e-6, 4 * e32, 1.5 * e3), (e2, e3, e32)
    assert e2 < 2e-6
    if M > 128 and N > 128:
        assert not torch.equal(dW, dW3)                    # the two-plane kernel ran
    assert torch.equal(ops.mlp_wgrad(dZ, X, pro=2, G2=Yp, coef4=coef4, xcoef=xcoef), dW)      # deterministic


@pytest.mark.parametrize("shape", [(2, 512, 512, 4096), (3, 256, 256, 8192), (2, 512, 256, 2048), (1, 384, 200, 1000),
                                   (2, 640, 512, 560), (16, 256, 256, 8192), (1, 512, 640, 36)])
@pytest.mark.parametrize("pooled", [False, True])
def test_full_line_wgrad_equals_the_half_line_kernel_bit_for_bit(shape, pooled, x2_forced):
    """Round 5: wgrad_x2l_kernel (every load event = whole 128-B lines: rows [0,128) / [128,256) x 32 positions, two stage
    PAIRS in LDS, one barrier per pair) against round 4's wgrad_x3_kernel<.., 2> (knob r5_forms = 1), which fetched half a
    line per row and stage.  Same stages, same order, same MFMAs: the weight gradient must be the SAME BITS -- plain
    (dZ, Y) and pooled (dpooled, arg, Y) prologues, row / column remainders (384 x 200, 512 x 640), an odd number of
    16-position stages (P = 1000, 560, 36), one and many clouds."""
    from usip_amd import _lib, ops
    nb, M, N, P = shape
    K = 8
    if pooled and P % K:
        P -= P % K
    g = torch.Generator().manual_seed(M * 3 + N + P + int(pooled))
    Yp = (torch.randn(nb, M, P, generator=g) * 2.0 + 0.5).to(DEV)
    X = (torch.randn(nb, N, P, generator=g) * 3.0 - 1.0).to(DEV)
    gm, bm = (1 + 0.3 * torch.randn(M, generator=g)).to(DEV), (0.3 * torch.randn(M, generator=g)).to(DEV)
    gn, bn = (1 + 0.3 * torch.randn(N, generator=g)).to(DEV), (0.3 * torch.randn(N, generator=g)).to(DEV)
    cf, xcoef = _bn_coef(Yp, gm, bm), _bn_coef(X, gn, bn)
    if pooled:
        dp = torch.randn(nb, M, P // K, generator=g).to(DEV)
        arg = torch.randint(0, K, (nb, M, P // K), generator=g, dtype=torch.int32).to(DEV)
        dZ = torch.zeros(nb, M, P // K, K, device=DEV).scatter_(3, arg.long().unsqueeze(-1), dp.unsqueeze(-1)).view(nb, M, P)
        kw = dict(pro=3, G2=Yp, pool=(dp, arg, K))
    else:
        dZ = torch.randn(nb, M, P, generator=g).to(DEV)
        kw = dict(pro=2, G2=Yp)
    _, _, coef4, _ = ops.bn_backward_reduce(dZ, Yp, cf, cf[2].contiguous(), cf[3].contiguous(), gm, True)
    assert coef4.shape[0] == 5
    args = (None, X) if pooled else (dZ, X)
    new = ops.mlp_wgrad(*args, coef4=coef4, xcoef=xcoef, **kw)
    _lib.lib().usip_set_tuning(b"r5_forms", 1)
    try:
        old = ops.mlp_wgrad(*args, coef4=coef4, xcoef=xcoef, **kw)
    finally:
        _lib.lib().usip_set_tuning(b"r5_forms", 0)
    assert torch.isfinite(new).all() and float(new.abs().max()) > 0
    assert torch.equal(new, old)
    # and it is the two-plane kernel that ran (f32x3 gives other bits)
    prev = ops.set_matmul_mode("f32x3")
    try:
        assert not torch.equal(ops.mlp_wgrad(*args, coef4=coef4, xcoef=xcoef, **kw), new)
    finally:
        ops.set_matmul_mode(prev)
    for _ in range(3):
        assert torch.equal(ops.mlp_wgrad(*args, coef4=coef4, xcoef=xcoef, **kw), new)        # run-to-run


def test_two_plane_split_is_exact_for_hard_values(x2_forced):
    """Single products isolated by a diagonal weight matrix: values with all 24 mantissa bits set, powers of two,
    exact zeros, elements 2^-20 below the tensor's largest.  With both operands scaled towards the top of the fp16
    range, an element x within 2^-13 of its tensor's bound keeps 22 bits; what the third plane of f32x3 would add
    is below 2^-22 relative."""
    from usip_amd import ops
    K = M = 128
    P = 256
    vals = torch.tensor([1.0, -1.0, 3.0, 1.0 + 2.0 ** -23, 2.0 - 2.0 ** -23, 0.333333343, 0.0, -0.0, 7.0e-3, 123.456789,
                         -9.87654321e-2, 1.9999, 0.5000001, 2.0 ** -6, -2.0 ** -9, 5.0], dtype=torch.float32)
    At = torch.zeros(K, M)
    for i in range(K):
        At[i, i] = vals[i % len(vals)]
    g = torch.Generator().manual_seed(3)
    X = (torch.randn(1, K, P, generator=g) * 4.0).to(DEV)
    ones, zeros = torch.ones(K, device=DEV), torch.zeros(K, device=DEV)
    coef = _bn_coef(X, ones, zeros)
    xin = torch.relu(_fma(X, coef[0].view(1, K, 1), coef[1].view(1, K, 1)))
    Y, _ = ops.mlp_gemm(At.to(DEV), X, pro=1, coef=coef)
    want = (At.double().t().to(DEV) @ xin[0].double()).unsqueeze(0)
    err = (Y.double() - want).abs()
    big = xin.abs().max().double() * At.abs().max().double()
    # relative to the element itself for elements within 2^-10 of the operand maxima, relative to the scale otherwise
    ok = (err <= want.abs() * 2.0 ** -21 + big * 2.0 ** -34)
    assert bool(ok.all()), float((err / (want.abs() + big * 2.0 ** -13)).max())


def test_layer_forward_backward_in_f32x2_mode(x2_forced):
    """Two stacked shared-MLP layers (conv1x1 + BatchNorm + ReLU, the second consuming the first's lazy activation:
    that is where the two-plane kernels get their operand bounds) forward and backward against fp64 truth, with the
    fp32 mode's own bar: <= 1e-5, or no worse than 4x ATen."""
    import torch.nn.functional as F
    from usip_amd import functional as Fh
    for (nb, C0, C1, C2, P) in [(2, 128, 256, 256, 1024), (1, 512, 512, 512, 1024), (2, 131, 256, 512, 512)]:
        g = torch.Generator().manual_seed(C0 + C1 + C2)
        x = torch.randn(nb, C0, P, generator=g).to(DEV)
        w1 = (torch.randn(C1, C0, generator=g) * (2.0 / C0) ** 0.5).to(DEV)
        w2 = (torch.randn(C2, C1, generator=g) * (2.0 / C1) ** 0.5).to(DEV)
        b1, b2 = (0.1 * torch.randn(C1, generator=g)).to(DEV), (0.1 * torch.randn(C2, generator=g)).to(DEV)
        gy = (torch.randn(nb, C2, P, generator=g) * 1e-5).to(DEV)
        g1, be1 = (1 + 0.1 * torch.randn(C1, generator=g)).to(DEV), (0.1 * torch.randn(C1, generator=g)).to(DEV)
        g2, be2 = (1 + 0.1 * torch.randn(C2, generator=g)).to(DEV), (0.1 * torch.randn(C2, generator=g)).to(DEV)

        def ref(dtype):
            t = [v.detach().to(dtype).requires_grad_(True) for v in (x, w1, b1, g1, be1, w2, b2, g2, be2)]
            h = torch.relu(F.batch_norm(torch.matmul(t[1], t[0]) + t[2].view(1, -1, 1), None, None, t[3], t[4], True, 0.1, 1e-5))
            y = torch.relu(F.batch_norm(torch.matmul(t[5], h) + t[6].view(1, -1, 1), None, None, t[7], t[8], True, 0.1, 1e-5))
            y.backward(gy.to(dtype))
            return [y.detach(), t[0].grad, t[1].grad, t[5].grad, t[3].grad, t[7].grad]
        truth, aten = ref(torch.float64), ref(torch.float32)
        xs = x.clone().requires_grad_(True)
        w1s, w2s = w1.clone().view(C1, C0, 1).requires_grad_(True), w2.clone().view(C2, C1, 1).requires_grad_(True)
        b1s, b2s = b1.clone().requires_grad_(True), b2.clone().requires_grad_(True)
        bn1, bn2 = torch.nn.BatchNorm1d(C1).to(DEV).train(), torch.nn.BatchNorm1d(C2).to(DEV).train()
        bn1.weight.data.copy_(g1); bn1.bias.data.copy_(be1); bn2.weight.data.copy_(g2); bn2.bias.data.copy_(be2)
        h = Fh.conv1x1_bn_act(xs, w1s, b1s, bn1, True, defer=True)
        y = Fh.conv1x1_bn_act(h, w2s, b2s, bn2, True)
        y.backward(gy)
        got = [y.detach(), xs.grad, w1s.grad.view(C1, C0), w2s.grad.view(C2, C1), bn1.weight.grad, bn2.weight.grad]
        for name, a, t, f32 in zip(["y", "dx", "dw1", "dw2", "dgamma1", "dgamma2"], got, truth, aten):
            err, aten_err = _rel(a, t), _rel(f32, t)
            assert err <= max(1e-5, 4 * aten_err), (name, (nb, C0, C1, C2, P), err, aten_err)


def test_f32x2_falls_back_where_no_bound_exists(x2_forced):
    """Launches whose streamed operand has no bound (raw inputs, eval-mode BatchNorm: coef with two rows) run the
    f32x3 kernel in f32x2 mode -- same bits as f32x3 mode."""
    from usip_amd import ops
    K, M, P = 256, 256, 1024
    g = torch.Generator().manual_seed(1)
    At = torch.randn(K, M, generator=g).to(DEV)
    X = torch.randn(2, K, P, generator=g).to(DEV)
    coef2 = torch.stack([1 + 0.1 * torch.randn(K, generator=g), 0.1 * torch.randn(K, generator=g)]).to(DEV)
    y_raw, _ = ops.mlp_gemm(At, X)
    y_eval, _ = ops.mlp_gemm(At, X, pro=1, coef=coef2)
    prev = ops.set_matmul_mode("f32x3")
    assert torch.equal(ops.mlp_gemm(At, X)[0], y_raw)
    assert torch.equal(ops.mlp_gemm(At, X, pro=1, coef=coef2)[0], y_eval)
    ops.set_matmul_mode(prev)


def test_f32x2_falls_back_when_the_statistics_cover_more_samples_than_the_launch(x2_forced, monkeypatch):
    """The two-plane kernels bound |relu(bn(y))| by |gamma| sqrt(n) + |beta| with n = the positions of THEIR launch
    (ADVICE r3): coefficients whose statistics were taken over more samples (here: the launch sees half the tensor)
    do not carry that bound, so the launch must take the three-plane kernel -- same bits as f32x3 mode."""
    from usip_amd import ops
    K, M, P = 256, 256, 2048
    g = torch.Generator().manual_seed(2)
    At = torch.randn(K, M, generator=g).to(DEV)
    X = torch.randn(4, K, P, generator=g).to(DEV)
    mean, var = X.mean((0, 2)), X.var((0, 2), unbiased=False)
    invstd = torch.rsqrt(var + 1e-5)
    coef = torch.stack([invstd, -mean * invstd, mean, invstd]).contiguous()
    assert ops.bound_covers(coef, 4 * P)                    # no recorded count + the tests' opt-in: the launch's own
    monkeypatch.delenv("USIP_ASSUME_LAUNCH_SAMPLES")
    assert not ops.bound_covers(coef, 4 * P)                # the library's default: no count, no bound
    ops.declare_samples(coef, 4 * P)                        # what ops.bn_finalize records (by address: it survives
    assert ops.bound_covers(coef.detach()[:], 4 * P)        # detach(), views and ctx.saved_tensors)
    assert ops.bound_covers(coef, 4 * P) and ops.bound_covers(coef, 8 * P) and not ops.bound_covers(coef, 2 * P)
    half = X[:2].contiguous()
    y_half = ops.mlp_gemm(At, half, pro=1, coef=coef)[0]
    y_full = ops.mlp_gemm(At, X, pro=1, coef=coef)[0]
    prev = ops.set_matmul_mode("f32x3")
    try:
        assert torch.equal(ops.mlp_gemm(At, half, pro=1, coef=coef)[0], y_half)        # fell back
        assert not torch.equal(ops.mlp_gemm(At, X, pro=1, coef=coef)[0], y_full)       # the full launch did not
    finally:
        ops.set_matmul_mode(prev)


def _bn_layer_inputs(g, nb, C, P, scale=1.0):
    """A pre-BatchNorm tensor with its training-mode forward coefficients [4, C] (scale, shift, mean, invstd)."""
    x = (torch.randn(nb, C, P, generator=g) * scale + 0.3 * torch.randn(1, C, 1, generator=g)).to(DEV)
    gamma = (1 + 0.1 * torch.randn(C, generator=g)).to(DEV)
    beta = (0.1 * torch.randn(C, generator=g)).to(DEV)
    mean, var = x.mean((0, 2)), x.var((0, 2), unbiased=False)
    invstd = torch.rsqrt(var + 1e-5)
    coef = torch.stack([gamma * invstd, beta - mean * gamma * invstd, mean, invstd]).contiguous()
    return x, gamma, mean.contiguous(), invstd.contiguous(), coef


@pytest.mark.parametrize("cfg", [(2, 64, 64, 4096, 64, 0, 1.0), (3, 64, 64, 1024, 128, 64, 1.0), (1, 64, 64, 640, 64, 0, 1e-4),
                                 (2, 64, 64, 2048, 64, 0, 1e3), (4, 64, 64, 8192, 64, 0, 1.0), (2, 64, 128, 2048, 128, 64, 1.0),
                                 (1, 64, 128, 640, 128, 0, 1e-3), (3, 64, 128, 1088, 64, 0, 50.0), (4, 64, 128, 8192, 128, 64, 1.0)])
def test_fused_layer_backward_x2_equals_fp64_truth_and_the_separate_products(cfg):
    """csrc/layer_bwd_x2.hip: data gradient, weight gradient and the producing layer's BatchNorm-backward sums of a
    64-input layer from ONE pass over (dZ, Y, X) with f32x2 products against fp64 truth, against the generic kernels,
    and bit for bit on a re-run; gradient magnitudes from 1e-4 to 1e3 (the operand scales come from bounds)."""
    from usip_amd import ops
    nb, Cin, Cout, P, Ctot, wcol, gscale = cfg
    g = torch.Generator().manual_seed(Cin + Cout + P + wcol)
    prev = ops.set_matmul_mode("f32x2")
    try:
        y, gamma_y, mean_y, invstd_y, coef_y = _bn_layer_inputs(g, nb, Cout, P)
        x, gamma_x, mean_x, invstd_x, xcoef = _bn_layer_inputs(g, nb, Cin, P)
        dz = (torch.randn(nb, Cout, P, generator=g) * gscale).to(DEV)
        w2 = (torch.randn(Cout, Ctot, generator=g) * (2.0 / Cin) ** 0.5).to(DEV)
        _, _, coef4, _ = ops.bn_backward_reduce(dz, y, coef_y, mean_y, invstd_y, gamma_y, True)
        assert coef4.shape[0] == 5
        assert ops.layer_backward_x2_supported(Cin, Cout, P, (dz, y, x), coef4, xcoef)
        dw_out = torch.full((Cout, Ctot), 7.0, device=DEV)
        dx, dw, red = ops.mlp_layer_backward_x2(dz, y, coef4, x, xcoef, w2, wcol=wcol, dw_out=dw_out, Cin=Cin, want_red=True)
        c = [coef4[i].double().view(1, Cout, 1) for i in range(4)]
        fma = lambda a_, b_, c_: (a_.double() * b_.double() + c_.double()).float()
        dyh = torch.where(fma(y, c[0], c[1]) > 0, dz, torch.zeros_like(dz))
        dy = fma(c[0], dyh, fma(c[2], y, c[3])).double()
        ax = torch.relu(fma(x, xcoef[0].view(1, Cin, 1), xcoef[1].view(1, Cin, 1))).double()
        wsub = w2[:, wcol:wcol + Cin].double()
        want_dx = torch.einsum("oc,bop->bcp", wsub, dy)
        want_dw = torch.einsum("bop,bcp->oc", dy, ax)
        assert _rel(dx, want_dx) < 2e-6, _rel(dx, want_dx)
        assert _rel(dw[:, wcol:wcol + Cin], want_dw) < 2e-6, _rel(dw[:, wcol:wcol + Cin], want_dw)
        if Ctot > Cin:
            keep = torch.ones(Ctot, dtype=torch.bool)
            keep[wcol:wcol + Cin] = False
            assert bool((dw[:, keep.to(DEV)] == 7.0).all())
        # the producing layer's sums and the maximum, against its own reduction pass over (dX, X)
        sums = red.sums
        on = fma(x, xcoef[0].view(1, Cin, 1), xcoef[1].view(1, Cin, 1)) > 0
        d = torch.where(on, dx, torch.zeros_like(dx)).double()
        xhat = ((x.double() - mean_x.double().view(1, Cin, 1)) * invstd_x.double().view(1, Cin, 1))
        assert _rel(sums[0].double().sum(0), d.sum((0, 2))) < 5e-6
        assert _rel(sums[1].double().sum(0), (d * xhat).sum((0, 2))) < 5e-6
        assert float(red.maxima.max()) == float(d.abs().max().float())
        dg, db, c4 = ops.bn_backward_from_partials(red, nb * P, xcoef, mean_x, invstd_x)
        dg2, db2, c42, _ = ops.bn_backward_reduce(dx, x, xcoef, mean_x, invstd_x, gamma_x, True)
        assert c4.shape[0] == 5 and c42.shape[0] == 5
        for a_, b_, n in ((dg, dg2, "dgamma"), (db, db2, "dbeta"), (c4[:4], c42[:4], "coef4")):
            assert _rel(a_, b_) < 5e-6, (n, _rel(a_, b_))
        assert float(c4[4, 0]) >= float(c42[4, 0]) * (1 - 1e-6) and float(c4[4, 0]) <= 64 * float(c42[4, 0])
        dx3, dw3, red3 = ops.mlp_layer_backward_x2(dz, y, coef4, x, xcoef, w2, wcol=wcol, Cin=Cin, want_red=True,
                                                   dw_out=torch.full((Cout, Ctot), 7.0, device=DEV))
        assert torch.equal(dx3, dx) and torch.equal(dw3, dw) and torch.equal(red3.flat, red.flat)
    finally:
        ops.set_matmul_mode(prev)


@pytest.mark.parametrize("cfg", [(2, 128, 16, 1.0), (1, 64, 64, 1e-3), (3, 32, 16, 30.0), (2, 66, 32, 1.0), (1, 9, 64, 5.0)])
def test_fused_pooled_layer_backward_x2_equals_fp64_truth(cfg):
    """The pooled form (conv5 of the Ball detector: dZ = dpooled at the arg-max neighbour, zero elsewhere, never
    materialised), 128 inputs and 128 outputs; neighbourhoods of 16 (two per tile: per-thread loads of the pooled pair)
    and of 32 / 64 positions (one per tile: the pair handed over through LDS), odd tile counts included."""
    from usip_amd import ops
    nb, M, K, gscale = cfg
    Cin = Cout = 128
    P = M * K
    if P % 64:
        pytest.skip("positions must be a multiple of 64")
    g = torch.Generator().manual_seed(M + K)
    prev = ops.set_matmul_mode("f32x2")
    try:
        y, gamma_y, mean_y, invstd_y, coef_y = _bn_layer_inputs(g, nb, Cout, P)
        x, gamma_x, mean_x, invstd_x, xcoef = _bn_layer_inputs(g, nb, Cin, P)
        w2 = (torch.randn(Cout, Cin, generator=g) * (2.0 / Cin) ** 0.5).to(DEV)
        pooled, arg = ops.group_max_act(y.view(nb, Cout, M, K), coef_y, True)
        dpooled = (torch.randn(nb, Cout, M, generator=g) * gscale).to(DEV)
        _, _, coef4 = ops.bn_pool_backward_reduce(dpooled, arg, y.view(nb, Cout, M, K), coef_y, mean_y, invstd_y, gamma_y, True)
        assert coef4.shape[0] == 5
        assert ops.layer_backward_x2_supported(Cin, Cout, P, (y, x), coef4, xcoef, pooled=True)
        dx, dw = ops.mlp_layer_backward_x2(None, y, coef4, x, xcoef, w2, Cin=Cin, pool=(dpooled, arg, K))
        dz = torch.zeros(nb, Cout, M, K, device=DEV).scatter_(3, arg.long().unsqueeze(3), dpooled.unsqueeze(3)).view(nb, Cout, P)
        c = [coef4[i].double().view(1, Cout, 1) for i in range(4)]
        fma = lambda a_, b_, c_: (a_.double() * b_.double() + c_.double()).float()
        dyh = torch.where(fma(y, c[0], c[1]) > 0, dz, torch.zeros_like(dz))
        dy = fma(c[0], dyh, fma(c[2], y, c[3])).double()
        ax = torch.relu(fma(x, xcoef[0].view(1, Cin, 1), xcoef[1].view(1, Cin, 1))).double()
        want_dx = torch.einsum("oc,bop->bcp", w2.double(), dy)
        want_dw = torch.einsum("bop,bcp->oc", dy, ax)
        assert _rel(dx, want_dx) < 2e-6, _rel(dx, want_dx)
        assert _rel(dw, want_dw) < 2e-6, _rel(dw, want_dw)
        dx3, dw3 = ops.mlp_layer_backward_x2(None, y, coef4, x, xcoef, w2, Cin=Cin, pool=(dpooled, arg, K))
        assert torch.equal(dx3, dx) and torch.equal(dw3, dw)
    finally:
        ops.set_matmul_mode(prev)


@pytest.mark.parametrize("cfg", [(2, 128, 32, 1.0), (1, 64, 64, 1e-3), (3, 33, 64, 30.0), (1, 6, 128, 5.0), (5, 2, 32, 1.0)])
def test_fused_pooled_layer_backward_x2_also_leaves_the_producing_layers_sums(cfg):
    """The pooled form with the sums for the layer that produced X (conv4 of the Ball detector, a pooled-concat layer):
    BatchNorm-backward partials, the maxima for the bound, and the per-neighbourhood sums of dX [relu on] and X, against
    that layer's own stand-alone pass bn_backward_reduce(group=K) over the dX the kernel wrote; dX / dW unchanged by the
    extra outputs; odd tile and neighbourhood counts per workgroup."""
    from usip_amd import ops
    nb, M, K, gscale = cfg
    Cin = Cout = 128
    P = M * K
    g = torch.Generator().manual_seed(7 * M + K)
    prev = ops.set_matmul_mode("f32x2")
    try:
        y, gamma_y, mean_y, invstd_y, coef_y = _bn_layer_inputs(g, nb, Cout, P)
        x, gamma_x, mean_x, invstd_x, xcoef = _bn_layer_inputs(g, nb, Cin, P)
        w2 = (torch.randn(Cout, Cin, generator=g) * (2.0 / Cin) ** 0.5).to(DEV)
        pooled, arg = ops.group_max_act(y.view(nb, Cout, M, K), coef_y, True)
        dpooled = (torch.randn(nb, Cout, M, generator=g) * gscale).to(DEV)
        _, _, coef4 = ops.bn_pool_backward_reduce(dpooled, arg, y.view(nb, Cout, M, K), coef_y, mean_y, invstd_y, gamma_y, True)
        dx0, dw0 = ops.mlp_layer_backward_x2(None, y, coef4, x, xcoef, w2, Cin=Cin, pool=(dpooled, arg, K))
        dx, dw, red = ops.mlp_layer_backward_x2(None, y, coef4, x, xcoef, w2, Cin=Cin, pool=(dpooled, arg, K),
                                                want_red=True, want_gsum=True)
        assert _rel(dx, dx0) < 1e-6 and _rel(dw, dw0) < 1e-6
        dg2, db2, c42, gsum2 = ops.bn_backward_reduce(dx, x, xcoef, mean_x, invstd_x, gamma_x, True, group=K)
        assert red.gsum.shape == gsum2.shape == (2, nb, Cin, M)
        on = (x.double() * xcoef[0].double().view(1, Cin, 1) + xcoef[1].double().view(1, Cin, 1)).float() > 0
        d = torch.where(on, dx, torch.zeros_like(dx)).double()
        want_g0 = d.view(nb, Cin, M, K).sum(3)
        want_g1 = x.double().view(nb, Cin, M, K).sum(3)
        assert _rel(red.gsum[0], want_g0) < 2e-6, _rel(red.gsum[0], want_g0)
        assert _rel(red.gsum[1], want_g1) < 2e-6, _rel(red.gsum[1], want_g1)
        assert _rel(red.gsum[0], gsum2[0]) < 2e-6 and _rel(red.gsum[1], gsum2[1]) < 2e-6
        assert float(red.maxima.max()) == float(d.abs().max().float())
        dg, db, c4 = ops.bn_backward_from_partials(red, nb * P, xcoef, mean_x, invstd_x)
        assert c4.shape[0] == 5 and c42.shape[0] == 5
        for a_, b_, n in ((dg, dg2, "dgamma"), (db, db2, "dbeta"), (c4[:4], c42[:4], "coef4")):
            assert _rel(a_, b_) < 5e-6, (n, _rel(a_, b_))
        nbe = Cin // 64                                   # row 4 holds one bound per 64 channels
        assert bool((c4[4, :nbe] >= c42[4, :nbe] * (1 - 1e-6)).all()) and bool((c4[4, :nbe] <= 64 * c42[4, :nbe]).all())
        dx3, dw3, red3 = ops.mlp_layer_backward_x2(None, y, coef4, x, xcoef, w2, Cin=Cin, pool=(dpooled, arg, K),
                                                   want_red=True, want_gsum=True)
        assert torch.equal(dx3, dx) and torch.equal(dw3, dw) and torch.equal(red3.flat, red.flat)
        assert torch.equal(red3.gsum, red.gsum)
    finally:
        ops.set_matmul_mode(prev)


def test_fused_layer_backward_is_bit_stable_over_repeated_launches():
    """40 launches of every fused-backward form on the same inputs, at sizes that put two workgroups on every CU and
    several tiles on every workgroup, must agree in every bit and with the fp64 truth.  (Built with hipcc's SLP vectoriser
    the 64 -> 128 form passed the small fp64-truth tests and still produced 1-4 slightly wrong tiles of 1024 on most
    launches at these sizes -- a packed fp32 FMA with swapped halves going wrong with two waves on a SIMD; the library
    is compiled with -fno-slp-vectorize since: usip_amd/build.py.)"""
    from usip_amd import ops
    g = torch.Generator().manual_seed(5)
    prev = ops.set_matmul_mode("f32x2")
    try:
        nb, P = 8, 8192
        for Cout in (64, 128):
            y, gamma_y, mean_y, invstd_y, coef_y = _bn_layer_inputs(g, nb, Cout, P)
            x, _, _, _, xcoef = _bn_layer_inputs(g, nb, 64, P)
            dz = torch.randn(nb, Cout, P, generator=g).to(DEV)
            w2 = (torch.randn(Cout, 128, generator=g) * 0.18).to(DEV)
            coef4 = ops.bn_backward_reduce(dz, y, coef_y, mean_y, invstd_y, gamma_y, True)[2]
            c = [coef4[i].double().view(1, Cout, 1) for i in range(4)]
            fma = lambda a_, b_, c_: (a_.double() * b_.double() + c_.double()).float()
            dyh = torch.where(fma(y, c[0], c[1]) > 0, dz, torch.zeros_like(dz))
            dy = fma(c[0], dyh, fma(c[2], y, c[3])).double()
            want_dx = torch.einsum("oc,bop->bcp", w2[:, 64:].double(), dy)
            for want_red in (True, False):
                ref = None
                for _ in range(40):
                    res = ops.mlp_layer_backward_x2(dz, y, coef4, x, xcoef, w2, wcol=64, Cin=64, want_red=want_red,
                                                    dw_out=torch.zeros(Cout, 128, device=DEV))
                    cur = (res[0], res[1]) + ((res[2].flat,) if want_red else ())
                    ref = ref or tuple(t.clone() for t in cur)
                    assert all(torch.equal(a, b) for a, b in zip(ref, cur))
                assert _rel(ref[0], want_dx) < 2e-6
                assert float((ref[0].double() - want_dx).abs().max()) < 1e-5 * float(want_dx.abs().max())   # no tile off
        M, K = 128, 64
        y, gamma_y, mean_y, invstd_y, coef_y = _bn_layer_inputs(g, nb, 128, M * K)
        x, _, _, _, xcoef = _bn_layer_inputs(g, nb, 128, M * K)
        w2 = (torch.randn(128, 128, generator=g) * 0.12).to(DEV)
        pooled, arg = ops.group_max_act(y.view(nb, 128, M, K), coef_y, True)
        dpooled = torch.randn(nb, 128, M, generator=g).to(DEV)
        coef4 = ops.bn_pool_backward_reduce(dpooled, arg, y.view(nb, 128, M, K), coef_y, mean_y, invstd_y, gamma_y, True)[2]
        for red in (True, False):
            ref = None
            for _ in range(40):
                res = ops.mlp_layer_backward_x2(None, y, coef4, x, xcoef, w2, Cin=128, pool=(dpooled, arg, K), want_red=red,
                                                want_gsum=red)
                cur = (res[0], res[1]) + ((res[2].flat, res[2].gsum) if red else ())
                ref = ref or tuple(t.clone() for t in cur)
                assert all(torch.equal(a, b) for a, b in zip(ref, cur))
    finally:
        ops.set_matmul_mode(prev)


def test_streaming_and_split_kernels_are_bit_stable_over_repeated_launches():
    """The same guard for the other kernels of the Ball front end and the second stage at sizes that fill the chip
    (two workgroups per CU, several tiles each): 30 launches of each on the same inputs agree in every bit."""
    from usip_amd import ops
    g = torch.Generator().manual_seed(9)
    prev = ops.set_matmul_mode("f32x2")

    def stable(fn, n=30):
        ref = None
        for _ in range(n):
            cur = [t for t in fn() if isinstance(t, torch.Tensor)]
            ref = ref or [t.clone() for t in cur]
            assert all(torch.equal(a, b) for a, b in zip(ref, cur))

    try:
        nb, P, G = 16, 4096, 64
        for K, M in ((64, 64), (64, 128), (128, 128)):                      # narrow_fwd / gemm_x2r (row bias) / gemm_x2r
            At = (torch.randn(K, M, generator=g) * (2.0 / K) ** 0.5).to(DEV)
            X = (torch.randn(nb, K, P, generator=g) * 3.0 + 1.0).to(DEV)
            coef = _bn_coef(X, (1 + 0.3 * torch.randn(K, generator=g)).to(DEV), (0.3 * torch.randn(K, generator=g)).to(DEV))
            bias = (0.1 * torch.randn(M, generator=g)).to(DEV)
            rb = torch.randn(nb, M, P // G, generator=g).to(DEV) if (K, M) == (64, 128) else None
            stable(lambda: ops.mlp_gemm(At, X, bias=bias, want_stats=True, pro=1, coef=coef, rowbias=rb, rb_group=G if rb is not None else 0))
        # the 64 -> 128 backward that stays on the fp32-MFMA kernel, and a second-stage layer: forward, data and weight gradient
        y, gamma_y, mean_y, invstd_y, coef_y = _bn_layer_inputs(g, nb, 128, P)
        x, _, _, _, xcoef = _bn_layer_inputs(g, nb, 64, P)
        dz = torch.randn(nb, 128, P, generator=g).to(DEV)
        w2 = (torch.randn(128, 64, generator=g) * 0.18).to(DEV)
        coef4 = ops.bn_backward_reduce(dz, y, coef_y, mean_y, invstd_y, gamma_y, True)[2]
        stable(lambda: (lambda r: (r[0], r[1], r[2].flat))(ops.mlp_narrow_backward(dz, y, coef4, x, xcoef, w2, want_red=True)))
        nb2, P2, C = 4, 8192, 256
        y, gamma_y, mean_y, invstd_y, coef_y = _bn_layer_inputs(g, nb2, C, P2)
        x, _, _, _, xcoef = _bn_layer_inputs(g, nb2, C, P2)
        dz = torch.randn(nb2, C, P2, generator=g).to(DEV)
        w2 = (torch.randn(C, C, generator=g) * 0.09).to(DEV)
        coef4 = ops.bn_backward_reduce(dz, y, coef_y, mean_y, invstd_y, gamma_y, True)[2]
        wt = w2.t().contiguous()
        stable(lambda: ops.mlp_gemm(wt, x, want_stats=True, pro=1, coef=xcoef), 15)
        stable(lambda: ops.mlp_gemm(w2, dz, pro=2, X2=y, coef=coef4, tag="dgrad"), 15)
        stable(lambda: (ops.mlp_wgrad(dz, x, pro=2, G2=y, coef4=coef4, xcoef=xcoef),), 15)
    finally:
        ops.set_matmul_mode(prev)


# ---------------------------------------------------------------------------------------------------------------------
# Round 4: csrc/gemm_x2d.hip -- the same products with the streamed operand going global -> registers -> MFMA.  It takes
# the launches with 256-row tiles (>= 256 tiles of 256 x 128), which the shapes above are too small for: these shapes
# fill the chip.  Cases: K with an even / odd number of 16-k stages (two / one stage of loads in flight), a K tail
# (K % 16 != 0), a partial row tile (M = 300), positions that are not a multiple of the tile (P = 8200) or of 4.
DIRECT_SHAPES = [(4, 256, 256, 8192), (2, 512, 512, 8192), (4, 272, 256, 8192), (4, 136, 256, 8192), (2, 640, 512, 8192),
                 (3, 256, 300, 8200), (4, 256, 256, 8191)]


def _direct(shape):
    from usip_amd import _lib
    nb, K, M, P = shape
    return _lib.lib().usip_mlp_x3p_tile_rows(M, P, nb) == 256


@pytest.mark.parametrize("shape", DIRECT_SHAPES)
@pytest.mark.parametrize("heavy", [False, True])
def test_direct_gemm_forward_is_fp32_accurate(shape, heavy, x2_forced):
    from usip_amd import _lib, ops
    assert _direct(shape)
    nb, K, M, P = shape
    g = torch.Generator().manual_seed(K * 7 + M + P)
    At = (torch.randn(K, M, generator=g) * (2.0 / K) ** 0.5).to(DEV)
    X = torch.randn(nb, K, P, generator=g)
    if heavy:
        X = X * torch.exp(1.5 * torch.randn(nb, K, P, generator=g))
    X = (X * 7.0 + 3.0).to(DEV)
    gamma = (1 + 0.3 * torch.randn(K, generator=g)).to(DEV)
    beta = (0.3 * torch.randn(K, generator=g)).to(DEV)
    bias = (0.1 * torch.randn(M, generator=g)).to(DEV)
    coef = _bn_coef(X, gamma, beta)
    xin = torch.relu(_fma(X, coef[0].view(1, K, 1), coef[1].view(1, K, 1)))
    Y, stats = ops.mlp_gemm(At, X, bias=bias, want_stats=True, pro=1, coef=coef)
    want = torch.matmul(At.double().t().unsqueeze(0), xin.double()) + bias.double().view(1, M, 1)
    _lib.lib().usip_set_tuning(b"x2_direct", 1)            # the LDS-staged kernel of round 3 on the same operands
    try:
        Yold, _ = ops.mlp_gemm(At, X, bias=bias, want_stats=True, pro=1, coef=coef)
    finally:
        _lib.lib().usip_set_tuning(b"x2_direct", 0)
    prev = ops.set_matmul_mode("f32")
    Y32, _ = ops.mlp_gemm(At, X, bias=bias, want_stats=True, pro=1, coef=coef)
    ops.set_matmul_mode(prev)
    e2, e32 = _rel(Y, want), _rel(Y32, want)
    assert torch.isfinite(Y).all()
    assert e2 <= max(5e-7, 2 * e32), (e2, e32)
    assert e2 < 2e-6
    assert _rel(Y, Yold) < 1e-6
    s = stats.double().sum(-1)
    assert _rel(s[0], Y.double().sum((0, 2))) < 1e-5 and _rel(s[1], (Y.double() ** 2).sum((0, 2))) < 1e-5
    for _ in range(3):                                     # bit-stable over repeated launches (counted waits, no races)
        assert torch.equal(ops.mlp_gemm(At, X, bias=bias, want_stats=True, pro=1, coef=coef)[0], Y)


@pytest.mark.parametrize("pooled", [False, True])
def test_data_gradient_stores_straight_from_the_registers_at_chip_filling_size(pooled, x2_forced):
    """The DIRECT epilogue of csrc/gemm_x2d.hip (data gradients: unswapped MFMA operands, 128 dword stores per wave and
    tile, `s_nop 7` behind each group) at the size where the store hazard of round 4 showed -- 2048 tiles, two persistent
    workgroups per CU: 20 launches agree in every bit, and with the LDS-transposing epilogue (knob x2_direct = 8)."""
    from usip_amd import _lib, ops
    nb, C, P, K = 16, 512, 8192, 16
    g = torch.Generator().manual_seed(21)
    y, gamma_y, mean_y, invstd_y, coef_y = _bn_layer_inputs(g, nb, C, P)
    w2 = (torch.randn(C, C, generator=g) * 0.06).to(DEV)
    ops.PLANES_CACHE = {}
    try:
        if pooled:
            M = P // K
            pooled_v, arg = ops.group_max_act(y.view(nb, C, M, K), coef_y, True)
            dpooled = torch.randn(nb, C, M, generator=g).to(DEV)
            coef4 = ops.bn_pool_backward_reduce(dpooled, arg, y.view(nb, C, M, K), coef_y, mean_y, invstd_y, gamma_y, True)[2]
            run = lambda: ops.mlp_gemm(w2, None, pro=3, X2=y, coef=coef4, tag="dgrad", pool=(dpooled, arg, K))[0]
        else:
            dz = torch.randn(nb, C, P, generator=g).to(DEV)
            coef4 = ops.bn_backward_reduce(dz, y, coef_y, mean_y, invstd_y, gamma_y, True)[2]
            run = lambda: ops.mlp_gemm(w2, dz, pro=2, X2=y, coef=coef4, tag="dgrad")[0]
        ref = run().clone()
        for _ in range(20):
            assert torch.equal(run(), ref)
        _lib.lib().usip_set_tuning(b"x2_direct", 8)
        other = run()
        _lib.lib().usip_set_tuning(b"x2_direct", 0)
        assert torch.equal(other, ref)
        assert bool(torch.isfinite(ref).all()) and float(ref.abs().max()) > 0
    finally:
        _lib.lib().usip_set_tuning(b"x2_direct", 0)
        ops.PLANES_CACHE = None


@pytest.mark.parametrize("shape", [(4, 256, 256, 8192), (2, 512, 512, 8192), (8, 256, 512, 4096), (4, 64, 256, 8192)])
@pytest.mark.parametrize("stats", [False, True])
def test_all_dma_forward_gemm_equals_the_direct_one_bit_for_bit(shape, stats, x2_forced):
    """csrc/gemm_x2e.hip (both operands by LDS-DMA, one 8-wave workgroup per CU; opt-in, knob x2_direct = 10) computes
    what csrc/gemm_x2d.hip computes, in the same order: outputs bit-identical, statistics equal after the sum over the
    tiles (its 256-position tile fills the first of two 128-position slots), with and without a row bias."""
    from usip_amd import _lib, ops
    nb, K, M, P = shape
    g = torch.Generator().manual_seed(11)
    At = (torch.randn(K, M, generator=g) * (2.0 / K) ** 0.5).to(DEV)
    X = (torch.randn(nb, K, P, generator=g) * 1.7 + 0.2).to(DEV)
    bias = torch.randn(M, generator=g).to(DEV)
    mu, var = X.mean(dim=(0, 2)), X.var(dim=(0, 2), unbiased=False)
    istd = torch.rsqrt(var + 1e-5)
    coef = torch.stack([istd, -mu * istd, mu, istd]).contiguous()
    rowbias = torch.randn(nb, M, P // 64, generator=g).to(DEV)
    ops.PLANES_CACHE = {}
    try:
        for rb in (None, rowbias):
            kw = dict(want_stats=stats, pro=1, coef=coef, rowbias=rb, rb_group=64 if rb is not None else 1)
            y0, s0 = ops.mlp_gemm(At, X, bias, **kw)
            _lib.lib().usip_set_tuning(b"x2_direct", 10)
            y1, s1 = ops.mlp_gemm(At, X, bias, **kw)
            _lib.lib().usip_set_tuning(b"x2_direct", 0)
            assert torch.equal(y0, y1)
            if stats:
                assert not torch.equal(s0, s1)                  # (it did run: its second slots are zero)
                t0, t1 = s0.double().sum(dim=2), s1.double().sum(dim=2)
                assert float((t0 - t1).abs().max() / t0.abs().max()) < 1e-6
    finally:
        _lib.lib().usip_set_tuning(b"x2_direct", 0)
        ops.PLANES_CACHE = None


X2F_SHAPES = [(4, 256, 256, 8192), (16, 256, 512, 8192), (16, 512, 512, 4096), (2, 640, 512, 8192), (8, 128, 256, 8192)]


@pytest.mark.parametrize("shape", X2F_SHAPES)
def test_one_wave_per_simd_gemm_equals_the_direct_one_bit_for_bit(shape, x2_forced):
    """Round 6, csrc/gemm_x2f.hip (one wave per SIMD, 256 x 256 tiles, a wave = 64 positions as even / odd blocks loaded with
    dwordx2, next tile requested before the epilogue; the default wherever it fits) computes what csrc/gemm_x2d.hip computes
    (knob x2_direct = 12), in the same order per accumulator: forward outputs bit-identical with bias, with and without a row
    bias and statistics (statistics equal per 128-position slot up to fp32 summation order); data gradients (pro 2, and
    pro 3 = the pooled form) bit-identical; persistent workgroups with 1..4 tiles each, whose software pipeline
    runs across the tile boundary; repeated launches bit-stable."""
    from usip_amd import _lib, ops
    nb, K, M, P = shape
    lib = _lib.lib()
    assert lib.usip_mlp_x3p_tile_rows(M, P, nb) == 256
    assert lib.usip_mlp_gemm_x2f_used(M, K, P, nb, 1, 1, 1, 0, 0, 0) == 1
    g = torch.Generator().manual_seed(K * 5 + M + P + nb)
    At = (torch.randn(K, M, generator=g) * (2.0 / K) ** 0.5).to(DEV)
    X = (torch.randn(nb, K, P, generator=g) * 1.7 + 0.2).to(DEV)
    bias = torch.randn(M, generator=g).to(DEV)
    gamma = (1 + 0.3 * torch.randn(K, generator=g)).to(DEV)
    beta = (0.3 * torch.randn(K, generator=g)).to(DEV)
    coef = _bn_coef(X, gamma, beta)
    rowbias = torch.randn(nb, M, P // 16, generator=g).to(DEV)
    ops.PLANES_CACHE = {}
    try:
        for rb in (None, rowbias):
            for stats in (True, False):
                kw = dict(want_stats=stats, pro=1, coef=coef, rowbias=rb, rb_group=16 if rb is not None else 1)
                y1, s1 = ops.mlp_gemm(At, X, bias, **kw)
                lib.usip_set_tuning(b"x2_direct", 12)
                y0, s0 = ops.mlp_gemm(At, X, bias, **kw)
                lib.usip_set_tuning(b"x2_direct", 0)
                assert torch.equal(y0, y1), (rb is not None, stats, float((y0 - y1).abs().max()))
                if stats:
                    assert s0.shape == s1.shape
                    assert _rel(s1[0], s0[0]) < 2e-6 and _rel(s1[1], s0[1]) < 2e-6
                    assert _rel(s1.double().sum(-1)[0], y1.double().sum((0, 2))) < 1e-5
                for _ in range(2):
                    assert torch.equal(ops.mlp_gemm(At, X, bias, **kw)[0], y1)
        if K <= 512:
            W = (torch.randn(K, M, generator=g) * (2.0 / K) ** 0.5).to(DEV)
            Yp = (torch.randn(nb, K, P, generator=g) * 2.0 + 0.5).to(DEV)
            dZ = torch.randn(nb, K, P, generator=g).to(DEV)
            cfy = _bn_coef(Yp, gamma, beta)
            coef4 = ops.bn_backward_reduce(dZ, Yp, cfy, cfy[2].contiguous(), cfy[3].contiguous(), gamma, True)[2]
            G = 16
            dp = torch.randn(nb, K, P // G, generator=g).to(DEV)
            arg = torch.randint(0, G, (nb, K, P // G), generator=g, dtype=torch.int32).to(DEV)
            c4p = ops.bn_pool_backward_reduce(dp, arg, Yp.view(nb, K, P // G, G), cfy, cfy[2].contiguous(), cfy[3].contiguous(),
                                              gamma, True)[2]
            runs = (lambda: ops.mlp_gemm(W, dZ, pro=2, X2=Yp, coef=coef4, tag="dgrad")[0],
                    lambda: ops.mlp_gemm(W, None, pro=3, X2=Yp, coef=c4p, tag="dgrad", pool=(dp, arg, G))[0])
            for run in runs:
                d1 = run()
                lib.usip_set_tuning(b"x2_direct", 12)
                d0 = run()
                lib.usip_set_tuning(b"x2_direct", 0)
                assert torch.equal(d0, d1), float((d0 - d1).abs().max())
                assert bool(torch.isfinite(d1).all()) and float(d1.abs().max()) > 0
                for _ in range(3):
                    assert torch.equal(run(), d1)
    finally:
        lib.usip_set_tuning(b"x2_direct", 0)
        ops.PLANES_CACHE = None


@pytest.mark.parametrize("shape", [s for s in DIRECT_SHAPES if s[1] <= 512])
@pytest.mark.parametrize("gscale", [1.0, 1e-6])
def test_direct_gemm_backward_is_fp32_accurate(shape, gscale, x2_forced):
    """pro = 2 and pro = 3 (the gradient synthesised from a max-pool's (dpooled, arg) pair) through the direct kernel."""
    from usip_amd import _lib, ops
    assert _direct(shape)
    nb, K, M, P = shape
    g = torch.Generator().manual_seed(K * 11 + M + P)
    W = (torch.randn(K, M, generator=g) * (2.0 / K) ** 0.5).to(DEV)
    Yp = (torch.randn(nb, K, P, generator=g) * 2.0 + 0.5).to(DEV)
    dZ = torch.randn(nb, K, P, generator=g)
    dZ[torch.rand(nb, K, P, generator=g) < 1e-4] *= 1e3
    dZ = (dZ * gscale).to(DEV)
    gamma = (1 + 0.3 * torch.randn(K, generator=g)).to(DEV)
    beta = (0.3 * torch.randn(K, generator=g)).to(DEV)
    cf = _bn_coef(Yp, gamma, beta)
    _, _, coef4, _ = ops.bn_backward_reduce(dZ, Yp, cf, cf[2].contiguous(), cf[3].contiguous(), gamma, True)
    c = [coef4[i].view(1, K, 1) for i in range(4)]
    dyh = torch.where(_fma(Yp, c[0], c[1]) > 0, dZ, torch.zeros_like(dZ))
    dy = _fma(c[0], dyh, _fma(c[2], Yp, c[3]))
    dX, _ = ops.mlp_gemm(W, dZ, pro=2, X2=Yp, coef=coef4, tag="dgrad")
    want = torch.matmul(W.double().t().unsqueeze(0), dy.double())
    prev = ops.set_matmul_mode("f32")
    dX32, _ = ops.mlp_gemm(W, dZ, pro=2, X2=Yp, coef=coef4[:4].contiguous(), tag="dgrad")
    ops.set_matmul_mode(prev)
    e2, e32 = _rel(dX, want), _rel(dX32, want)
    assert torch.isfinite(dX).all()
    assert e2 <= max(5e-7, 2 * e32), (e2, e32)
    assert e2 < 2e-6
    for _ in range(3):
        assert torch.equal(ops.mlp_gemm(W, dZ, pro=2, X2=Yp, coef=coef4, tag="dgrad")[0], dX)
    if P % 16 == 0:
        # pro = 3: dZ[c][m][k] = (k == arg[c][m]) ? dpooled[c][m] : 0 over groups of 16 positions; the same coefficients
        # applied to that dense tensor give the same arithmetic, so the two launches must agree bit for bit
        G = 16
        dp = (torch.randn(nb, K, P // G, generator=g) * gscale).to(DEV)
        arg = torch.randint(0, G, (nb, K, P // G), generator=g, dtype=torch.int32).to(DEV)
        dense = torch.zeros(nb, K, P // G, G, device=DEV).scatter_(3, arg.long().unsqueeze(3), dp.unsqueeze(3)).view(nb, K, P)
        a = ops.mlp_gemm(W, None, pro=3, X2=Yp, coef=coef4, tag="dgrad", pool=(dp, arg, G))[0]
        b = ops.mlp_gemm(W, dense, pro=2, X2=Yp, coef=coef4, tag="dgrad")[0]
        assert torch.equal(a, b)


@pytest.mark.parametrize("shape,group", [((4, 256, 256, 8192), 0), ((2, 512, 512, 8192), 16), ((8, 512, 256, 4096), 32)])
@pytest.mark.parametrize("pooled", [False, True])
def test_direct_gemm_backward_leaves_the_producing_layers_bn_sums(shape, group, pooled, x2_forced):
    """Round 4 (an option of the product, off by default because the step measured slower with it --
    usip_amd/functional.py GEMM_RED): the data-gradient launches of the direct kernel leave, from the dX tile they hold,
    the BatchNorm-backward partial sums (and, for a pooled-concat producer, the per-neighbourhood sums) of the layer that
    produced the activation -- against float64 sums over the dX the SAME launch wrote, and against the stand-alone
    reduction's coefficients."""
    from usip_amd import _lib, ops
    nb, K, M, P = shape
    assert _lib.lib().usip_mlp_gemm_x2d_red_tiles(M, K, P, nb, group) == nb * (P // 128)
    g = torch.Generator().manual_seed(K * 13 + M + P + group)
    W = (torch.randn(K, M, generator=g) * (2.0 / K) ** 0.5).to(DEV)
    Yp = (torch.randn(nb, K, P, generator=g) * 2.0 + 0.5).to(DEV)            # this layer's pre-BN output
    gamma = (1 + 0.3 * torch.randn(K, generator=g)).to(DEV)
    beta = (0.3 * torch.randn(K, generator=g)).to(DEV)
    cf = _bn_coef(Yp, gamma, beta)
    Yprev = (torch.randn(nb, M, P, generator=g) * 1.5 - 0.2).to(DEV)         # the producing layer's pre-BN output
    cprev = _bn_coef(Yprev, (1 + 0.3 * torch.randn(M, generator=g)).to(DEV), (0.3 * torch.randn(M, generator=g)).to(DEV))
    if pooled:
        G = 16
        dp = torch.randn(nb, K, P // G, generator=g).to(DEV)
        arg = torch.randint(0, G, (nb, K, P // G), generator=g, dtype=torch.int32).to(DEV)
        _, _, coef4 = ops.bn_pool_backward_reduce(dp, arg, Yp.view(nb, K, P // G, G), cf, cf[2].contiguous(), cf[3].contiguous(),
                                                  gamma, True)
        dX, _, red = ops.mlp_gemm(W, None, pro=3, X2=Yp, coef=coef4, tag="dgrad", pool=(dp, arg, G), red=(Yprev, cprev),
                                  red_group=group)
        plain = ops.mlp_gemm(W, None, pro=3, X2=Yp, coef=coef4, tag="dgrad", pool=(dp, arg, G))[0]
    else:
        dZ = torch.randn(nb, K, P, generator=g).to(DEV)
        _, _, coef4, _ = ops.bn_backward_reduce(dZ, Yp, cf, cf[2].contiguous(), cf[3].contiguous(), gamma, True)
        dX, _, red = ops.mlp_gemm(W, dZ, pro=2, X2=Yp, coef=coef4, tag="dgrad", red=(Yprev, cprev), red_group=group)
        plain = ops.mlp_gemm(W, dZ, pro=2, X2=Yp, coef=coef4, tag="dgrad")[0]
    assert red is not None and torch.equal(dX, plain)                        # the sums change nothing of the product
    on = _fma(Yprev, cprev[0].view(1, M, 1), cprev[1].view(1, M, 1)) > 0
    d = torch.where(on, dX, torch.zeros_like(dX)).double()
    xhat = (Yprev.double() - cprev[2].double().view(1, M, 1)) * cprev[3].double().view(1, M, 1)
    s1, s2 = d.sum((0, 2)), (d * xhat).sum((0, 2))
    got = red.sums.double().sum(1)
    scale = max(float(s1.abs().max()), float(s2.abs().max()))
    assert float((got[0] - s1).abs().max()) <= 2e-6 * scale + 1e-6 * float(d.abs().sum((0, 2)).max())
    assert float((got[1] - s2).abs().max()) <= 2e-6 * scale + 1e-6 * float((d * xhat).abs().sum((0, 2)).max())
    assert float(red.maxima.max()) == float(d.abs().max())
    if group:
        gs = red.gsum.double()
        want_d = d.view(nb, M, P // group, group).sum(3)
        want_y = Yprev.double().view(nb, M, P // group, group).sum(3)
        assert float((gs[0] - want_d).abs().max()) <= 1e-5 * float(want_d.abs().max())
        assert float((gs[1] - want_y).abs().max()) <= 1e-5 * float(want_y.abs().max())
    # the consumer's view: the coefficients the finalisation derives from these partials = the stand-alone pass's
    dgam, dbet, c4 = ops.bn_backward_from_partials([red], nb * P, cprev, cprev[2].contiguous(), cprev[3].contiguous())
    dgam2, dbet2, c4b, _ = ops.bn_backward_reduce(dX, Yprev, cprev, cprev[2].contiguous(), cprev[3].contiguous(),
                                                  torch.ones(M, device=DEV), True)
    assert _rel(dbet, dbet2) < 1e-5 and _rel(dgam, dgam2) < 1e-5
    assert _rel(c4[:4], c4b[:4]) < 1e-5
    for _ in range(2):                                                       # bit-stable (fixed summation order)
        again = ops.mlp_gemm(W, None if pooled else dZ, pro=3 if pooled else 2, X2=Yp, coef=coef4, tag="dgrad",
                             pool=(dp, arg, G) if pooled else None, red=(Yprev, cprev), red_group=group)[2]
        assert torch.equal(again.flat, red.flat)
