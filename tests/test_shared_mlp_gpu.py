"""GPU parity of the hand-written shared-MLP kernels (csrc/shared_mlp.hip) against a plain
PyTorch reference of the same op evaluated in float64 (the 'truth' both fp32 paths approximate)
and in float32 (ATen on the same GPU): the HIP result must be within 1e-5 relative of truth and no
further from it than a small multiple of ATen's own fp32 error."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ref(x, w, b, gamma, beta, relu, gy, dtype):
    x = x.detach().to(dtype).requires_grad_(True)
    w = w.detach().to(dtype).requires_grad_(True)
    b = b.detach().to(dtype).requires_grad_(True)
    y = torch.matmul(w, x) + b.view(1, -1, 1)
    params = [x, w, b]
    if gamma is not None:
        gamma = gamma.detach().to(dtype).requires_grad_(True)
        beta = beta.detach().to(dtype).requires_grad_(True)
        y = F.batch_norm(y, None, None, gamma, beta, True, 0.1, 1e-5)
        params += [gamma, beta]
    if relu:
        y = torch.relu(y)
    y.backward(gy.to(dtype))
    return [y.detach()] + [p.grad for p in params]


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


SHAPES = [  # (nb, Cin, Cout, P)
    (3, 7, 12, 60), (4, 6, 9, 50), (2, 7, 64, 2048), (2, 64, 64, 4096), (2, 128, 128, 1024),
    (2, 131, 256, 512), (1, 512, 512, 1024), (2, 640, 512, 512), (3, 256, 4, 130), (1, 64, 64, 37),
]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("bn", [True, False])
def test_shared_mlp_layer_fwd_bwd(shape, bn):
    from usip_amd import functional as Fh
    nb, Cin, Cout, P = shape
    g = torch.Generator().manual_seed(nb * 1000 + Cin + Cout + P)
    x = torch.randn(nb, Cin, P, generator=g).to(DEV)
    w = (torch.randn(Cout, Cin, generator=g) * (2.0 / Cin) ** 0.5).to(DEV)
    b = (0.1 * torch.randn(Cout, generator=g)).to(DEV)
    gy = torch.randn(nb, Cout, P, generator=g).to(DEV)
    gamma = (1 + 0.1 * torch.randn(Cout, generator=g)).to(DEV) if bn else None
    beta = (0.1 * torch.randn(Cout, generator=g)).to(DEV) if bn else None
    relu = bn
    truth = _ref(x, w, b, gamma, beta, relu, gy, torch.float64)
    aten = _ref(x, w, b, gamma, beta, relu, gy, torch.float32)

    xs = x.clone().requires_grad_(True)
    ws = w.clone().view(Cout, Cin, 1).requires_grad_(True)
    bs = b.clone().requires_grad_(True)
    bn_mod = None
    if bn:
        bn_mod = torch.nn.BatchNorm1d(Cout).to(DEV)
        bn_mod.weight.data.copy_(gamma)
        bn_mod.bias.data.copy_(beta)
        bn_mod.train()
    y = Fh.conv1x1_bn_act(xs, ws, bs, bn_mod, relu)
    y.backward(gy)
    got = [y.detach(), xs.grad, ws.grad.view(Cout, Cin), bs.grad]
    if bn:
        got += [bn_mod.weight.grad, bn_mod.bias.grad]
    names = ["y", "dx", "dw", "db", "dgamma", "dbeta"]
    for name, a, t, f32 in zip(names, got, truth, aten):
        if name == "db" and bn:
            assert float(a.abs().max()) == 0.0      # analytically zero, returned as exact zeros
            continue
        err, aten_err = _rel(a, t), _rel(f32, t)
        assert err <= max(1e-5, 4 * aten_err), (name, err, aten_err)
    if bn:
        ref_bn = torch.nn.BatchNorm1d(Cout).to(DEV).train()
        ref_bn(torch.matmul(w, x) + b.view(1, -1, 1))
        assert _rel(bn_mod.running_mean, ref_bn.running_mean) <= 1e-5
        assert _rel(bn_mod.running_var, ref_bn.running_var) <= 1e-5
        assert int(bn_mod.num_batches_tracked) == 1


def test_shared_mlp_eval_mode_uses_running_stats():
    from usip_amd import functional as Fh
    nb, Cin, Cout, P = 2, 16, 24, 256
    x = torch.randn(nb, Cin, P, device=DEV)
    conv = torch.nn.Conv1d(Cin, Cout, 1).to(DEV)
    bn = torch.nn.BatchNorm1d(Cout).to(DEV)
    bn.running_mean.normal_()
    bn.running_var.uniform_(0.5, 2.0)
    bn.eval()
    want = torch.relu(bn(conv(x)))
    got = Fh.conv1x1_bn_act(x, conv.weight, conv.bias, bn, True)
    assert _rel(got, want) <= 1e-5


def test_lazy_activation_chain_equals_materialised_chain():
    """conv-bn-relu x3 -> max over K with activations handed on as LazyAct (BN+ReLU applied in the consumer's
    prologue, fused BN+ReLU+max) must equal the same chain with every activation materialised: same forward,
    same parameter and input gradients (identical arithmetic per element, so essentially bit-equal)."""
    from usip_amd import functional as Fh
    from usip_amd import layers
    torch.manual_seed(3)
    B, C0, M, K = 2, 7, 24, 16
    x = torch.randn(B, C0, M, K, device=DEV)
    gy = torch.randn(B, 48, M, device=DEV)

    def build():
        torch.manual_seed(11)
        mods = [layers.MyConv2d(C0, 32, (1, 1), activation="relu", normalization="batch"),
                layers.MyConv2d(32, 40, (1, 1), activation="relu", normalization="batch"),
                layers.MyConv2d(40, 48, (1, 1), activation="relu", normalization="batch")]
        return [m.to(DEV).train() for m in mods]

    outs = []
    for lazy in (False, True):
        mods = build()
        xi = x.clone().requires_grad_(True)
        h = xi
        for m in mods:
            h = m(h, None, defer=lazy)
        if lazy:
            assert isinstance(h, Fh.LazyAct)
        pooled = Fh.group_max(h)
        pooled.backward(gy)
        outs.append([pooled.detach(), xi.grad] + [p.grad for m in mods for p in m.parameters()])
    for a, b in zip(*outs):
        assert _rel(a, b) <= 2e-6


def test_gemm_reads_transposed_operand_in_place():
    """C ABI: a negative lda makes usip_mlp_gemm_f32 read the matrix operand stored [M][K] transposed in place."""
    from usip_amd import ops
    g = torch.Generator().manual_seed(5)
    W = torch.randn(48, 36, generator=g).to(DEV)
    X = torch.randn(2, 36, 200, generator=g).to(DEV)
    y1, _ = ops.mlp_gemm(W.t().contiguous(), X)
    y2, _ = ops.mlp_gemm(W, X, a_trans=True)
    assert torch.equal(y1, y2)


def test_nograd_prefix_skips_rows_of_the_data_gradient():
    """conv1x1_bn_act(..., nograd_prefix=3): the data gradient of the remaining channels is computed straight
    into its slice of the full tensor (GEMM with a destination row offset) and equals the full computation; the
    prefix rows are zeros; parameter gradients do not change."""
    from usip_amd import functional as Fh
    B, Cin, Cout, M, K = 2, 3 + 128, 64, 24, 16
    g = torch.Generator().manual_seed(8)
    x = torch.randn(B, Cin, M, K, generator=g).to(DEV)
    w = (torch.randn(Cout, Cin, 1, 1, generator=g) * 0.1).to(DEV)
    gy = torch.randn(B, Cout, M, K, generator=g).to(DEV)

    def run(prefix):
        xs = x.clone().requires_grad_(True)
        ws = w.clone().requires_grad_(True)
        bn = torch.nn.BatchNorm2d(Cout).to(DEV)
        y = Fh.conv1x1_bn_act(xs, ws, None, bn, True, nograd_prefix=prefix)
        y.backward(gy)
        return y.detach(), xs.grad, ws.grad, bn.weight.grad, bn.bias.grad

    y0, gx0, gw0, gg0, gb0 = run(0)
    y1, gx1, gw1, gg1, gb1 = run(3)
    assert torch.equal(y0, y1) and torch.equal(gw0, gw1) and torch.equal(gg0, gg1) and torch.equal(gb0, gb1)
    assert float(gx1[:, :3].abs().max()) == 0.0 and float(gx0[:, :3].abs().max()) > 0.0
    assert _rel(gx1[:, 3:], gx0[:, 3:]) < 1e-6              # different tile split: summation order only


def test_gemm_writes_a_channel_slice_of_a_wider_tensor():
    """C ABI: y_rows / a destination row offset (ops.mlp_gemm(out=, out_row_offset=))."""
    from usip_amd import ops
    nb, K, M, P = 3, 40, 70, 260
    g = torch.Generator().manual_seed(9)
    At = torch.randn(K, M, generator=g).to(DEV)
    X = torch.randn(nb, K, P, generator=g).to(DEV)
    want, _ = ops.mlp_gemm(At, X)
    out = torch.full((nb, M + 9, P), 7.0, device=DEV)
    ops.mlp_gemm(At, X, out=out, out_row_offset=5)
    assert torch.equal(out[:, 5:5 + M], want)
    assert bool((out[:, :5] == 7.0).all()) and bool((out[:, 5 + M:] == 7.0).all())
    with pytest.raises(RuntimeError):
        ops.mlp_gemm(At, X, out=out, out_row_offset=10)


@pytest.mark.parametrize("cfg", [(2, 64, 4096, True, 64, 0), (2, 128, 2048, True, 64, 0), (3, 64, 1024, False, 64, 0),
                                 (2, 128, 4096, True, 128, 64), (1, 128, 640, False, 128, 0)])
def test_fused_narrow_backward_equals_separate_products(cfg):
    """csrc/narrow_bwd.hip: data gradient and weight gradient of a 64-input layer from ONE pass over (dZ, Y, X) must
    equal the two generic products (usip_mlp_gemm_f32 pro=2 and usip_mlp_wgrad_f32 pro=2) and fp64 truth; also for a
    column block of a wider weight matrix (the feature half of the pooled-concat layer) and bit for bit on a re-run."""
    from usip_amd import ops
    nb, Cout, P, xpro, Ctot, wcol = cfg
    Cin = 64
    g = torch.Generator().manual_seed(Cout + P + wcol)
    dz = torch.randn(nb, Cout, P, generator=g).to(DEV)
    y = torch.randn(nb, Cout, P, generator=g).to(DEV)
    x = torch.randn(nb, Cin, P, generator=g).to(DEV)
    w2 = (torch.randn(Cout, Ctot, generator=g) * (2.0 / Cin) ** 0.5).to(DEV)
    coef4 = torch.stack([1 + 0.1 * torch.randn(Cout, generator=g), 0.1 * torch.randn(Cout, generator=g),
                         0.05 * torch.randn(Cout, generator=g), 0.05 * torch.randn(Cout, generator=g)]).to(DEV)
    xcoef = torch.stack([1 + 0.1 * torch.randn(Cin, generator=g), 0.1 * torch.randn(Cin, generator=g)]).to(DEV) if xpro else None
    dw_out = torch.full((Cout, Ctot), 7.0, device=DEV)
    dx, dw = ops.mlp_narrow_backward(dz, y, coef4, x, xcoef, w2, wcol=wcol, dw_out=dw_out)
    # truth in float64
    c = [coef4[i].double().view(1, Cout, 1) for i in range(4)]
    fma = lambda a_, b_, c_: (a_.double() * b_.double() + c_.double()).float()
    dyh = torch.where(fma(y, c[0], c[1]) > 0, dz, torch.zeros_like(dz))
    dy = fma(c[0], dyh, fma(c[2], y, c[3])).double()
    ax = torch.relu(fma(x, xcoef[0].view(1, Cin, 1), xcoef[1].view(1, Cin, 1))).double() if xpro else x.double()
    wsub = w2[:, wcol:wcol + Cin].double()
    want_dx = torch.einsum("oc,bop->bcp", wsub, dy)
    want_dw = torch.einsum("bop,bcp->oc", dy, ax)
    assert _rel(dx, want_dx) < 2e-6 and _rel(dw[:, wcol:wcol + Cin], want_dw) < 2e-6
    if Ctot > Cin:                                       # columns outside the block are untouched
        keep = torch.ones(Ctot, dtype=torch.bool)
        keep[wcol:wcol + Cin] = False
        assert bool((dw[:, keep.to(DEV)] == 7.0).all())
    # the generic pair
    wsub_c = w2[:, wcol:wcol + Cin].contiguous()
    dx2 = ops.mlp_gemm(wsub_c, dz, pro=2, X2=y, coef=coef4, tag="dgrad")[0]
    dw2 = ops.mlp_wgrad(dz, x, pro=2, G2=y, coef4=coef4, xcoef=xcoef)
    assert _rel(dx, dx2) < 2e-6 and _rel(dw[:, wcol:wcol + Cin], dw2) < 2e-6
    dx3, dw3 = ops.mlp_narrow_backward(dz, y, coef4, x, xcoef, w2, wcol=wcol, dw_out=torch.full((Cout, Ctot), 7.0, device=DEV))
    assert torch.equal(dx3, dx) and torch.equal(dw3, dw)


@pytest.mark.parametrize("Cout", [64, 128])
def test_fused_narrow_backward_takes_the_producers_bn_sums(Cout):
    """want_red: the BatchNorm-backward sums of the layer that PRODUCED X, taken by the fused kernel on its way out
    (dX tile in registers, X tile in LDS), must give the same dgamma / dbeta / coef4 as that layer's own reduction
    pass over (dX, X) -- also when a sparse max-pool gradient is added on top (sums are linear in the gradient)."""
    from usip_amd import ops
    nb, Cin, P, K = 2, 64, 4096, 64
    g = torch.Generator().manual_seed(Cout)
    dz = torch.randn(nb, Cout, P, generator=g).to(DEV)
    y = torch.randn(nb, Cout, P, generator=g).to(DEV)
    x = torch.randn(nb, Cin, P, generator=g).to(DEV)                       # pre-BN output of the producing layer
    w2 = (torch.randn(Cout, Cin, generator=g) * 0.2).to(DEV)
    coef4 = torch.stack([1 + 0.1 * torch.randn(Cout, generator=g), 0.1 * torch.randn(Cout, generator=g),
                         0.05 * torch.randn(Cout, generator=g), 0.05 * torch.randn(Cout, generator=g)]).to(DEV)
    gamma = (1 + 0.1 * torch.randn(Cin, generator=g)).to(DEV)
    beta = (0.1 * torch.randn(Cin, generator=g)).to(DEV)
    mean, var = x.mean((0, 2)), x.var((0, 2), unbiased=False)
    invstd = torch.rsqrt(var + 1e-5)
    xcoef = torch.stack([gamma * invstd, beta - mean * gamma * invstd, mean, invstd]).contiguous()
    dx, _, red = ops.mlp_narrow_backward(dz, y, coef4, x, xcoef, w2, want_red=True)
    dg, db, c4 = ops.bn_backward_from_partials(red, nb * P, xcoef, mean.contiguous(), invstd.contiguous())
    dg2, db2, c42, _ = ops.bn_backward_reduce(dx, x, xcoef, mean.contiguous(), invstd.contiguous(), gamma, True)
    for a_, b_, n in ((dg, dg2, "dgamma"), (db, db2, "dbeta"), (c4[:4], c42[:4], "coef4")):
        assert _rel(a_, b_) < 5e-6, (n, _rel(a_, b_))
    # the kernel also leaves the maxima of |dX [relu on]|: coef4 gets its fifth row, a bound of the producing layer's
    # |dY| = |a1 dYhat + q1 x + q0| (what that layer's f32x2 backward scales its operand by)
    assert c4.shape[0] == 5
    on = (x * xcoef[0].view(1, Cin, 1) + xcoef[1].view(1, Cin, 1)) > 0
    dyl = c4[0].view(1, Cin, 1) * torch.where(on, dx, torch.zeros_like(dx)) + c4[2].view(1, Cin, 1) * x + c4[3].view(1, Cin, 1)
    assert float(c4[4, 0]) >= float(dyl.abs().max()) and float(c4[4, 0]) < 1e3 * float(dyl.abs().max())
    # plus a sparse pooling gradient at the arg-max of the activated producer output
    M = P // K
    pooled, arg = ops.group_max_act(x.view(nb, Cin, M, K), xcoef, True)
    dpooled = torch.randn(nb, Cin, M, generator=g).to(DEV)
    sparse = ops.bn_pool_backward_partials(dpooled, arg, x.view(nb, Cin, M, K), xcoef, xcoef[2], xcoef[3], True)
    dg3, db3, c43 = ops.bn_backward_from_partials([red, sparse], nb * P, xcoef, mean.contiguous(), invstd.contiguous())
    total = ops.group_max_backward_add_(dx.clone().view(nb, Cin, M, K), dpooled, arg).view(nb, Cin, P)
    dg4, db4, c44, _ = ops.bn_backward_reduce(total, x, xcoef, mean.contiguous(), invstd.contiguous(), gamma, True)
    for a_, b_, n in ((dg3, dg4, "dgamma"), (db3, db4, "dbeta"), (c43[:4], c44[:4], "coef4")):
        assert _rel(a_, b_) < 5e-6, (n, _rel(a_, b_))


def test_narrow_chain_gradients_do_not_depend_on_the_fused_backward():
    """conv2 -> conv3 -> (max over K, conv4 on cat(features, max)) -> conv5 + max, the Ball front end's chain, with the
    fused narrow backward (and the BatchNorm sums it hands upstream) on and off: all gradients agree to rounding."""
    from usip_amd import functional as Fh
    from usip_amd import layers

    def run(fused):
        torch.manual_seed(3)
        kw = dict(kernel_size=(1, 1), stride=1, padding=0, bias=True, activation="relu", normalization="batch")
        convs = [layers.MyConv2d(7, 64, **kw), layers.MyConv2d(64, 64, **kw), layers.MyConv2d(64, 64, **kw),
                 layers.MyConv2d(128, 128, **kw), layers.MyConv2d(128, 128, **kw)]
        convs = [c_.to(DEV).train() for c_ in convs]
        g = torch.Generator().manual_seed(8)
        x = torch.randn(2, 7, 32, 64, generator=g).to(DEV)
        prev, Fh.FUSED_NARROW_BWD = Fh.FUSED_NARROW_BWD, fused
        try:
            h = convs[2](convs[1](convs[0](x, defer=True), defer=True), defer=True)
            pooled, h = Fh.group_max_fork(h)
            h = Fh.conv1x1_bn_act_pooled(h, pooled, convs[3].conv.weight, convs[3].conv.bias, convs[3].norm, True,
                                         pooled_first=False, defer=True)
            out = Fh.conv1x1_bn_relu_max(h, convs[4].conv.weight, convs[4].conv.bias, convs[4].norm)
            (out * torch.randn(out.shape, generator=g).to(DEV)).sum().backward()
        finally:
            Fh.FUSED_NARROW_BWD = prev
            Fh.PRE_BN_SUMS.clear()
        return [p.grad.clone() for c_ in convs for p in c_.parameters()]

    for a_, b_ in zip(run(True), run(False)):
        if float(b_.abs().max()) > 0:
            assert _rel(a_, b_) < 2e-5


# ------------------------------------------------------------------ streaming forward kernel of the narrow layers
@pytest.mark.parametrize("M", [64, 128])
@pytest.mark.parametrize("P,nb", [(32768, 16), (8192, 16), (16644, 16), (40000, 8)])
@pytest.mark.parametrize("variant", ["plain", "bnrelu", "rowbias", "into_wider"])
def test_narrow_forward_streaming_kernel_matches_generic(M, P, nb, variant):
    """csrc/narrow_fwd.hip against the generic tile kernel on the same inputs (same fp32 MFMA arithmetic, another
    summation order over k): outputs to 2e-6 of the output scale, BatchNorm sums (partials added up in double) to
    1e-6 of the magnitude they add up; ragged last tiles (P not a multiple of 256 / 128), a row bias per 64-position group, and a
    result written into the rows of a wider tensor."""
    from usip_amd import ops
    K = 64
    if ops._lib.lib().usip_mlp_narrow_forward_blocks(M, K, P, nb) == 0:
        pytest.skip("too few tiles for the persistent workgroups: this shape runs the generic kernel")
    g = torch.Generator(device="cpu").manual_seed(M + P)
    At = (torch.randn(K, M + 8, generator=g) / 8).to(DEV)                 # lda > M: a column slice of a wider matrix
    X = torch.randn(nb, K, P, generator=g).to(DEV)
    bias = torch.randn(M, generator=g).to(DEV)
    kw = dict(M=M, a_offset=4)
    if variant in ("bnrelu", "rowbias"):
        coef = torch.stack([1 + 0.2 * torch.randn(K, generator=g), 0.3 * torch.randn(K, generator=g)]).to(DEV).contiguous()
        kw.update(pro=1, coef=coef)
    if variant == "rowbias":
        if P % 64:
            pytest.skip("row-bias groups of 64 positions need P % 64 == 0")
        kw.update(rowbias=torch.randn(nb, M, P // 64, generator=g).to(DEV), rb_group=64)
    outs = []
    for streaming in (True, False):
        ops.NARROW_FWD = streaming
        try:
            if variant == "into_wider":
                out = torch.full((nb, M + 5, P), 7.0, device=DEV)
                y, st = ops.mlp_gemm(At, X, bias, want_stats=True, out=out, out_row_offset=3, **kw)
                assert float((out[:, :3] - 7.0).abs().max()) == 0.0 and float((out[:, 3 + M:] - 7.0).abs().max()) == 0.0
                y = out[:, 3:3 + M]
            else:
                y, st = ops.mlp_gemm(At, X, bias, want_stats=True, **kw)
        finally:
            ops.NARROW_FWD = True
        outs.append((y, st.double().sum(dim=2)))
    (y1, s1), (y0, s0) = outs
    scale = float(y0.abs().max())
    assert float((y1 - y0).abs().max()) <= 2e-6 * scale
    # sums against what they add up: sum |y| per channel for the plain sums (they cancel), the sum itself for squares
    mag = torch.stack((y0.double().abs().sum(dim=(0, 2)), s0[1]))
    assert float(((s1 - s0).abs() / mag).max()) <= 1e-6
    # ... and without statistics
    y2, st2 = ops.mlp_gemm(At, X, bias, **kw)
    assert st2 is None and torch.equal(y2, y1 if variant != "into_wider" else y2)
