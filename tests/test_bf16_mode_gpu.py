"""The bf16-multiply perf mode of the shared-MLP kernels (csrc/shared_mlp_bf16.hip; BASELINE.json configs[1]).

Kernel-level: the result must equal the fp64 product of the bf16-ROUNDED operands (prologue evaluated in fp32
first) to fp32 summation-order accuracy -- that pins the kernels exactly, independent of how lossy bf16 is.
Model-level: a whole detector step in bf16 mode stays within the documented tolerance of the fp32 mode."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _bf(t):
    return t.to(torch.bfloat16).to(torch.float64)


def _fma(a, b, c):
    """fp32 fmaf(a, b, c): the product is exact in fp64, one rounding to fp32 at the end."""
    return (a.double() * b.double() + c.double()).float()


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


@pytest.fixture
def bf16_mode():
    from usip_amd import ops
    prev = ops.set_matmul_mode("bf16")
    yield
    ops.set_matmul_mode(prev)


GEMM_SHAPES = [  # (nb, K, M, P)
    (2, 7, 64, 2048), (2, 64, 64, 4096), (2, 128, 128, 1024), (1, 131, 256, 512), (1, 512, 512, 1024),
    (3, 256, 4, 130), (1, 64, 64, 37), (2, 40, 130, 333), (1, 33, 70, 64),
]


@pytest.mark.parametrize("shape", GEMM_SHAPES)
@pytest.mark.parametrize("pro", [0, 1, 2])
def test_gemm_bf16_equals_product_of_rounded_operands(shape, pro, bf16_mode):
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
    fwd = pro < 2                                   # forward products carry the BatchNorm statistics epilogue
    Y, stats = ops.mlp_gemm(At, X, bias=bias, want_stats=fwd, pro=pro, X2=X2, coef=coef)
    want = torch.matmul(_bf(At).t().unsqueeze(0), _bf(xin)) + bias.double().view(1, M, 1)
    assert _rel(Y, want) < 2e-6
    if fwd:                                         # the statistics are those of the fp32 output
        s = stats.double().sum(-1)
        assert _rel(s[0], Y.double().sum((0, 2))) < 1e-5 or float(s[0].abs().max()) < 1e-3
        assert _rel(s[1], (Y.double() ** 2).sum((0, 2))) < 1e-5
    # and the fp32 mode really is a different kernel: bf16 rounding must be visible
    prev = ops.set_matmul_mode("f32")
    Y32, _ = ops.mlp_gemm(At, X, bias=bias, want_stats=fwd, pro=pro, X2=X2, coef=coef)
    ops.set_matmul_mode(prev)
    assert 1e-5 < _rel(Y, Y32) < 3e-2


@pytest.mark.parametrize("shape", [(2, 64, 7, 2048), (2, 64, 64, 4096), (2, 128, 128, 1024), (1, 256, 131, 512),
                                   (1, 512, 512, 1024), (3, 4, 256, 130), (1, 64, 64, 37), (2, 130, 40, 333)])
@pytest.mark.parametrize("pro", [0, 2])
@pytest.mark.parametrize("xpro", [False, True])
def test_wgrad_bf16_equals_product_of_rounded_operands(shape, pro, xpro, bf16_mode):
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
    want = torch.einsum("bmp,bnp->mn", _bf(gin), _bf(xin))
    assert _rel(dW, want) < 2e-6


def test_pooled_prologue_bf16(bf16_mode):
    """pro = 3 (gradient of a max-pooled layer synthesised from (dpooled, arg)) in both bf16 kernels."""
    from usip_amd import ops
    nb, K, M, Mn, Kn = 2, 128, 64, 96, 16
    P = Mn * Kn
    g = torch.Generator().manual_seed(5)
    At = torch.randn(K, M, generator=g).to(DEV) * 0.1
    Yl = torch.randn(nb, K, P, generator=g).to(DEV)
    dp = torch.randn(nb, K, Mn, generator=g).to(DEV)
    arg = torch.randint(0, Kn, (nb, K, Mn), generator=g, dtype=torch.int32).to(DEV)
    coef = torch.stack([1 + 0.1 * torch.randn(K, generator=g), 0.1 * torch.randn(K, generator=g),
                        0.05 * torch.randn(K, generator=g), 0.05 * torch.randn(K, generator=g)]).to(DEV)
    dZ = torch.zeros(nb, K, Mn, Kn, device=DEV).scatter_(3, arg.long().unsqueeze(-1), dp.unsqueeze(-1)).view(nb, K, P)
    c = [coef[i].view(1, K, 1) for i in range(4)]
    dyh = torch.where(_fma(Yl, c[0], c[1]) > 0, dZ, torch.zeros_like(dZ))
    dY = _fma(c[0], dyh, _fma(c[2], Yl, c[3]))
    got, _ = ops.mlp_gemm(At, None, pro=3, X2=Yl, coef=coef, pool=(dp, arg, Kn))
    assert _rel(got, torch.matmul(_bf(At).t().unsqueeze(0), _bf(dY))) < 2e-6
    Xin = torch.randn(nb, 40, P, generator=g).to(DEV)
    dW = ops.mlp_wgrad(None, Xin, pro=3, G2=Yl, coef4=coef, pool=(dp, arg, Kn))
    assert _rel(dW, torch.einsum("bmp,bnp->mn", _bf(dY), _bf(Xin))) < 2e-6


@pytest.mark.parametrize("model", ["ball", "som"])
def test_detector_step_bf16_mode_tracks_fp32(model, bf16_mode):
    """Whole step in bf16 mode vs fp32 mode on the same weights and inputs.  Tolerances are those DESIGN.md
    documents for the mode: keypoints 2e-2 of the cloud extent, sigmas 5e-2 relative, loss 2e-2 relative,
    gradient direction cosine > 0.98 per large tensor."""
    from usip_amd import ops, synth
    from usip_amd.networks import DetectorOptions
    from usip_amd.step import DetectorStep, batch_to_device
    opt = DetectorOptions(node_num=64, node_knn_k_1=16, surface_normal_len=4)
    batch = batch_to_device(synth.make_pair_batch(11, 2, 2048, 64, 4, "sphere"), DEV)
    torch.manual_seed(3)
    st = DetectorStep(model, opt, DEV)
    sd = {k: v.clone() for k, v in st.detector.state_dict().items()}
    loss_b = float(st.step(batch).detach())
    kp_b, sg_b, g_b = st.last["keypoints"].clone(), st.last["sigmas"].clone(), st.bucket.flat.clone()
    ops.set_matmul_mode("f32")
    st.detector.load_state_dict(sd)
    loss_f = float(st.step(batch).detach())
    ops.set_matmul_mode("bf16")
    kp_f, sg_f, g_f = st.last["keypoints"], st.last["sigmas"], st.bucket.flat
    assert abs(loss_b - loss_f) <= 2e-2 * abs(loss_f) + 1e-3
    assert float((kp_b - kp_f).detach().abs().max()) < 2e-2 * 2.4
    assert _rel(sg_b.detach(), sg_f.detach()) < 5e-2
    cos = float(torch.dot(g_b, g_f) / (g_b.norm() * g_f.norm()))
    assert cos > 0.98, cos


def test_config2_full_size_bf16(bf16_mode):
    """BASELINE.json configs[1] as stated: ModelNet40 detector (RPN_Detector), N=5000, M=64, node_knn_k_1=32,
    Cs=3, batch 24 pairs, bf16.  The integer front end (SOM assignment, index_max, kNN) does not depend on the
    multiply precision of the MLP *inputs* it sees first, so the SOM assignment must be identical to the fp32
    mode; floats stay within the bf16-mode tolerance; three Adam steps keep the loss finite and decreasing on
    the fixed batch."""
    from usip_amd import ops, synth
    from usip_amd.networks import DetectorOptions
    from usip_amd.step import DetectorStep, batch_to_device
    opt = DetectorOptions(surface_normal_len=3, node_knn_k_1=32, loss_sigma_lower_bound=1e-4, keypoint_on_pc_alpha=1.0)
    batch = batch_to_device(synth.make_pair_batch(2024, 24, 5000, 64, 3, "sphere"), DEV)
    torch.manual_seed(1)
    st = DetectorStep("som", opt, DEV, with_optimizer=True)
    sd = {k: v.clone() for k, v in st.detector.state_dict().items()}
    losses = [float(st.step(batch).detach()) for _ in range(3)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    assign_b = st.detector.last_indices["min_idx"].clone()
    # the same first step in fp32 mode
    ops.set_matmul_mode("f32")
    st2 = DetectorStep("som", opt, DEV)
    st2.detector.load_state_dict(sd)
    loss_f = float(st2.step(batch).detach())
    ops.set_matmul_mode("bf16")
    assert torch.equal(assign_b, st2.detector.last_indices["min_idx"])
    assert abs(losses[0] - loss_f) <= 2e-2 * abs(loss_f) + 1e-3
