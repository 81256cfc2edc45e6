"""GPU parity tests of the native operators (through the C ABI) against the oracle and the
golden vectors.  Integer/index outputs: bit-exact."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import native

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ops():
    from usip_amd import ops
    return ops


# ------------------------------------------------------------------ index_max
@pytest.mark.parametrize("tag", ["random", "ties", "floor", "empty", "nan", "tiny", "n_lt_wave"])
def test_index_max_golden(tag):
    g = load_golden("index_max_cases.npz")
    d = torch.from_numpy(g[tag + "_data"]).to(DEV)
    i = torch.from_numpy(g[tag + "_index"]).to(DEV)
    import usip_amd
    im, _ = usip_amd.install()
    for fn in (im.forward_cuda, im.forward_cuda_shared_mem):
        out = fn(d, i, int(g[tag + "_K"]))
        assert out.dtype == torch.int32 and out.is_cuda
        assert np.array_equal(out.cpu().numpy(), g[tag + "_out"])


@pytest.mark.parametrize("shape", [(2, 64, 1024, 64), (4, 128, 5000, 64), (3, 6, 1001, 33), (16, 64, 16384, 512),
                                   (1, 3, 4096, 4096), (2, 8, 8192, 8192)])
def test_index_max_vs_oracle(shape):
    B, C, N, K = shape
    rng = np.random.default_rng(B * 1000 + C)
    data = rng.normal(0, 1, (B, C, N)).astype(np.float32)
    data[:, :, ::5] = np.round(data[:, :, ::5])            # plenty of exact ties, incl. +-0
    data[0, 0, :7] = -0.0
    idx = rng.integers(0, K, (B, N)).astype(np.int32)
    want = native.index_max(data, idx, K)
    got = _ops().index_max(torch.from_numpy(data).to(DEV), torch.from_numpy(idx).to(DEV), K)
    assert np.array_equal(got.cpu().numpy(), want)


def test_index_max_full_size_properties():
    """BASELINE config 3 size (B'=16, C=128, N=16384, M=512): size-independent properties.
    (1) every returned n is assigned to its node and attains the node's max; (2) idempotence
    under a permutation of channels."""
    B, C, N, K = 16, 128, 16384, 512
    g = torch.Generator(device="cpu").manual_seed(5)
    data = torch.randn(B, C, N, generator=g).to(DEV)
    idx = torch.randint(0, K, (B, N), generator=g, dtype=torch.int32).to(DEV)
    out = _ops().index_max(data, idx, K).long()
    assigned = torch.gather(idx.long().unsqueeze(1).expand(B, C, N), 2, out)       # node of the winner
    node_ids = torch.arange(K, device=DEV).view(1, 1, K).expand(B, C, K)
    counts = torch.zeros(B, K, device=DEV).scatter_add_(1, idx.long(), torch.ones(B, N, device=DEV))
    nonempty = (counts > 0).unsqueeze(1).expand(B, C, K)
    assert torch.equal(assigned[nonempty], node_ids[nonempty])
    seg_max = torch.full((B, C, K), -float("inf"), device=DEV).scatter_reduce_(
        2, idx.long().unsqueeze(1).expand(B, C, N), data, reduce="amax")
    assert torch.equal(torch.gather(data, 2, out)[nonempty], seg_max[nonempty])
    assert int(out[~nonempty].abs().sum()) == 0
    perm = torch.randperm(C, device=DEV)
    out_p = _ops().index_max(data[:, perm].contiguous(), idx, K).long()
    assert torch.equal(out_p, out[:, perm])


# ------------------------------------------------------------------ pairwise distance + ball_query
def test_pairwise_dist_golden_bit_exact():
    g = load_golden("dist_ball_cases.npz")
    d = _ops().pairwise_dist(torch.from_numpy(g["node"]).to(DEV), torch.from_numpy(g["x"]).to(DEV))
    assert np.array_equal(d.cpu().numpy(), g["dist"])


def test_ball_query_golden():
    g = load_golden("dist_ball_cases.npz")
    import usip_amd
    _, bq = usip_amd.install()
    dist = torch.from_numpy(g["dist"]).to(DEV)
    for fn in (bq.forward_cuda_shared_mem, bq.forward_cuda):
        out = fn(dist, float(g["radius"]), int(g["K"]))
        assert out.dtype == torch.int32
        assert np.array_equal(out.cpu().numpy(), g["ball_idx_unpinned"])
    fused = _ops().ball_query_coords(torch.from_numpy(g["node"]).to(DEV), torch.from_numpy(g["x"]).to(DEV),
                                     float(g["radius"]), int(g["K"]))
    assert np.array_equal(fused.cpu().numpy(), g["ball_idx_unpinned"])


@pytest.mark.parametrize("shape", [(2, 64, 1024, 64), (4, 64, 5000, 32), (3, 17, 1001, 7), (16, 512, 4096, 64),
                                   (1, 8, 16384, 64), (2, 100, 16384, 64), (1, 2, 300, 1), (2, 3, 64, 128)])
@pytest.mark.parametrize("density", ["sparse", "dense", "mixed"])
def test_ball_query_vs_oracle(shape, density):
    B, M, N, K = shape
    rng = np.random.default_rng(B + M + N)
    dist = rng.uniform(0, 10, (B, M, N)).astype(np.float32)
    r = {"sparse": 0.002, "dense": 6.0, "mixed": 0.08}[density]
    if density == "mixed":
        dist[:, ::3] += 20.0                               # every third row: empty ball
        dist[:, 1::3, N // 2:] = 0.0                       # hits only in the second half
    dist[0, 0, -1] = np.float32(r)                         # boundary: <= is inclusive
    dist[-1, -1, 0] = np.nan
    want = native.ball_query(dist, r, K)
    got = _ops().ball_query(torch.from_numpy(dist).to(DEV), r, K)
    assert np.array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize("kind,N,M", [("slab:20", 8192, 256), ("cube", 4096, 64), ("sphere", 1000, 33)])
def test_fused_coords_equals_dist_then_query(kind, N, M):
    from usip_amd import synth
    rng = np.random.default_rng(11)
    B, K = 3, 64
    x = np.stack([synth.make_cloud(rng, N, kind) for _ in range(B)])
    node = np.ascontiguousarray(np.stack([x[b][:, rng.permutation(N)[:M]] for b in range(B)]))
    r = 2.0 if kind != "sphere" else 0.3
    tx, tn = torch.from_numpy(x).to(DEV), torch.from_numpy(node).to(DEV)
    dist = _ops().pairwise_dist(tn, tx)
    assert np.array_equal(dist.cpu().numpy(), native.pairwise_dist(node, x))
    a = _ops().ball_query(dist, r, K)
    b = _ops().ball_query_coords(tn, tx, r, K)
    assert torch.equal(a, b)
    assert np.array_equal(a.cpu().numpy(), native.ball_query(dist.cpu().numpy(), r, K))


def test_ball_query_full_size_properties():
    """BASELINE config 3 size (B'=16, M=512, N=16384, K=64) on 'slab' clouds: every listed index
    is inside the ball, strictly ascending over the genuine hits, cyclic beyond them, and the
    number of genuine hits equals min(K, #inside)."""
    from usip_amd import synth
    B, M, N, K, r = 16, 512, 16384, 64, 2.0
    rng = np.random.default_rng(3)
    x = np.stack([synth.make_cloud(rng, N, "slab") for _ in range(B)])
    node = np.ascontiguousarray(np.stack([x[b][:, rng.permutation(N)[:M]] for b in range(B)]))
    tx, tn = torch.from_numpy(x).to(DEV), torch.from_numpy(node).to(DEV)
    dist = _ops().pairwise_dist(tn, tx)
    out = _ops().ball_query(dist, r, K).long()
    inside = dist <= r
    n_in = inside.sum(-1)
    u = torch.clamp(n_in, max=K)
    assert bool(torch.gather(inside, 2, out)[n_in > 0].all())
    j = torch.arange(K, device=DEV).view(1, 1, K)
    genuine = j < u.unsqueeze(-1)
    asc = (out[..., 1:] > out[..., :-1]) | ~genuine[..., 1:]
    assert bool(asc.all())
    cyc = torch.gather(out, 2, j % torch.clamp(u, min=1).unsqueeze(-1))
    assert torch.equal(out, torch.where(u.unsqueeze(-1) > 0, cyc, torch.zeros_like(out)))
    # the genuine hits are the FIRST ones: rank of out[j] among the inside points is j
    rank = torch.cumsum(inside.long(), -1) - 1
    assert torch.equal(torch.gather(rank, 2, out)[genuine], j.expand_as(out)[genuine])
    assert torch.equal(_ops().ball_query_coords(tn, tx, r, K).long(), out)


# ------------------------------------------------------------------ node KNN
@pytest.mark.parametrize("shape", [(3, 64, 64, 32), (2, 512, 512, 16), (2, 100, 777, 5), (1, 7, 1024, 64), (2, 30, 30, 30)])
def test_knn_matches_oracle_topk(shape):
    """usip_knn_f32 against torch.norm + torch.topk(sorted=True) on the pinned CPU platform."""
    from usip_amd import synth
    B, M, N, K = shape
    rng = np.random.default_rng(M + N + K)
    db = np.ascontiguousarray(np.stack([synth.make_cloud(rng, N, "slab:10") for _ in range(B)]))
    q = db[:, :, :M].copy() if M <= N else np.ascontiguousarray(np.stack([synth.make_cloud(rng, M, "slab:10") for _ in range(B)]))
    tq, td = torch.from_numpy(q), torch.from_numpy(db)
    norm = torch.norm(tq.unsqueeze(3) - td.unsqueeze(2), dim=1)
    want_d, want_i = torch.topk(norm, k=K, dim=2, largest=False, sorted=True)
    got = _ops().knn(tq.to(DEV), td.to(DEV), K).cpu().long()
    got_d = torch.gather(norm, 2, got)
    assert torch.equal(got_d, want_d)                       # same distances in the same order (bit-exact values)
    distinct = (want_d[..., 1:] != want_d[..., :-1]).all(-1)
    assert torch.equal(got[distinct], want_i[distinct])     # identical indices wherever there is no exact tie


# ------------------------------------------------------------------ degenerate / extreme inputs
def test_empty_and_degenerate_inputs_do_not_launch_garbage():
    """Zero-sized dimensions return correctly shaped outputs without touching memory; K larger than any
    hit count pads cyclically; a single point / single node works."""
    ops = _ops()
    f32, i32 = torch.float32, torch.int32
    assert ops.index_max(torch.empty(0, 4, 16, device=DEV), torch.empty(0, 16, dtype=i32, device=DEV), 8).shape == (0, 4, 8)
    assert ops.ball_query(torch.empty(0, 5, 32, device=DEV), 1.0, 4).shape == (0, 5, 4)
    assert ops.ball_query(torch.empty(2, 0, 32, device=DEV), 1.0, 4).shape == (2, 0, 4)
    assert ops.pairwise_dist(torch.empty(0, 3, 4, device=DEV), torch.empty(0, 3, 9, device=DEV)).shape == (0, 4, 9)
    assert ops.ball_query_coords(torch.empty(0, 3, 4, device=DEV), torch.empty(0, 3, 9, device=DEV), 1.0, 4).shape == (0, 4, 4)
    # one point, one node, K far above the number of hits: the single hit is repeated
    x = torch.tensor([[[1.0], [2.0], [3.0]]], device=DEV)
    out = ops.ball_query_coords(x.clone(), x, 0.5, 7)
    assert out.tolist() == [[[0] * 7]]
    d = ops.pairwise_dist(x.clone(), x)
    assert float(d) == 0.0 and ops.ball_query(d, 0.0, 3).tolist() == [[[0, 0, 0]]]      # <= is inclusive at 0
    # radius below every distance: all-zero row; negative radius and NaN radius: no hit either
    far = torch.full((1, 2, 40), 5.0, device=DEV)
    for r in (1.0, -1.0, float("nan")):
        assert int(ops.ball_query(far, r, 6).abs().sum()) == 0
    xs = torch.randn(1, 3, 40, device=DEV)
    nd = xs[:, :, :2].contiguous() + 100.0
    for r in (1.0, -1.0, float("nan")):
        assert int(ops.ball_query_coords(nd, xs, r, 6).abs().sum()) == 0
    # infinite radius: the first K points
    assert ops.ball_query_coords(nd, xs, float("inf"), 6).tolist() == [[list(range(6))] * 2]
    # index_max with K = 1 and with every value at the floor
    data = torch.full((1, 2, 10), -1000.0, device=DEV)
    assert int(ops.index_max(data, torch.zeros(1, 10, dtype=i32, device=DEV), 1).abs().sum()) == 0


def test_largest_supported_group_sizes():
    """K at the kernels' documented limits: ball_query K = 8192 (LDS list), fused coords K = 1024."""
    ops = _ops()
    rng = np.random.default_rng(9)
    dist = rng.uniform(0, 1, (1, 3, 20000)).astype(np.float32)
    want = native.ball_query(dist, 0.9, 8192)
    assert np.array_equal(ops.ball_query(torch.from_numpy(dist).to(DEV), 0.9, 8192).cpu().numpy(), want)
    x = rng.uniform(-1, 1, (1, 3, 5000)).astype(np.float32)
    node = np.ascontiguousarray(x[:, :, :5])
    d = native.pairwise_dist(node, x)
    want = native.ball_query(d, 0.8, 1024)
    got = ops.ball_query_coords(torch.from_numpy(node).to(DEV), torch.from_numpy(x).to(DEV), 0.8, 1024)
    assert np.array_equal(got.cpu().numpy(), want)
    with pytest.raises(RuntimeError, match="USIP_EINVAL"):
        ops.ball_query(torch.from_numpy(dist).to(DEV), 0.9, 8193)


def test_chamfer_prob_kernels_match_torch_float64():
    """usip_chamfer_prob_f32 (+ backward) against the reference formulas (models/losses.py:82-99) evaluated
    by autograd in float64, ragged M != N, repeated partners (many-to-one gather)."""
    ops = _ops()
    B, M, N = 3, 700, 1300                       # N > 1024: more than one LDS chunk in the segmented sums
    g = torch.Generator().manual_seed(2)
    a = torch.rand(B, M, generator=g).to(DEV)
    c = torch.rand(B, N, generator=g).to(DEV)
    J = torch.randint(0, N, (B, M), generator=g, dtype=torch.int32).to(DEV)
    I = torch.randint(0, 40, (B, N), generator=g, dtype=torch.int32).to(DEV)      # heavy collisions
    ss = (0.05 + torch.rand(B, M, generator=g)).to(DEV)
    sd = (0.05 + torch.rand(B, N, generator=g)).to(DEV)
    out = ops.chamfer_prob(a, J, c, I, ss, sd)
    gl = torch.tensor(0.7, device=DEV)
    da, dc, dss, dsd = ops.chamfer_prob_backward(gl, a, J, c, I, ss, sd)
    A, C, SS, SD = (t.double().requires_grad_(True) for t in (a, c, ss, sd))
    sf = (SS + torch.gather(SD, 1, J.long())) / 2
    sb = (SD + torch.gather(SS, 1, I.long())) / 2
    loss = (torch.log(sf) + A / sf).mean() + (torch.log(sb) + C / sb).mean()
    pure = A.mean() + C.mean()
    wf, wb = (1 / sf) / (1 / sf).mean(), (1 / sb) / (1 / sb).mean()
    weighted = (wf * A).mean() + (wb * C).mean()
    (0.7 * loss).backward()
    for got, want in ((out[0], loss), (out[1], pure), (out[2], weighted)):
        assert abs(float(got) - float(want)) <= 2e-6 * abs(float(want))
    for got, want, name in ((da, A.grad, "da"), (dc, C.grad, "dc"), (dss, SS.grad, "dss"), (dsd, SD.grad, "dsd")):
        err = float((got.double() - want).abs().max() / want.abs().max())
        assert err < 2e-6, (name, err)
    # deterministic: bit-identical on a second run
    assert torch.equal(ops.chamfer_prob_backward(gl, a, J, c, I, ss, sd)[3], dsd)


def test_ball_query_kernels_match_reference_numba_ancestor():
    """Both HIP entry points (dist-in, and coords-in on a matrix rebuilt from coordinates is covered elsewhere)
    against the rows the reference's numba ancestor produced (tests/golden/ball_query_ancestor_cases.npz)."""
    ops = _ops()
    g = load_golden("ball_query_ancestor_cases.npz")
    for n in "abc":
        dist, K, r = g[n + "_dist"], int(g[n + "_K"]), float(g[n + "_radius"])
        got = ops.ball_query(torch.from_numpy(dist).to(DEV), r, K).cpu().numpy()
        empty = g[n + "_empty"]
        assert np.array_equal(got[~empty], g[n + "_idx"][~empty]), n
        assert not got[empty].any()



def test_multi_transpose_matches_per_tensor_transposes():
    ops = _ops()
    g = torch.Generator().manual_seed(4)
    shapes = [(64, 7), (64, 64), (131, 256), (4, 256), (33, 1), (512, 640)]
    src = torch.randn(sum(r * c for r, c in shapes) + 11, generator=g).to(DEV)
    rows, off, dof, tiles = [], 5, 0, 0
    for r, c in shapes:
        rows.append((off, r, c, dof, tiles))
        off += r * c
        dof += r * c
        tiles += ((r + 31) // 32) * ((c + 31) // 32)
    dst = torch.full((dof,), float("nan"), device=DEV)
    ops.multi_transpose(src, dst, torch.tensor(rows, dtype=torch.int32, device=DEV), tiles)
    for (so, r, c, do, _), _s in zip(rows, shapes):
        assert torch.equal(dst[do:do + r * c].view(c, r), src[so:so + r * c].view(r, c).t())


def test_knn_overflowed_and_nan_distances_are_enumerated_in_index_order():
    """Degenerate inputs: database points so far away that the distance overflows to +inf (or is NaN) are still
    returned -- after every finite candidate, lowest index first, no index twice -- as torch.topk's K distinct
    indices would be; they used to come back as index 0 repeated."""
    ops = _ops()
    B, M, N, K = 2, 9, 70, 70
    g = torch.Generator().manual_seed(11)
    db = torch.randn(B, 3, N, generator=g)
    far = [5, 17, 18, 64, 69]
    db[:, 0, far] = 3.0e38                         # dx*dx overflows
    db[1, 1, 33] = float("nan")
    q = torch.randn(B, 3, M, generator=g)
    got = ops.knn(q.to(DEV), db.to(DEV), K).cpu().numpy()
    dist = torch.norm(q.unsqueeze(3) - db.unsqueeze(2), dim=1)
    for b in range(B):
        bad = sorted(far + ([33] if b == 1 else []))
        for m in range(M):
            row = got[b, m]
            assert sorted(row.tolist()) == list(range(N))                      # a permutation: nothing twice
            nfin = N - len(bad)
            assert row[nfin:].tolist() == bad                                    # the non-finite tail, in index order
            d = dist[b, m, torch.from_numpy(row[:nfin].astype(np.int64))]
            assert bool((d[1:] >= d[:-1]).all()) and bool(torch.isfinite(d).all())


def test_nms_rejects_negative_or_nan_radius():
    """usip_nms_f32 terminates because the selected point suppresses itself at distance 0 <= radius; a negative or
    NaN radius would spin forever on the device, so it is refused on the host."""
    ops = _ops()
    kp = torch.randn(1, 3, 32, device=DEV)
    sg = torch.rand(1, 32, device=DEV)
    for r in (-1.0, float("nan")):
        with pytest.raises(RuntimeError):
            ops.nms(kp, sg, r)
    order, count = ops.nms(kp, sg, 0.0)           # radius 0: only exact duplicates are suppressed
    assert int(count[0]) == 32


def test_som_entry_points_validate_their_limits():
    ops = _ops()
    x = torch.randn(1, 3, 64, device=DEV)
    with pytest.raises(RuntimeError):
        ops.som_assign(x, torch.randn(1, 3, 6000, device=DEV))       # node table would not fit 64 KiB of LDS
    with pytest.raises(RuntimeError):
        ops.som_cluster(x, torch.zeros(1, 63, dtype=torch.int32, device=DEV), 8)


def test_index_max_launch_geometries_agree():
    """Channel rows per workgroup (1..8), prefetch depth (1, 2, 4) and workgroup size are speed knobs: every geometry must give
    the bit-identical result, ragged N (not a multiple of the 1024*U step) and ties included."""
    ops = _ops()
    from usip_amd import _lib
    lib = _lib.lib()
    g = torch.Generator().manual_seed(3)
    try:
        for (B, C, N, K, ties) in [(2, 64, 16384, 512, False), (3, 8, 5000, 37, True), (1, 16, 1028, 5, True)]:
            data = torch.randn(B, C, N, generator=g)
            if ties:
                data = torch.round(data * 2) / 2
            idx = torch.randint(0, K, (B, N), generator=g, dtype=torch.int32)
            want = native.index_max(data.numpy(), idx.numpy(), K)
            for ch in (1, 2, 4, 8):
                for u in (1, 2, 4):
                    for th in (256, 512, 1024):
                        lib.usip_set_tuning(b"index_max_ch", ch)
                        lib.usip_set_tuning(b"index_max_unroll", u)
                        lib.usip_set_tuning(b"index_max_threads", th)
                        got = ops.index_max(data.to(DEV), idx.to(DEV), K).cpu().numpy()
                        assert np.array_equal(got, want), (B, C, N, K, ch, u, th)
    finally:
        lib.usip_set_tuning(b"index_max_ch", 0)
        lib.usip_set_tuning(b"index_max_unroll", 0)
        lib.usip_set_tuning(b"index_max_threads", 0)
    assert lib.usip_set_tuning(b"no_such_knob", 1) != 0


def test_detector_tail_elementwise_ops_match_torch():
    """csrc/head.hip against the ATen composition it replaces (models/networks.py:150-154,
    keypoint_detector.py:182-184, :196-204), values and gradients, incl. softplus beyond its threshold of 20."""
    from usip_amd import functional as Fh
    g = torch.Generator(device="cpu").manual_seed(11)
    B, M = 6, 333
    ks0 = torch.randn(B, 4, M, generator=g)
    ks0[0, 3, :5] = torch.tensor([25.0, 19.999, 20.001, -30.0, 0.0])
    centre = torch.randn(B, 3, M, generator=g).to(DEV)
    R = torch.linalg.qr(torch.randn(B // 2, 3, 3, generator=g))[0].to(DEV)
    scale = (0.5 + torch.rand(B // 2, generator=g)).to(DEV)
    shift = torch.randn(B // 2, 3, 1, generator=g).to(DEV)
    pc_d = torch.rand(B, M, generator=g).to(DEV)                          # stands for the keypoint-to-cloud distances
    wk, ws, wt = (torch.randn(s, generator=g).to(DEV) for s in ((B, 3, M), (B, M), (B // 2, 3, M)))

    def run(fused):
        ks = ks0.to(DEV).clone().requires_grad_(True)
        d = pc_d.clone().requires_grad_(True)
        if fused:
            kp, sg = Fh.detector_head(ks, centre, 0.001)
            kp_t = Fh.rigid_transform(kp[:B // 2], R, scale, shift)
        else:
            off, raw = torch.split(ks, [3, 1], dim=1)
            kp = off + centre
            sg = torch.nn.functional.softplus(raw.squeeze(1)) + 0.001
            kp_t = torch.baddbmm(shift, R * scale.view(-1, 1, 1), kp[:B // 2])
        chamfer = (kp * wk).sum() + (sg * ws).sum() + (kp_t * wt).sum()
        if fused:
            loss, on_src, on_dst = Fh.detector_loss_combine(d, chamfer, 0.7)
        else:
            on = d.view(2, -1).mean(dim=1) * 0.7
            on_src, on_dst = on[0], on[1]
            loss = chamfer + on.sum()
        loss.backward()
        return [t.detach() for t in (kp, sg, kp_t, loss, on_src, on_dst, ks.grad, d.grad)]

    for a, b, name in zip(run(True), run(False), ("kp", "sigma", "kp_t", "loss", "on_src", "on_dst", "dks", "dd")):
        scale_ = float(b.abs().max())
        assert float((a - b).abs().max()) <= 2e-6 * max(1.0, scale_), name


@pytest.mark.parametrize("case", [(2, 40, 2048, 64, "slab:14"), (1, 17, 16384, 64, "slab"), (3, 9, 300, 64, "sphere"),
                                  (2, 5, 64, 64, "sphere"), (1, 33, 5000, 16, "sphere"), (1, 8, 777, 200, "slab:10")])
def test_knn_points_matches_stable_sort_of_the_distance_rows(case):
    """usip_knn_points_f32 (RPN_Detector_KNN front end, models/networks.py:576-581): the K nearest cloud points of
    every node, nearest first, ties towards the lower index == the first K columns of a stable sort of the
    torch.norm distance row (oracle/detector.py knn_rows_canonical), bit for bit; and as a SET it is what
    torch.topk(sorted=False) picks."""
    from oracle import detector as od
    from usip_amd import ops, synth
    B, M, N, K, kind = case
    rng = np.random.default_rng(B * 1000 + N + K)
    x = np.stack([synth.make_cloud(rng, N, kind) for _ in range(B)])
    node = np.ascontiguousarray(np.stack([x[b][:, rng.permutation(N)[:M]] for b in range(B)]))
    node[0, :, 0] += 0.37                                       # a node that is not a cloud point
    got = ops.knn_points(torch.from_numpy(node).to(DEV), torch.from_numpy(x).to(DEV), K).cpu()
    dist = od.pairwise_norm(torch.from_numpy(node), torch.from_numpy(x))
    want = od.knn_rows_canonical(dist, K)
    assert torch.equal(got.long(), want)
    ref = torch.topk(dist, k=K, dim=2, largest=False, sorted=False)[1]
    assert torch.equal(torch.sort(ref, dim=2)[0], torch.sort(got.long(), dim=2)[0])


def test_knn_points_degenerate_clouds_take_the_exact_path():
    """Clouds padded by repetition (thousands of points at the same few distances) overflow the candidate list of the
    fast path; the bisection path must return the same canonical rows: ties towards the lower index."""
    from oracle import detector as od
    from usip_amd import ops
    rng = np.random.default_rng(5)
    B, M, N, K = 2, 12, 8192, 64
    base = rng.normal(0, 3, (B, 3, 40)).astype(np.float32)
    x = np.ascontiguousarray(base[:, :, rng.integers(0, 40, N)])          # 40 distinct points, ~200 copies each
    x[1, :, :3000] = rng.normal(0, 3, (3, 3000)).astype(np.float32)       # second cloud: part distinct, part repeated
    node = np.ascontiguousarray(x[:, :, :M] + np.float32(0.01))
    got = ops.knn_points(torch.from_numpy(node).to(DEV), torch.from_numpy(x).to(DEV), K).cpu()
    want = od.knn_rows_canonical(od.pairwise_norm(torch.from_numpy(node), torch.from_numpy(x)), K)
    assert torch.equal(got.long(), want)
    allsame = np.zeros((1, 3, 4096), np.float32)                           # every distance equal: rows are 0..K-1
    got = ops.knn_points(torch.zeros(1, 3, 4, device=DEV) + 1.0, torch.from_numpy(allsame).to(DEV), K).cpu()
    assert torch.equal(got, torch.arange(K, dtype=torch.int32).expand(1, 4, K))


@pytest.mark.gpu
def test_nearest_first_index_when_squared_distances_differ_but_distances_tie():
    """usip_nearest_f32 scans the squared distances without taking a sqrt and settles value and FIRST index afterwards
    (csrc/nearest.hip).  The case that scan cannot settle on its own: two candidates of the SAME lane (indices 64 apart)
    whose squared distances differ by one ulp but whose correctly rounded square roots are equal, the later one being the
    closer -- torch.min over the distances returns the EARLIER index.  Also: the same pair in different lanes, exact
    duplicates across candidate chunks, and random clouds; everything against the oracle's torch.norm + torch.min."""
    from usip_amd import ops
    from oracle import detector as od
    g = torch.Generator().manual_seed(77)
    B, Ma, Nb = 2, 70, 4096
    a = torch.randn(B, 3, Ma, generator=g) * 0.2
    b = torch.randn(B, 3, Nb, generator=g)
    b = b / b.norm(dim=1, keepdim=True) * (3.0 + torch.rand(B, 1, Nb, generator=g))     # everything at distance >= 2.5
    near = torch.tensor([1.25, 3.0 * 2.0 ** -13, 0.0])        # squared distance 1.5625 + 1 ulp, distance 1.25
    exact = torch.tensor([1.25, 0.0, 0.0])                    # squared distance 1.5625,        distance 1.25
    a[0, :, 0] = 0.0
    b[0, :, 5], b[0, :, 5 + 64] = near, exact                 # same lane: the later candidate has the smaller square
    a[0, :, 1] = torch.tensor([10.0, 0.0, 0.0])
    b[0, :, 200], b[0, :, 200 + 33] = near + a[0, :, 1], exact + a[0, :, 1]              # different lanes
    a[0, :, 2] = torch.tensor([0.0, -10.0, 0.0])
    b[0, :, 100] = b[0, :, 1124] = b[0, :, 3000] = torch.tensor([0.5, -10.0, 0.0])       # duplicates in three chunks
    a[1, :, 3] = torch.tensor([0.0, 0.0, 10.0])
    b[1, :, 900 + 64], b[1, :, 900] = near + a[1, :, 3], exact + a[1, :, 3]              # same lane, closer one first
    d, arg = ops.nearest(a.to(DEV), b.to(DEV))
    want_d, want_arg = torch.min(od.pairwise_norm(a, b), dim=2)
    assert int(want_arg[0, 0]) == 5 and float(want_d[0, 0]) == 1.25                       # the fixture is what it claims
    assert np.array_equal(d.cpu().numpy(), want_d.numpy())
    assert np.array_equal(arg.cpu().numpy(), want_arg.numpy())
    assert int(arg[0, 1]) == 200 and int(arg[0, 2]) == 100 and int(arg[1, 3]) == 900


@pytest.mark.parametrize("B,M", [(3, 37), (16, 515), (40, 513)])
def test_ball_query_coords_rows_per_wave_variants_equal_the_dist_in_pair(B, M):
    """The fused coords-in kernel picks 1, 2 or 4 node rows per wave by the number of rows of the launch (round 4: two
    rows at the detector's 16 x 512 -- 55 instead of 64 us -- because 2048 waves do not hide the L2 round trips); every
    variant, with node counts that leave a partial wave, equals usip_pairwise_dist_f32 + usip_ball_query_f32 bit for bit."""
    ops = _ops()
    g = torch.Generator().manual_seed(100 + B)
    N, K = 2052, 64
    x = (torch.rand(B, 3, N, generator=g) * 6 - 3).to(DEV)
    node = x[:, :, torch.randint(0, N, (M,), generator=g).to(DEV)].contiguous()
    d = ops.pairwise_dist(node, x)
    for r in (float(d[0, 0, 7]), 1.3, 0.0):
        assert torch.equal(ops.ball_query_coords(node, x, r, K), ops.ball_query(d, r, K))
