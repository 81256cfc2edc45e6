"""GPU tests of the destination-sorted index form (csrc/segment.hip) and of the fused SOM pooling layer built on it:
the sorted segments against numpy, the segment sums against float64 scatter-adds and against the round-1 kernels,
index_max-with-values against index_max + torch.gather + mask (the reference's own composition,
models/networks.py:114-133), and the fused layer against that composition run through autograd."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ops():
    from usip_amd import ops
    return ops


@pytest.mark.parametrize("shape", [(3, 1000, 7), (16, 8192, 512), (2, 16384, 512), (4, 8192, 1024), (1, 5, 64),
                                   (2, 5000, 64), (2, 4096, 1820)])
def test_csr_by_index_is_the_counting_sort(shape):
    B, P, N = shape
    rng = np.random.default_rng(P + N)
    idx = rng.integers(0, N, (B, P)).astype(np.int32)
    idx[0, : min(P, 50)] = N - 1                                   # a heavy cell
    if N > 3:
        idx[idx == 2] = 3                                          # an empty cell
    start, perm = _ops().csr_by_index(torch.from_numpy(idx).to(DEV), N)
    start, perm = start.cpu().numpy(), perm.cpu().numpy()
    for b in range(B):
        counts = np.bincount(idx[b], minlength=N)
        assert np.array_equal(start[b], np.concatenate(([0], np.cumsum(counts))))
        assert np.array_equal(np.sort(perm[b]), np.arange(P))      # a permutation
        assert np.array_equal(idx[b][perm[b]], np.sort(idx[b]))    # grouped by cell, cells ascending
    # slots are handed out slice by slice (8 slices), 64 positions at a time: within a segment the positions of
    # different 64-blocks come in ascending order
    b = B - 1
    for n in range(0, N, max(1, N // 16)):
        seg = perm[b, start[b, n]:start[b, n + 1]]
        assert np.all(np.diff(seg // 64) >= 0)


def test_csr_by_index_leaves_out_of_range_entries_out_and_validates():
    ops = _ops()
    idx = torch.tensor([[0, 5, -1, 2, 7, 2]], dtype=torch.int32, device=DEV)
    start, perm = ops.csr_by_index(idx, 4)
    assert start.cpu().tolist() == [[0, 1, 1, 3, 3]]
    assert sorted(perm.cpu().tolist()[0][:3]) == [0, 3, 5]
    with pytest.raises(RuntimeError):
        ops.csr_by_index(torch.zeros((1, 8), dtype=torch.int32, device=DEV), 5000)      # table beyond LDS
    s, p = ops.csr_by_index(torch.zeros((0, 8), dtype=torch.int32, device=DEV), 4)
    assert s.shape == (0, 5) and p.shape == (0, 8)


@pytest.mark.parametrize("shape", [(2, 5, 64, 300, 1), (16, 64, 512, 8192, 1), (4, 128, 512, 512, 16), (3, 7, 33, 1001, 1),
                                   (2, 64, 64, 16384, 1)])
def test_segment_sum_matches_float64_scatter_add_and_round1_kernel(shape):
    B, C, N, M, K = shape                                           # gather from N sources to M*K positions
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(N + M)
    idx = torch.randint(0, N, (B, M, K), generator=g, dtype=torch.int32).to(DEV)
    coff = 3
    dout = torch.randn(B, coff + C, M, K, generator=g).to(DEV)
    assert ops.segment_sum_supported(N, M * K)
    start, perm = ops.csr_by_index(idx.view(B, M * K), N)
    got = ops.segment_sum(dout, start, perm, C, coff=coff)
    want = torch.zeros(B, C, N, dtype=torch.float64, device=DEV)
    want.scatter_add_(2, idx.view(B, 1, M * K).expand(B, C, M * K).long(), dout[:, coff:].reshape(B, C, M * K).double())
    old = ops.group_gather_backward(dout, idx, C, N, coff=coff)
    scale = float(want.abs().max())
    assert float((got.double() - want).abs().max()) <= 2e-6 * scale
    assert float((old.double() - want).abs().max()) <= 2e-6 * scale
    for _ in range(3):                                              # a fixed summation order: the same bits every run
        s2, p2 = ops.csr_by_index(idx.view(B, M * K), N)
        assert torch.equal(p2, perm) and torch.equal(ops.segment_sum(dout, s2, p2, C, coff=coff), got)


@pytest.mark.parametrize("shape", [(2, 1000, 64), (16, 8192, 512), (3, 16384, 512), (2, 5000, 64)])
def test_som_cluster_from_segments_equals_the_scan(shape):
    B, N, M = shape
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(N)
    x = (torch.randn(B, 3, N, generator=g) * 10).to(DEV)
    node = (torch.randn(B, 3, M, generator=g) * 10).to(DEV)
    node[:, :, -1] = 1e4                                            # a node nobody is assigned to
    min_idx = ops.som_assign(x, node)
    mean0, count0, dec0 = ops.som_cluster(x, min_idx, M)
    csr = ops.csr_by_index(min_idx, M)
    mean1, count1, dec1 = ops.som_cluster(x, min_idx, M, csr=csr)
    assert torch.equal(count0, count1) and int(count1[:, -1].sum()) == 0
    assert torch.equal(mean0, mean1) and torch.equal(dec0, dec1)    # double-precision sums: correctly rounded both ways


@pytest.mark.parametrize("shape", [(2, 64, 1024, 64), (16, 64, 8192, 512), (4, 128, 5000, 64), (3, 6, 1001, 33)])
def test_index_max_values_is_index_max_then_gather_then_mask(shape):
    B, C, N, K = shape
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(C + N)
    data = torch.randn(B, C, N, generator=g)
    data[:, :, ::5] = torch.round(data[:, :, ::5])                  # ties
    index = torch.randint(0, K - 1, (B, N), generator=g, dtype=torch.int32)       # node K-1 stays empty
    data[0, 0][index[0] == 1] = -2000.0                             # node 1 of row (0,0): every member below the floor
    data, index = data.to(DEV), index.to(DEV)
    count = torch.zeros(B, K, device=DEV).scatter_add_(1, index.long(), torch.ones(B, N, device=DEV)).int()
    want_idx = ops.index_max(data, index, K)
    want_val = data.gather(2, want_idx.long()) * (count > 0).float().unsqueeze(1)
    idx, val = ops.index_max_values(data, index, count, K)
    assert torch.equal(idx, want_idx)
    assert torch.equal(val, want_val)
    assert int(want_idx[0, 0, 1]) == 0 and float(val[0, 0, 1]) == float(data[0, 0, 0])
    assert float(val[:, :, K - 1].abs().max()) == 0.0
    # a channel slice of a wider tensor (the concatenated layout of the fused layer)
    wide = torch.cat((data, torch.randn_like(data)), dim=1).contiguous()
    idx2, val2 = ops.index_max_values(wide, index, count, K, C=C)
    assert torch.equal(idx2, want_idx) and torch.equal(val2, want_val)
    # backward: the gather's scatter-add, added into an existing dense gradient
    gval = torch.randn(B, C, K, device=DEV)
    dense = torch.randn(B, C, N, device=DEV)
    want = dense.double().clone()
    want.scatter_add_(2, want_idx.long(), (gval * (count > 0).float().unsqueeze(1)).double())
    got = ops.index_max_values_backward_add_(dense.clone(), gval, idx, count)
    assert float((got.double() - want).abs().max()) <= 1e-6
    # ... and as one dense pass: on top of a channel slice of a wider tensor, and on top of nothing
    wide_g = torch.cat((torch.randn_like(dense), dense), dim=1).contiguous()
    got2 = ops.index_max_values_backward(gval, idx, count, index, N, src=wide_g, soff=C)
    assert float((got2.double() - want).abs().max()) <= 1e-6
    got3 = ops.index_max_values_backward(gval, idx, count, index, N)
    assert float((got3.double() - (want - dense.double())).abs().max()) <= 1e-6


@pytest.mark.parametrize("cfg", [(2, 7, 64, 1000, 64), (4, 64, 64, 8192, 512), (2, 128, 128, 5000, 64)])
@pytest.mark.parametrize("concat", [True, False])
def test_fused_som_pool_layer_matches_the_unfused_composition(cfg, concat):
    """conv1x1 -> index_max -> gather * mask (-> broadcast -> cat), fused, against the same steps as separate
    autograd nodes (the round-1 path, itself checked against the reference's fixtures): same indices bit for bit,
    outputs equal, gradients within fp32 summation-order noise."""
    from usip_amd import functional as Fh
    B, Cin, C, N, M = cfg
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(Cin + N)
    x0 = torch.randn(B, Cin, N, generator=g).to(DEV)
    w0 = (torch.randn(C, Cin, 1, generator=g) / Cin ** 0.5).to(DEV)
    b0 = torch.randn(C, generator=g).to(DEV)
    pts = (torch.randn(B, 3, N, generator=g) * 5).to(DEV)
    node = (torch.randn(B, 3, M, generator=g) * 5).to(DEV)
    node[:, :, 0] = 1e4                                             # an empty node
    min_idx = ops.som_assign(pts, node)
    csr = ops.csr_by_index(min_idx, M)
    _, count, _ = ops.som_cluster(pts, min_idx, M, decenter=False, csr=csr)
    has = (count > 0).float().unsqueeze(1)
    gout = torch.randn((B, 2 * C, N) if concat else (B, C, M), generator=g).to(DEV)

    def run(fused):
        x, w, b = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
        if fused:
            out, idx = Fh.som_pool_layer(x, w, b, min_idx, count, csr, M, concat)
        else:
            y = Fh.conv1x1_bn_act(x, w, b, None, False)
            idx = ops.index_max(y.detach().contiguous(), min_idx, M)
            ymax = y.gather(2, idx.long()) * has
            out = torch.cat((y, Fh.cluster_broadcast(ymax, min_idx)), dim=1) if concat else ymax
        (out * gout).sum().backward()
        return out.detach(), idx, x.grad, w.grad, b.grad

    f, u = run(True), run(False)
    assert torch.equal(f[1], u[1])
    assert torch.equal(f[0], u[0])
    for a, c, name in zip(f[2:], u[2:], ("dx", "dw", "db")):
        scale = float(c.abs().max())
        assert float((a - c).abs().max()) <= 2e-5 * scale, name
