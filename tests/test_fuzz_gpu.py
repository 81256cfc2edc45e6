"""Randomised shapes and value patterns (hypothesis) for the integer-exact HIP entry points against the C oracle:
ragged sizes around the kernels' tile and vector boundaries, radii that hit stored distances exactly, duplicated
points, K larger than the number of hits."""
import numpy as np
import pytest
import torch
from hypothesis import given, settings, strategies as st

from oracle import native

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
FUZZ = settings(max_examples=30, deadline=None, derandomize=True)


def _ops():
    from usip_amd import ops
    return ops


@FUZZ
@given(B=st.integers(1, 3), M=st.integers(1, 70), N=st.integers(1, 1100), K=st.integers(1, 80),
       seed=st.integers(0, 10**6), exact=st.booleans())
def test_fuzz_ball_query_dist_in(B, M, N, K, seed, exact):
    rng = np.random.default_rng(seed)
    dist = rng.uniform(0, 4, (B, M, N)).astype(np.float32)
    r = float(dist[rng.integers(B), rng.integers(M), rng.integers(N)]) if exact else float(rng.uniform(0, 4))
    want = native.ball_query(dist, r, K)
    got = _ops().ball_query(torch.from_numpy(dist).to(DEV), r, K).cpu().numpy()
    assert np.array_equal(got, want)


@FUZZ
@given(B=st.integers(1, 3), M=st.integers(1, 40), N=st.integers(1, 1300), K=st.integers(1, 70),
       seed=st.integers(0, 10**6), dup=st.booleans())
def test_fuzz_ball_query_coords_equals_dist_then_query(B, M, N, K, seed, dup):
    rng = np.random.default_rng(seed)
    x = rng.uniform(-3, 3, (B, 3, N)).astype(np.float32)
    if dup and N > 4:
        x[:, :, N // 2:] = x[:, :, :N - N // 2]                      # duplicated points: equal distances
    node = np.ascontiguousarray(x[:, :, rng.integers(0, N, M)])
    d = native.pairwise_dist(node, x)
    # a radius that is exactly one of the stored distances half of the time
    r = float(d[0, 0, rng.integers(N)]) if seed % 2 else float(rng.uniform(0.2, 3))
    want = native.ball_query(d, r, K)
    ops = _ops()
    tn, tx = torch.from_numpy(node).to(DEV), torch.from_numpy(x).to(DEV)
    assert np.array_equal(ops.pairwise_dist(tn, tx).cpu().numpy(), d)
    assert np.array_equal(ops.ball_query_coords(tn, tx, r, K).cpu().numpy(), want)


@FUZZ
@given(B=st.integers(1, 3), C=st.integers(1, 70), N=st.integers(1, 1200), K=st.integers(1, 70),
       seed=st.integers(0, 10**6), ties=st.booleans())
def test_fuzz_index_max(B, C, N, K, seed, ties):
    rng = np.random.default_rng(seed)
    data = rng.normal(0, 1, (B, C, N)).astype(np.float32)
    if ties:
        data = np.round(data * 2) / 2                                  # many equal maxima: lowest n must win
    if seed % 3 == 0:
        data[:, :, ::3] = -2000.0                                      # below the -1000 floor
    index = rng.integers(0, K, (B, N)).astype(np.int32)
    want = native.index_max(data, index, K)
    got = _ops().index_max(torch.from_numpy(data).to(DEV), torch.from_numpy(index).to(DEV), K).cpu().numpy()
    assert np.array_equal(got, want)
