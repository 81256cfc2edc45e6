/*
 * oracle/usip_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the two native operators on the USIP detector hot
 * path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this file's library; the product (usip_amd/) never does.
 *
 * Pinning status
 *   index_max  : PINNED -- checked bit-for-bit against the reference's own
 *                index_max.cpp CPU entry points built unmodified into
 *                oracle/_ref/ (see oracle/build_ref.py) and against the golden
 *                vectors in tests/golden/ that were produced by that build.
 *   ball_query : the reference ships no CPU implementation and no test vectors
 *                (models/ball_query_ext/ball_query.cpp:23-31 is a stub; only the
 *                CUDA kernel exists and there is no nvcc here).  The
 *                restatement below follows the CUDA kernel line by line
 *                (ball_query_cuda.cu:22-46).  It is pinned as far as the
 *                reference allows: tests/golden/ball_query_ancestor_cases.npz
 *                holds rows produced by executing the reference's own earlier
 *                statement of the kernel, the numba-CUDA function it keeps
 *                commented out at models/operations.py:295-329 (un-commented
 *                in memory and run through a thread-index shim by
 *                tests/golden/make_golden.py).  That ancestor is undefined for
 *                an empty ball; the all-zeros rule there comes from reading
 *                the CUDA kernel only ("parity unpinned" for those rows).
 *
 * Build: make -C oracle   ->  oracle/libusip_oracle.so
 */
#include <stdint.h>
#include <stddef.h>
#include <math.h>

/* index_max: follows models/index_max_ext/index_max.cpp:73-112 (forward_cpu)
 * and models/index_max_ext/index_max_cuda.cu:29-61 (shared-mem kernel):
 *   val[b,c,k] <- -1000, idx[b,c,k] <- 0
 *   for n in 0..N-1: k = index[b,n]; if data[b,c,n] > val[b,c,k]: val<-data, idx<-n
 * max_idx must hold B*C*K int32; it is fully written here (zeros where nothing won).
 * scratch must hold K floats. */
void oracle_index_max_f32(const float* data, const int32_t* index, int32_t* max_idx,
                          float* scratch, int B, int C, int N, int K)
{
    for (int b = 0; b < B; ++b) {
        for (int c = 0; c < C; ++c) {
            int32_t* out = max_idx + ((size_t)b * C + c) * K;
            const float* row = data + ((size_t)b * C + c) * N;
            const int32_t* idx = index + (size_t)b * N;
            for (int k = 0; k < K; ++k) { scratch[k] = -1000.0f; out[k] = 0; }
            for (int n = 0; n < N; ++n) {
                int k = idx[n];
                float v = row[n];
                if (v > scratch[k]) { scratch[k] = v; out[k] = n; }
            }
        }
    }
}

/* ball_query: follows models/ball_query_ext/ball_query_cuda.cu:22-46
 *   u = 0; for n in 0..N-1: if u < K: if dist[b,m,n] <= radius: out[u++] = n; else break
 *   u == 0      -> row stays all zero
 *   0 < u < K   -> out[u+i] = out[i % u] for i in 0..K-u-1
 * radius is a C float exactly as in the kernel signature (ball_query_cuda.cu:12).
 * prefix_len (optional, may be NULL) receives, per (b,m), the number of dist
 * elements the scan had to look at: position of the K-th hit + 1, or N.  It is
 * what SURVEY.md 8(d) calls the algorithmic bytes / 4 of a row. */
void oracle_ball_query_f32(const float* dist, int32_t* out_idx, float radius, int K,
                           int B, int M, int N, int32_t* prefix_len)
{
    for (int b = 0; b < B; ++b) {
        for (int m = 0; m < M; ++m) {
            const float* row = dist + ((size_t)b * M + m) * N;
            int32_t* out = out_idx + ((size_t)b * M + m) * K;
            int u = 0, n = 0;
            for (int j = 0; j < K; ++j) out[j] = 0;
            for (n = 0; n < N; ++n) {
                if (u < K) {
                    if (row[n] <= radius) { out[u] = n; u += 1; }
                } else {
                    break;
                }
            }
            if (prefix_len) prefix_len[(size_t)b * M + m] = n;
            if (u > 0 && u < K) {
                for (int i = 0; i < K - u; ++i) out[u + i] = out[i % u];
            }
        }
    }
}

/* Pairwise Euclidean distance exactly as torch.norm(a.unsqueeze(3) - b.unsqueeze(2), dim=1)
 * evaluates it on the pinned oracle platform (torch 2.10.0 CPU, probed bit-for-bit in
 * this container, see tests/golden/make_golden.py `pairwise_dist`): the squared
 * differences are accumulated in channel order 0,1,2 with FUSED multiply-add,
 *     s = fma(dz, dz, fma(dy, dy, dx*dx)),   d = sqrt(s)   (correctly rounded)
 * (models/networks.py:694-696, models/layers.py:420, models/losses.py:65,141.)
 * a: [B,3,M], b: [B,3,N] -> dist [B,M,N] = |a[:, :, m] - b[:, :, n]|. */
void oracle_pairwise_dist_f32(const float* a, const float* b, float* dist, int B, int M, int N)
{
    for (int bi = 0; bi < B; ++bi) {
        const float* ab = a + (size_t)bi * 3 * M;
        const float* bb = b + (size_t)bi * 3 * N;
        for (int m = 0; m < M; ++m) {
            float ax = ab[m], ay = ab[M + m], az = ab[2 * M + m];
            float* row = dist + ((size_t)bi * M + m) * N;
            for (int n = 0; n < N; ++n) {
                float dx = ax - bb[n];
                float dy = ay - bb[N + n];
                float dz = az - bb[2 * N + n];
                float s = dx * dx;          /* -ffp-contract=off: stays a plain multiply */
                s = fmaf(dy, dy, s);
                s = fmaf(dz, dz, s);
                row[n] = sqrtf(s);
            }
        }
    }
}
