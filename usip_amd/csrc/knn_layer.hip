// usip_amd/csrc/knn_layer.hip -- the FIRST shared-MLP layer of GeneralKNNFusionModule (models/layers.py:417-431 +
// :208-216) without the gathered tensor (round 6).
//
// The reference gathers the K nearest database points of every query, decenters their coordinates, concatenates
// coordinates and features to B x (3+C) x M x K and runs conv1x1 + BatchNorm + ReLU over it.  The convolution is linear
// and the features enter it undecentered, so
//
//     Y[b, :, m, k] = W_c . (db[b, :, n] - q[b, :, m])  +  (W_f . feat[b] + bias)[:, n],      n = idx[b, m, k]
//
// The second term is a product over the N database points (512) instead of the M*K grouped positions (8192): 16 x less
// multiply work, an M-sized GEMM the shared-MLP library already has (U below).  What is left per grouped position is a
// gather of one U column, three FMAs per channel and the store: a pass bound by writing Y.  Same math as the reference up
// to fp32 summation order (the coordinate differences are formed exactly as there).
//
//   forward : Y = U[:, idx] + W_c . d (+ per-channel sum / sum^2 partials for the BatchNorm that follows)
//   backward: dY = BatchNorm-backward of (dZ, Y) (the shared library's prologue, mlp_common.h::pro_apply<PRO_BN_BWD>)
//             dU[b, c, n]  = sum over the grouped positions that picked n, in the fixed order of the CSR lists
//                            (usip_csr_by_index_i32: no float atomics)
//             dW_c[c, j]   = sum_p dY[c, p] d[j, p]   (one partial per cloud; the caller adds the B partials)
//           d feat = W_f^T . dU and dW_f = dU . feat^T are M-sized products of the shared-MLP library again.
//   Replaces, per step of the Ball detector: two group_gather launches + the 69 MB gathered tensor, the 256 x 131 forward
//   GEMM, its data gradient, its weight gradient and the gathered tensor's segment sum.
#include "common.h"

namespace {

constexpr int KL_FCH = 8;             // forward: channels per workgroup
constexpr int KL_FT = 256;            // forward: threads

// grid (ceil(Cout / KL_FCH), B).  LDS: U rows of the chunk [KL_FCH][N], database coordinates [3][N].
__global__ __launch_bounds__(KL_FT) void knn_layer_fwd_kernel(
    const float* __restrict__ U, const float* __restrict__ W, int ldw, const float* __restrict__ database,
    const float* __restrict__ query, const int32_t* __restrict__ idx, float* __restrict__ Y,
    float* __restrict__ stats, int Cout, int N, int M, int K)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];          // [KL_FCH][N] + [3][N]
    __shared__ float red[2][KL_FCH][KL_FT / 64];
    float* us = lds;
    float* ds = lds + KL_FCH * N;
    const int b = blockIdx.y, c0 = blockIdx.x * KL_FCH, tid = threadIdx.x;
    const int P = M * K;
    for (int i = tid; i < KL_FCH * N; i += KL_FT) {
        const int c = min(c0 + i / N, Cout - 1);
        us[i] = U[((long long)b * Cout + c) * N + (i % N)];
    }
    for (int i = tid; i < 3 * N; i += KL_FT) ds[i] = database[(long long)b * 3 * N + i];
    float wx[KL_FCH], wy[KL_FCH], wz[KL_FCH];
#pragma unroll
    for (int c = 0; c < KL_FCH; ++c) {
        const float* w = W + (long long)min(c0 + c, Cout - 1) * ldw;
        wx[c] = w[0]; wy[c] = w[1]; wz[c] = w[2];
    }
    float s1[KL_FCH], s2[KL_FCH];
#pragma unroll
    for (int c = 0; c < KL_FCH; ++c) { s1[c] = 0.f; s2[c] = 0.f; }
    __syncthreads();
    const int32_t* ib = idx + (long long)b * P;
    const float* qb = query + (long long)b * 3 * M;
    float* yb = Y + ((long long)b * Cout + c0) * P;
    constexpr int UNR = 4;
    for (int p0 = tid; p0 < P; p0 += KL_FT * UNR) {
        int n[UNR];
        float dx[UNR], dy[UNR], dz[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int p = min(p0 + u * KL_FT, P - 1);
            n[u] = min(max(ib[p], 0), N - 1);
            const int m = p / K;
            dx[u] = qb[m]; dy[u] = qb[M + m]; dz[u] = qb[2 * M + m];
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            dx[u] = ds[n[u]] - dx[u]; dy[u] = ds[N + n[u]] - dy[u]; dz[u] = ds[2 * N + n[u]] - dz[u];   // layers.py:428-430
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int p = p0 + u * KL_FT;
            if (p >= P) break;
#pragma unroll
            for (int c = 0; c < KL_FCH; ++c) {
                if (c0 + c >= Cout) break;
                const float t = __builtin_fmaf(wz[c], dz[u], __builtin_fmaf(wy[c], dy[u], wx[c] * dx[u]));
                const float y = us[c * N + n[u]] + t;
                yb[(long long)c * P + p] = y;
                s1[c] += y;
                s2[c] = __builtin_fmaf(y, y, s2[c]);
            }
        }
    }
    if (!stats) return;
    // per channel: lanes of a wave (fixed shuffle tree), then the four waves in order
#pragma unroll
    for (int c = 0; c < KL_FCH; ++c) {
        float a = s1[c], q = s2[c];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off); q += __shfl_down(q, off); }
        if ((tid & 63) == 0) { red[0][c][tid >> 6] = a; red[1][c][tid >> 6] = q; }
    }
    __syncthreads();
    if (tid < KL_FCH && c0 + tid < Cout) {
        const int G = gridDim.y;
        stats[(long long)(c0 + tid) * G + b] = (red[0][tid][0] + red[0][tid][1]) + (red[0][tid][2] + red[0][tid][3]);
        stats[(long long)G * Cout + (long long)(c0 + tid) * G + b] =
            (red[1][tid][0] + red[1][tid][1]) + (red[1][tid][2] + red[1][tid][3]);
    }
}

constexpr int KL_BT = 512;            // backward: threads

// grid (ceil(Cout / CPB), B).  LDS: dY rows of the chunk [CPB][P] (the 6 KB of database coordinates come from L1 / L2).
template <int CPB>
__global__ __launch_bounds__(KL_BT) void knn_layer_bwd_kernel(
    const float* __restrict__ dZ, const float* __restrict__ Yp, const float* __restrict__ coef4, int relu,
    const float* __restrict__ database, const float* __restrict__ query, const int32_t* __restrict__ idx,
    const int32_t* __restrict__ start, const int32_t* __restrict__ perm, float* __restrict__ dU,
    float* __restrict__ dWc_part, int Cout, int N, int M, int K)
{
    extern __shared__ __attribute__((aligned(16))) float rows[];         // [CPB][P]
    __shared__ float red[CPB][3][KL_BT / 64];
    const int P = M * K;
    const int b = blockIdx.y, c0 = blockIdx.x * CPB, tid = threadIdx.x;
    const float* ds = database + (long long)b * 3 * N;
    float a1[CPB], a0[CPB], q1[CPB], q0[CPB];
#pragma unroll
    for (int c = 0; c < CPB; ++c) {
        const int ch = min(c0 + c, Cout - 1);
        a1[c] = coef4[ch]; a0[c] = coef4[Cout + ch]; q1[c] = coef4[2 * Cout + ch]; q0[c] = coef4[3 * Cout + ch];
    }
    float sw[CPB][3];
#pragma unroll
    for (int c = 0; c < CPB; ++c) { sw[c][0] = 0.f; sw[c][1] = 0.f; sw[c][2] = 0.f; }
    const int32_t* ib = idx + (long long)b * P;
    const float* qb = query + (long long)b * 3 * M;
    // pass 1: dY of the chunk's rows into LDS (whole 16-B pieces: P % 4 == 0), the coordinate weight gradient on the way
    for (int p = tid * 4; p < P; p += KL_BT * 4) {
        const int4 nn = *reinterpret_cast<const int4*>(ib + p);
        const int n4[4] = {nn.x, nn.y, nn.z, nn.w};
        float d[3][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int n = min(max(n4[e], 0), N - 1), m = (p + e) / K;
            d[0][e] = ds[n] - qb[m]; d[1][e] = ds[N + n] - qb[M + m]; d[2][e] = ds[2 * N + n] - qb[2 * M + m];
        }
#pragma unroll
        for (int c = 0; c < CPB; ++c) {
            const long long off = ((long long)b * Cout + min(c0 + c, Cout - 1)) * P + p;
            const float4 z4 = usip_load_stream4(dZ + off), y4 = usip_load_stream4(Yp + off);   // last use of both
            const float zv[4] = {z4.x, z4.y, z4.z, z4.w}, yv[4] = {y4.x, y4.y, y4.z, y4.w};
            float g[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dyh = (!relu || __builtin_fmaf(yv[e], a1[c], a0[c]) > 0.0f) ? zv[e] : 0.0f;
                g[e] = __builtin_fmaf(a1[c], dyh, __builtin_fmaf(q1[c], yv[e], q0[c]));   // pro_apply<PRO_BN_BWD>
#pragma unroll
                for (int j = 0; j < 3; ++j) sw[c][j] = __builtin_fmaf(g[e], d[j][e], sw[c][j]);
            }
            *reinterpret_cast<float4*>(rows + c * P + p) = make_float4(g[0], g[1], g[2], g[3]);
        }
    }
    // coordinate weight gradient: lanes (fixed tree), then the eight waves in order
#pragma unroll
    for (int c = 0; c < CPB; ++c)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float v = sw[c][j];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
            if ((tid & 63) == 0) red[c][j][tid >> 6] = v;
        }
    __syncthreads();                                                       // rows complete, red complete
    if (tid < CPB * 3) {
        const int c = tid / 3, j = tid % 3;
        if (c0 + c < Cout) {
            const float* r = red[c][j];
            dWc_part[((long long)b * Cout + c0 + c) * 3 + j] = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        }
    }
    // pass 2: segment sums in list order (usip_segment_sum_f32's loop)
    const int32_t* st = start + (long long)b * (N + 1);
    const int32_t* pm = perm + (long long)b * P;
    float* ub = dU + ((long long)b * Cout + c0) * N;
    for (int n = tid; n < N; n += KL_BT) {
        const int s0 = st[n], s1 = st[n + 1];
        float acc[CPB];
#pragma unroll
        for (int c = 0; c < CPB; ++c) acc[c] = 0.f;
        int j = s0;
        for (; j + 3 < s1; j += 4) {
            const int p0 = pm[j], p1 = pm[j + 1], p2 = pm[j + 2], p3 = pm[j + 3];
#pragma unroll
            for (int c = 0; c < CPB; ++c) {
                const float* r = rows + c * P;
                acc[c] = (((acc[c] + r[p0]) + r[p1]) + r[p2]) + r[p3];
            }
        }
        for (; j < s1; ++j) {
            const int p = pm[j];
#pragma unroll
            for (int c = 0; c < CPB; ++c) acc[c] += rows[c * P + p];
        }
#pragma unroll
        for (int c = 0; c < CPB; ++c)
            if (c0 + c < Cout) ub[(long long)c * N + n] = acc[c];
    }
}

}  // namespace

// 1 when both directions take the shape: the forward stages (8 + 3) rows of N floats, the backward one or two rows of
// M*K floats in LDS (64 KiB each), M*K % 4 == 0 for its 16-B loads.
extern "C" int usip_knn_layer_supported(int N, int M, int K)
{
    if (N < 1 || M < 1 || K < 1) return 0;
    const long long P = (long long)M * K;
    if ((long long)(KL_FCH + 3) * N * 4 > 65536) return 0;
    if (P % 4 != 0 || P * 4 > 65536) return 0;
    return 1;
}

extern "C" int usip_knn_layer_forward_f32(const float* U, const float* W, int ldw, const float* database,
                                          const float* query, const int32_t* idx, float* Y, float* stats,
                                          int B, int Cout, int N, int M, int K, void* stream)
{
    if (B < 0 || Cout < 1 || ldw < 3 || !usip_knn_layer_supported(N, M, K)) return USIP_EINVAL;
    if (B == 0) return USIP_OK;
    if (!U || !W || !database || !query || !idx || !Y || B > 65535) return USIP_EINVAL;
    USIP_LAUNCH(knn_layer_fwd_kernel, dim3(usip_ceil_div(Cout, KL_FCH), B), dim3(KL_FT),
                (size_t)(KL_FCH + 3) * N * sizeof(float), (hipStream_t)stream, U, W, ldw, database, query, idx, Y, stats,
                Cout, N, M, K);
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}

extern "C" int usip_knn_layer_backward_f32(const float* dZ, const float* Y, const float* coef4, int relu,
                                           const float* database, const float* query, const int32_t* idx,
                                           const int32_t* start, const int32_t* perm, float* dU, float* dWc_part,
                                           int B, int Cout, int N, int M, int K, void* stream)
{
    if (B < 0 || Cout < 1 || !usip_knn_layer_supported(N, M, K)) return USIP_EINVAL;
    if (B == 0) return USIP_OK;
    if (!dZ || !Y || !coef4 || !database || !query || !idx || !start || !perm || !dU || !dWc_part || B > 65535)
        return USIP_EINVAL;
    if (((reinterpret_cast<uintptr_t>(dZ) | reinterpret_cast<uintptr_t>(Y) | reinterpret_cast<uintptr_t>(idx)) & 15u) != 0)
        return USIP_EINVAL;
    const long long P = (long long)M * K;
    hipStream_t st = (hipStream_t)stream;
    // two rows per workgroup when they fit 64 KiB of LDS
    const bool two = 2 * P * 4 <= 65536 && Cout % 2 == 0;
    if (two)
        USIP_LAUNCH((knn_layer_bwd_kernel<2>), dim3(Cout / 2, B), dim3(KL_BT), (size_t)(2 * P) * sizeof(float), st,
                    dZ, Y, coef4, relu, database, query, idx, start, perm, dU, dWc_part, Cout, N, M, K);
    else
        USIP_LAUNCH((knn_layer_bwd_kernel<1>), dim3(Cout, B), dim3(KL_BT), (size_t)P * sizeof(float), st,
                    dZ, Y, coef4, relu, database, query, idx, start, perm, dU, dWc_part, Cout, N, M, K);
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}
