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
//             dW_c[c, j]   = sum_p dY[c, p] d[j, p]   (one partial per cloud; the caller adds the B partials; d = the
//                            decentered neighbour coordinates [B][3][M*K], gathered once: usip_group_gather_f32)
//           d feat = W_f^T . dU and dW_f = dU . feat^T are M-sized products of the shared-MLP library again.
//   Replaces, per step of the Ball detector: two group_gather launches + the 69 MB gathered tensor, the 256 x 131 forward
//   GEMM, its data gradient, its weight gradient and the gathered tensor's segment sum.
#include "common.h"

namespace {

#ifndef KL_EXP
#define KL_EXP 0                      // measurement builds (tools/build_variant.py): 1 no pass 2, 2 no coordinate gradient, 4 no dZ/Y loads
#endif
#ifndef KL_ST_NT
#define KL_ST_NT 0                    // 1: the forward's output stored non-temporal (measured: see profiles/r06z_nt_cache_policy_ab.txt)
#endif
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
                if (KL_ST_NT) __builtin_nontemporal_store(y, yb + (long long)c * P + p); else yb[(long long)c * P + p] = y;
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

// Backward.  grid (ceil(Cout / CPB), B), NT threads.  LDS: dY rows of the chunk [CPB][P].  `dcoord` [B][3][P] = the
// decentered neighbour coordinates (usip_group_gather_f32 with sub = query: formed once per step, exactly as
// models/layers.py:428-430 does), read in whole 16-B pieces.
template <int CPB, int NT>
__global__ __launch_bounds__(NT) void knn_layer_bwd_kernel(
    const float* __restrict__ dZ, const float* __restrict__ Yp, const float* __restrict__ coef4, int relu,
    const float* __restrict__ dcoord, const int32_t* __restrict__ start, const int32_t* __restrict__ perm,
    float* __restrict__ dU, float* __restrict__ dWc_part, int Cout, int N, int P)
{
    extern __shared__ __attribute__((aligned(16))) float rows[];         // [CPB][P]
    __shared__ float red[CPB][3][NT / 64];
    const int b = blockIdx.y, tid = threadIdx.x;
    // the neighbour list of this thread's first destination (pass 2) is requested first: its two dependent round trips
    // hide behind pass 1
    const int32_t* st = start + (long long)b * (N + 1);
    const int32_t* pm = perm + (long long)b * P;
    constexpr int PF = 32;                                                 // the K = 16 lists have up to ~3 K entries
    int seg0 = 0, seg1 = 0, pre[PF];
    if (!(KL_EXP & 1)) {
        if (tid < N) { seg0 = st[tid]; seg1 = st[tid + 1]; }
#pragma unroll
        for (int i = 0; i < PF; ++i) pre[i] = pm[min(seg0 + i, P - 1)];
    } else {
#pragma unroll
        for (int i = 0; i < PF; ++i) pre[i] = 0;
    }
    // pass 1: dY of the chunk's rows into LDS (whole 16-B pieces: P % 4 == 0), the coordinate weight gradient on the way.
    // UN pieces per thread and trip, every load of a trip issued before the first use (clamped addresses: no branches)
    constexpr int UN = 4;
    const float* db = dcoord + (long long)b * 3 * P;
    // the workgroup's channel chunks: blockIdx.x, + gridDim.x, ... (its neighbour list is fetched once for all of them)
    for (int c0 = blockIdx.x * CPB; c0 < Cout; c0 += gridDim.x * CPB) {
    float a1[CPB], a0[CPB], q1[CPB], q0[CPB];
#pragma unroll
    for (int c = 0; c < CPB; ++c) {
        const int ch = min(c0 + c, Cout - 1);
        a1[c] = coef4[ch]; a0[c] = coef4[Cout + ch]; q1[c] = coef4[2 * Cout + ch]; q0[c] = coef4[3 * Cout + ch];
    }
    float sw[CPB][3];
#pragma unroll
    for (int c = 0; c < CPB; ++c) { sw[c][0] = 0.f; sw[c][1] = 0.f; sw[c][2] = 0.f; }
    for (int pb = tid * 4; pb < P; pb += NT * 4 * UN) {
        float4 d4[UN][3], z4[UN][CPB], y4[UN][CPB];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int p = min(pb + u * NT * 4, P - 4);
#pragma unroll
            for (int j = 0; j < 3; ++j) d4[u][j] = (KL_EXP & 2) ? make_float4(1.f, 2.f, 3.f, 4.f) : *reinterpret_cast<const float4*>(db + (long long)j * P + p);
#pragma unroll
            for (int c = 0; c < CPB; ++c) {
                const long long off = ((long long)b * Cout + min(c0 + c, Cout - 1)) * P + p;
                z4[u][c] = (KL_EXP & 4) ? make_float4(1.f, 2.f, 3.f, (float)p) : usip_load_stream4(dZ + off);   // last use of both in the step
                y4[u][c] = (KL_EXP & 4) ? make_float4(1.f, 2.f, 3.f, (float)p) : usip_load_stream4(Yp + off);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // (hipcc sinks the loads of pieces 1.. behind the `break` tests below, i.e. it keeps ONE piece's seven loads in flight
        // and 64 registers.  Forcing all four pieces' 28 loads ahead of the first use -- branch-free clamped pieces -- was
        // measured slower: 210 registers = one workgroup per CU, 84 us against 70; with two pieces 85 us.)
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int p = pb + u * NT * 4;
            if (p >= P) break;
            const float d[3][4] = {{d4[u][0].x, d4[u][0].y, d4[u][0].z, d4[u][0].w},
                                   {d4[u][1].x, d4[u][1].y, d4[u][1].z, d4[u][1].w},
                                   {d4[u][2].x, d4[u][2].y, d4[u][2].z, d4[u][2].w}};
#pragma unroll
            for (int c = 0; c < CPB; ++c) {
                const float zv[4] = {z4[u][c].x, z4[u][c].y, z4[u][c].z, z4[u][c].w};
                const float yv[4] = {y4[u][c].x, y4[u][c].y, y4[u][c].z, y4[u][c].w};
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
    }
    // coordinate weight gradient: lanes (fixed tree), then the waves in order
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
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < NT / 64; ++w) t += red[c][j][w];
            dWc_part[((long long)b * Cout + c0 + c) * 3 + j] = t;
        }
    }
    // pass 2: segment sums in list order (the summation order of usip_segment_sum_f32: one position after the other)
    float* ub = dU + ((long long)b * Cout + c0) * N;
    for (int n = tid; n < ((KL_EXP & 1) ? 0 : N); n += NT) {
        const bool mine = n == tid;                                        // the prefetched list
        const int s0 = mine ? seg0 : st[n], s1 = mine ? seg1 : st[n + 1];
        float acc[CPB];
#pragma unroll
        for (int c = 0; c < CPB; ++c) acc[c] = 0.f;
        int j = s0;
        if (mine) {
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                if (s0 + i < s1) {
#pragma unroll
                    for (int c = 0; c < CPB; ++c) acc[c] += rows[c * P + pre[i]];
                }
            }
            j = min(s0 + PF, s1);
        }
        for (; j + 3 < s1; j += 4) {                                       // four list entries in flight
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
    __syncthreads();                                                       // the next chunk overwrites rows and red
    }   // channel chunks
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
                                           const float* dcoord, const int32_t* start, const int32_t* perm, float* dU,
                                           float* dWc_part, int B, int Cout, int N, int M, int K, void* stream)
{
    if (B < 0 || Cout < 1 || !usip_knn_layer_supported(N, M, K)) return USIP_EINVAL;
    if (B == 0) return USIP_OK;
    if (!dZ || !Y || !coef4 || !dcoord || !start || !perm || !dU || !dWc_part || B > 65535) return USIP_EINVAL;
    if (((reinterpret_cast<uintptr_t>(dZ) | reinterpret_cast<uintptr_t>(Y) | reinterpret_cast<uintptr_t>(dcoord)) & 15u) != 0)
        return USIP_EINVAL;
    const int P = M * K;
    hipStream_t st = (hipStream_t)stream;
    // two rows and 512 threads per workgroup where they fit 64 KiB of LDS (two workgroups per CU); knob r5_forms bit 7
    // (128): one row and 256 threads (five per CU by LDS) -- measured slower, 94 against 72 us at the step's shape
    // (profiles/r06aj_knn_layer_kernels.txt)
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
        return n;
    }();
    // channel chunks per workgroup (they share the cloud's neighbour lists): as many as leave >= 2 workgroups per CU
    int tpw = 1;
    { const long long chunks = (long long)B * Cout / 2; while (tpw < 8 && chunks / (tpw * 2) >= 2LL * cus) tpw *= 2; }
    const bool two = !(usip_tuning_value(USIP_TUNE_R5_FORMS) & 128) && 2LL * P * 4 <= 65536 && Cout % 2 == 0;
    if (two)
        USIP_LAUNCH((knn_layer_bwd_kernel<2, 512>), dim3(usip_ceil_div(Cout / 2, tpw), B), dim3(512), (size_t)2 * P * sizeof(float), st,
                    dZ, Y, coef4, relu, dcoord, start, perm, dU, dWc_part, Cout, N, P);
    else
        USIP_LAUNCH((knn_layer_bwd_kernel<1, 256>), dim3(usip_ceil_div(Cout, tpw), B), dim3(256), (size_t)P * sizeof(float), st,
                    dZ, Y, coef4, relu, dcoord, start, perm, dU, dWc_part, Cout, N, P);
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}
