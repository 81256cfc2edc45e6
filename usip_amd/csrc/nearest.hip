// usip_amd/csrc/nearest.hip -- fused pairwise-distance minimum (value + arg) on gfx950.
//
// This is the O(M*N) core of both chamfer losses of the reference:
//   SingleSideChamferLoss_Brute (models/losses.py:132-143): min_n |kp[b,:,m] - pc[b,:,n]|
//   ChamferLoss_Brute           (models/losses.py:62-66,81-87): row minima + arg minima of the
//                               keypoint x keypoint matrix, in both directions.
// The reference materialises B x 3 x M x N (805 MB for keypoint-on-pc at B=8) and B x M x N
// tensors; here nothing but the two small point sets is read and B x M (value, index) pairs
// are written.  The distance is the path-wide one (FMA chain + correctly rounded sqrt) so the
// minimum VALUE is bit-identical to torch's.  The arg-minimum is the FIRST index attaining the
// minimum of the sqrt'ed distances, as torch.min(dim) returns.  sqrt is monotone, so only a
// candidate with a strictly smaller squared distance can win, and the (expensive, correctly
// rounded) sqrt is evaluated just for those: O(log N) times per lane instead of N.
#include "common.h"

namespace {

constexpr int R = 4;                 // query points per wave

// The candidate set is split into gridDim.z contiguous chunks (more workgroups when there are few queries
// and many candidates); chunk z writes its (distance, index) to slot z, nearest_merge_kernel takes the
// minimum over chunks, lower chunk (= lower index) first on equal distances.
__global__ __launch_bounds__(256) void nearest_kernel(
    const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ min_d,
    int32_t* __restrict__ arg, int Ma, int Nb, int chunk, long long slot_stride)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bi = blockIdx.y;
    const int i0 = (blockIdx.x * 4 + wave) * R;
    if (i0 >= Ma) return;
    const int jbeg = blockIdx.z * chunk, jend = min(Nb, jbeg + chunk);
    min_d += (long long)blockIdx.z * slot_stride;
    arg += (long long)blockIdx.z * slot_stride;
    const float* ab = a + (long long)bi * 3 * Ma;
    const float* bb = b + (long long)bi * 3 * Nb;
    float ax[R], ay[R], az[R], best_s[R], best_d[R];
    int best_j[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = min(i0 + r, Ma - 1);
        ax[r] = ab[i]; ay[r] = ab[Ma + i]; az[r] = ab[2 * Ma + i];
        best_s[r] = __builtin_inff(); best_d[r] = __builtin_inff(); best_j[r] = 0x7fffffff;
    }
    static_assert(R % 2 == 0, "queries are processed in pairs");
    usip_f32x2 qx[R / 2], qy[R / 2], qz[R / 2];
#pragma unroll
    for (int r = 0; r < R / 2; ++r) {
        qx[r] = usip_f32x2{ax[2 * r], ax[2 * r + 1]};
        qy[r] = usip_f32x2{ay[2 * r], ay[2 * r + 1]};
        qz[r] = usip_f32x2{az[2 * r], az[2 * r + 1]};
    }
    // The reference-exact rule -- the correctly rounded sqrt of every candidate that improves a lane's running minimum of
    // the SQUARED distance, first index kept on equal distances -- cost 70 of the kernel's 95 us: some lane of the wave
    // improves in more than half of the iterations, and hipcc predicates the sqrt into all of them.  Instead (r03):
    //   * one branch-free scan keeps, per lane and query, the strict running minimum s0 of the squared distance with its
    //     index j0 and the minimum it replaced, s1.  min_j sqrt(s_j) = sqrt(min_j s_j) (the rounded sqrt is monotone), and
    //     every candidate whose sqrt rounds to that value has s_j <= smin (1 + 2^-22 + ...): only candidates with
    //     s_j <= thr = smin (1 + 2^-20) can be the answer.  The FIRST of them is a strict running minimum of its lane
    //     (everything before it in the lane lies above thr), and the qualifying running minima of a lane are the tail
    //     of its sequence: if s1 > thr the lane's only qualifying candidate is (s0, j0);
    //   * a wave in which some lane has s1 <= thr as well (two candidates within 1e-6 of the minimum in one lane, or an
    //     empty / all-NaN chunk) settles the question by the exact scan below.
    // UNR candidates per lane and iteration, all their loads issued before the first is used (one candidate per iteration
    // was a chain of L2 latencies); the tail repeats the chunk's last candidate, which is never a strict improvement.
    constexpr int UNR = 4;
    float s1[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { best_s[r] = __builtin_inff(); s1[r] = __builtin_inff(); }
    for (int j0 = jbeg + lane; j0 < jend; j0 += 64 * UNR) {
        float bx[UNR], by[UNR], bz[UNR];
        int jj[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            jj[u] = min(j0 + 64 * u, jend - 1);
            bx[u] = bb[jj[u]]; by[u] = bb[Nb + jj[u]]; bz[u] = bb[2 * Nb + jj[u]];
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const usip_f32x2 cx = {bx[u], bx[u]}, cy = {by[u], by[u]}, cz = {bz[u], bz[u]};
#pragma unroll
            for (int r2 = 0; r2 < R / 2; ++r2) {
                const usip_f32x2 s2 = usip_sqdist2(qx[r2], qy[r2], qz[r2], cx, cy, cz);   // two queries per packed instruction
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int r = 2 * r2 + h;
                    const float sq = h ? s2.y : s2.x;
                    const bool lt = sq < best_s[r];                                // false for NaN
                    s1[r] = lt ? best_s[r] : s1[r];
                    best_s[r] = lt ? sq : best_s[r];
                    best_j[r] = lt ? jj[u] : best_j[r];
                }
            }
        }
    }
    float thr[R];
    bool ambiguous = false;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float m = best_s[r];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fminf(m, __shfl_xor(m, off));
        thr[r] = m * 1.00000095367431640625f;                  // 1 + 2^-20
        ambiguous = ambiguous || s1[r] <= thr[r];
        if (best_s[r] <= thr[r]) best_d[r] = sqrtf(best_s[r]);  // this lane's candidate (j0 is its index already)
        else best_j[r] = 0x7fffffff;
    }
    if (__any(ambiguous)) {
        // exact scan of the chunk: every candidate with s <= thr, in ascending order within a lane
#pragma unroll
        for (int r = 0; r < R; ++r) { best_d[r] = __builtin_inff(); best_j[r] = 0x7fffffff; }
        for (int j = jbeg + lane; j < jend; j += 64) {
            const float bx = bb[j], by = bb[Nb + j], bz = bb[2 * Nb + j];
            const usip_f32x2 cx = {bx, bx}, cy = {by, by}, cz = {bz, bz};
#pragma unroll
            for (int r2 = 0; r2 < R / 2; ++r2) {
                const usip_f32x2 s2 = usip_sqdist2(qx[r2], qy[r2], qz[r2], cx, cy, cz);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int r = 2 * r2 + h;
                    const float sq = h ? s2.y : s2.x;
                    if (sq <= thr[r]) {
                        const float d = sqrtf(sq);
                        if (d < best_d[r]) { best_d[r] = d; best_j[r] = j; }
                    }
                }
            }
        }
    }
    // wave reduction: smaller distance wins, then lower index (== first occurrence overall)
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float d = best_d[r];
        int j = best_j[r];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float od = __shfl_down(d, off);
            const int oj = __shfl_down(j, off);
            if (od < d || (od == d && oj < j)) { d = od; j = oj; }
        }
        if (lane == 0 && i0 + r < Ma) {
            if (j == 0x7fffffff) j = 0;                          // all-NaN / empty row
            min_d[(long long)bi * Ma + i0 + r] = d;
            arg[(long long)bi * Ma + i0 + r] = j;
        }
    }
}

__global__ __launch_bounds__(256) void nearest_merge_kernel(
    const float* __restrict__ pd, const int32_t* __restrict__ pj, float* __restrict__ min_d,
    int32_t* __restrict__ arg, long long n, int chunks)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float d = pd[i];
    int j = pj[i];
    for (int c = 1; c < chunks; ++c) {
        const float od = pd[(long long)c * n + i];
        if (od < d) { d = od; j = pj[(long long)c * n + i]; }      // strict: the lower chunk keeps ties
    }
    min_d[i] = d;
    arg[i] = j;
}

// General-dimension variant for the descriptor loss (DescPairScanLoss, models/losses.py:207-218): points are
// C-dimensional descriptors [B][C][M] (C = 128, M = 256 keypoints).  One wave per query; every lane keeps TJ
// candidate partial sums in registers while the channel loop streams b[c][j] coalesced along j.  The problem is
// tiny (B*M = 1024 queries), so the kernel is bound by load LATENCY, not bandwidth: CU channels are loaded
// together before any of them is used (one exposed L2 round trip per CU channels instead of per channel;
// 141 -> ~25 us at C = 128, Nb = 256).
constexpr int TJ_MAX = 16;           // candidates per lane -> Nb <= 1024

template <int TJ, int CU>
__global__ __launch_bounds__(256) void nearest_nd_kernel(
    const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ min_d,
    int32_t* __restrict__ arg, int C, int Ma, int Nb)
{
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: a[c][i] is a scalar load
    const int bi = blockIdx.y;
    if (i >= Ma) return;
    const float* ab = a + (long long)bi * C * Ma;
    const float* bb = b + (long long)bi * C * Nb;
    int jc[TJ];                                   // clamped candidate index: loads are branch-free
#pragma unroll
    for (int t = 0; t < TJ; ++t) jc[t] = min(t * 64 + lane, Nb - 1);
    float s[TJ];
#pragma unroll
    for (int t = 0; t < TJ; ++t) s[t] = 0.f;
    int c = 0;
    for (; c + CU <= C; c += CU) {
        float av[CU], bv[CU][TJ];
#pragma unroll
        for (int u = 0; u < CU; ++u) {
            av[u] = ab[(long long)(c + u) * Ma + i];
#pragma unroll
            for (int t = 0; t < TJ; ++t) bv[u][t] = bb[(long long)(c + u) * Nb + jc[t]];
        }
#pragma unroll
        for (int u = 0; u < CU; ++u)              // channel order as in the plain loop: same sums bit for bit
#pragma unroll
            for (int t = 0; t < TJ; ++t) {
                const float df = av[u] - bv[u][t];
                s[t] = __builtin_fmaf(df, df, s[t]);
            }
    }
    for (; c < C; ++c) {
        const float av = ab[(long long)c * Ma + i];
#pragma unroll
        for (int t = 0; t < TJ; ++t) {
            const float df = av - bb[(long long)c * Nb + jc[t]];
            s[t] = __builtin_fmaf(df, df, s[t]);
        }
    }
    float best = __builtin_inff();
    int bj = 0x7fffffff;
#pragma unroll
    for (int t = 0; t < TJ; ++t) {
        const int j = t * 64 + lane;
        if (j < Nb) {
            const float d = sqrtf(s[t]);
            if (d < best) { best = d; bj = j; }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float od = __shfl_xor(best, off);
        const int oj = __shfl_xor(bj, off);
        if (od < best || (od == best && oj < bj)) { best = od; bj = oj; }
    }
    if (lane == 0) {
        min_d[(long long)bi * Ma + i] = best;
        arg[(long long)bi * Ma + i] = (bj == 0x7fffffff) ? 0 : bj;
    }
}

// Backward of (min distance, arg-min): ga[b,:,i] = gd[b,i] * (a_i - b_J) / d (0 where d == 0, as the
// sub-gradient of torch.norm at zero).
__global__ __launch_bounds__(256) void nearest_bwd_kernel(
    const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ d,
    const int32_t* __restrict__ arg, const float* __restrict__ gd, float* __restrict__ ga, int C, int Ma, int Nb)
{
    const int bi = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Ma) return;
    const long long q = (long long)bi * Ma + i;
    const int j = arg[q];
    const float dist = d[q];
    const float sc = dist > 0.f ? gd[q] / dist : 0.f;
    const float* ab = a + (long long)bi * C * Ma;
    const float* bb = b + (long long)bi * C * Nb;
    for (int c = 0; c < C; ++c)
        ga[((long long)bi * C + c) * Ma + i] = (ab[(long long)c * Ma + i] - bb[(long long)c * Nb + j]) * sc;
}

// gb[b,:,j] = - sum_{i: arg[b,i] == j} ga[b,:,i]: several queries may share a partner.  A deterministic
// segmented sum instead of float atomics (training runs are reproducible bit for bit): a workgroup owns 64
// partners j of one cloud, its four waves each scan a quarter of the queries (arg and ga staged in LDS, read as
// 16-B vectors), quarters combined in a fixed order.  Every element of gb is written.
constexpr int BWD_CHUNK = 1024;
constexpr int BWD_CPB = 4;                       // channels per workgroup (descriptors: C = 128 -> 32 z-slices)
__global__ __launch_bounds__(256) void nearest_bwd_partner_kernel(
    const int32_t* __restrict__ arg, const float* __restrict__ ga, float* __restrict__ gb, int C, int Ma, int Nb)
{
    __shared__ __attribute__((aligned(16))) int s_arg[BWD_CHUNK];
    __shared__ __attribute__((aligned(16))) float s_val[BWD_CHUNK];
    __shared__ float part[4][64];
    const int bi = blockIdx.y, tl = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + tl;
    const int32_t* ar = arg + (long long)bi * Ma;
    const int cend = min(C, (int)(blockIdx.z + 1) * BWD_CPB);
    for (int c = blockIdx.z * BWD_CPB; c < cend; ++c) {
        const float* gr = ga + ((long long)bi * C + c) * Ma;
        float acc = 0.f;
        for (int c0 = 0; c0 < Ma; c0 += BWD_CHUNK) {
            const int len = min(BWD_CHUNK, Ma - c0);
            const int len4 = ((len + 15) / 16) * 4;
            __syncthreads();
            for (int i = threadIdx.x; i < 4 * len4; i += 256) {
                s_arg[i] = (i < len) ? ar[c0 + i] : -1;
                s_val[i] = (i < len) ? gr[c0 + i] : 0.f;
            }
            __syncthreads();
            for (int i = q * len4; i < (q + 1) * len4; i += 4) {
                const int4 k = *reinterpret_cast<const int4*>(&s_arg[i]);
                const float4 v = *reinterpret_cast<const float4*>(&s_val[i]);
                acc += (k.x == j) ? v.x : 0.f;
                acc += (k.y == j) ? v.y : 0.f;
                acc += (k.z == j) ? v.z : 0.f;
                acc += (k.w == j) ? v.w : 0.f;
            }
        }
        part[q][tl] = acc;
        __syncthreads();
        if (q == 0 && j < Nb)
            gb[((long long)bi * C + c) * Nb + j] = -((part[0][tl] + part[1][tl]) + (part[2][tl] + part[3][tl]));
    }
}

}  // namespace

extern "C" int usip_nearest_backward_f32(const float* a, const float* b, const float* d, const int32_t* arg,
                                         const float* gd, float* ga, float* gb, int B, int C, int Ma, int Nb,
                                         void* stream)
{
    if (B < 0 || Ma < 0 || Nb < 1 || C < 1) return USIP_EINVAL;
    if ((long long)B * Ma == 0) return USIP_OK;
    if (!a || !b || !d || !arg || !gd || !ga || B > 65535) return USIP_EINVAL;
    USIP_LAUNCH(nearest_bwd_kernel, dim3(usip_ceil_div(Ma, 256), B), dim3(256), 0, (hipStream_t)stream,
                a, b, d, arg, gd, ga, C, Ma, Nb);
    USIP_LAUNCH_CHECK();
    if (gb) {
        USIP_LAUNCH(nearest_bwd_partner_kernel, dim3(usip_ceil_div(Nb, 64), B, usip_ceil_div(C, BWD_CPB)), dim3(256),
                    0, (hipStream_t)stream,
                    arg, ga, gb, C, Ma, Nb);
        USIP_LAUNCH_CHECK();
    }
    return USIP_OK;
}

extern "C" int usip_nearest_nd_f32(const float* a, const float* b, float* min_d, int32_t* arg,
                                   int B, int C, int Ma, int Nb, void* stream)
{
    if (B < 0 || C < 1 || Ma < 0 || Nb < 1 || Nb > 64 * TJ_MAX) return USIP_EINVAL;
    if ((long long)B * Ma == 0) return USIP_OK;
    if (!a || !b || !min_d || !arg || B > 65535) return USIP_EINVAL;
    const dim3 grid(usip_ceil_div(Ma, 4), B), block(256);
    if (Nb <= 256)
        USIP_LAUNCH((nearest_nd_kernel<4, 8>), grid, block, 0, (hipStream_t)stream, a, b, min_d, arg, C, Ma, Nb);
    else
        USIP_LAUNCH((nearest_nd_kernel<TJ_MAX, 2>), grid, block, 0, (hipStream_t)stream, a, b, min_d, arg, C, Ma, Nb);
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}

extern "C" long long usip_nearest_workspace(int B, int Ma, int Nb)
{
    // floats AND ints: chunks * B * Ma of each (0 when one chunk suffices)
    const long long groups = (long long)B * ((Ma + 4 * R - 1) / (4 * R));
    int chunks = 1;
    while (groups * chunks < 1024 && Nb / (chunks * 2) >= 1024) chunks *= 2;
    return chunks > 1 ? (long long)chunks * B * Ma : 0;
}

extern "C" int usip_nearest_f32(const float* a, const float* b, float* min_d, int32_t* arg,
                                float* ws_d, int32_t* ws_j, int B, int Ma, int Nb, void* stream)
{
    if (B < 0 || Ma < 0 || Nb < 1) return USIP_EINVAL;
    if ((long long)B * Ma == 0) return USIP_OK;
    if (!a || !b || !min_d || !arg || B > 65535) return USIP_EINVAL;
    const long long n = (long long)B * Ma;
    const long long ws = usip_nearest_workspace(B, Ma, Nb);
    hipStream_t st = (hipStream_t)stream;
    if (ws == 0 || !ws_d || !ws_j) {
        dim3 grid(usip_ceil_div(Ma, 4 * R), B, 1), block(256);
        USIP_LAUNCH(nearest_kernel, grid, block, 0, st, a, b, min_d, arg, Ma, Nb, Nb, 0LL);
        USIP_LAUNCH_CHECK();
        return USIP_OK;
    }
    const int chunks = (int)(ws / n);
    const int chunk = ((Nb + chunks - 1) / chunks + 63) / 64 * 64;
    dim3 grid(usip_ceil_div(Ma, 4 * R), B, chunks), block(256);
    USIP_LAUNCH(nearest_kernel, grid, block, 0, st, a, b, ws_d, ws_j, Ma, Nb, chunk, n);
    USIP_LAUNCH_CHECK();
    USIP_LAUNCH(nearest_merge_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ws_d, ws_j, min_d, arg, n,
                chunks);
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}
