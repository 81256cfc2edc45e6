// usip_amd/csrc/knn.hip -- K nearest database points of every query point, nearest first, on gfx950
// (SURVEY 8 a-7 front end).
//
// Replaces torch.norm(query - database) -> torch.topk(K, largest=False, sorted=True)
// (models/layers.py:417-421), which materialises B x M x N distances and runs a multi-pass radix
// select.  Here one wave owns one query: every lane keeps N/64 candidate distances in registers
// (computed with the path-wide arithmetic, so they are bit-identical to torch's) and the wave extracts
// the minimum K times with a (distance, index) butterfly -- ascending distance, lower index first on
// exact ties.  N <= 1024 (node counts are 64..512); larger N is rejected (callers fall back to the
// distance-matrix route).
#include "common.h"

namespace {

constexpr int CPL_MAX = 16;          // candidates per lane -> N <= 1024

// CPL = candidate slots per lane: 4, 8 or 16 for N <= 256, 512, 1024 (every selection round scans all slots; with the
// node counts of the path -- 512 -- half of 16 slots were never candidates)
template <int CPL>
__global__ __launch_bounds__(256) void knn_kernel(
    const float* __restrict__ query, const float* __restrict__ database, int32_t* __restrict__ out,
    int M, int N, int K)
{
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.y;
    if (m >= M) return;
    const float* qb = query + (long long)b * 3 * M;
    const float* db = database + (long long)b * 3 * N;
    const float qx = qb[m], qy = qb[M + m], qz = qb[2 * M + m];
    float d[CPL];
    unsigned taken = 0;                                               // bit i: candidate slot i is out of the race
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const int j = i * 64 + lane;
        d[i] = (j < N) ? usip_dist(qx, qy, qz, db[j], db[N + j], db[2 * N + j]) : __builtin_inff();
        if (!(d[i] == d[i])) d[i] = __builtin_inff();                 // NaN sorts last, like an overflowed distance
        if (j >= N) taken |= 1u << i;
    }
    int32_t* o = out + ((long long)b * M + m) * K;
    for (int k = 0; k < K; ++k) {
        float best = __builtin_inff();
        int bj = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            const int j = i * 64 + lane;
            // ascending i: lowest index on ties; an infinite distance is still a candidate (it is selected, in
            // index order, once the finite ones are gone -- torch.topk returns distinct indices there too)
            if (!((taken >> i) & 1u) && (bj == 0x7fffffff || d[i] < best)) { best = d[i]; bj = j; }
        }
        float wbest = best;
        int wj = bj;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(wbest, off);
            const int oj = __shfl_xor(wj, off);
            if (ov < wbest || (ov == wbest && oj < wj)) { wbest = ov; wj = oj; }
        }
        if (lane == 0) o[k] = (wj == 0x7fffffff) ? 0 : wj;            // unreachable for K <= N
        if (wj != 0x7fffffff && (wj & 63) == lane) taken |= 1u << (wj >> 6);
    }
}

}  // namespace

extern "C" int usip_knn_f32(const float* query, const float* database, int32_t* idx,
                            int B, int M, int N, int K, void* stream)
{
    if (B < 0 || M < 0 || N < 1 || K < 1 || K > N) return USIP_EINVAL;
    if ((long long)B * M == 0) return USIP_OK;
    if (!query || !database || !idx || B > 65535 || N > 64 * CPL_MAX) return USIP_EINVAL;
    const dim3 grid(usip_ceil_div(M, 4), B), block(256);
    if (N <= 256) USIP_LAUNCH(knn_kernel<4>, grid, block, 0, (hipStream_t)stream, query, database, idx, M, N, K);
    else if (N <= 512) USIP_LAUNCH(knn_kernel<8>, grid, block, 0, (hipStream_t)stream, query, database, idx, M, N, K);
    else USIP_LAUNCH(knn_kernel<16>, grid, block, 0, (hipStream_t)stream, query, database, idx, M, N, K);
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}

// ------------------------------------------------------------------------------------------------
// usip_knn_points_f32: the K nearest CLOUD points of every node (models/networks.py:576-581, RPN_Detector_KNN:
// torch.norm over B x M x N, then torch.topk(k = 64, largest=False, sorted=False)) without the distance matrix.
// topk(sorted=False) leaves the order of its k picks unspecified; rows come out nearest first, ties towards the
// lower index (the order of a stable sort of the reference's distance row), which pins the SET the reference picks.
//
// One workgroup of 256 threads per node.  A thread keeps the (bit patterns of the) exact distances of its N/256
// points in registers.  Selection without a histogram (8-bit radix passes put almost every point into two or three
// exponent bins and serialise on LDS atomics):
//   1. every thread's minimum is a distinct point; a wave sorts its 64 minima (shuffle network) and takes the
//      ceil(K/4)-th smallest; tau = the largest of the four waves' values is >= the K-th smallest distance overall
//      and, for any reasonable cloud, only ~1.5 K points lie below it;
//   2. the points with distance <= tau are compacted into LDS as 48-bit keys (distance bits << 16 | n);
//   3. every candidate counts the candidates with a smaller key (keys are unique): that is its output slot.
// Degenerate inputs (thousands of points at the same few distances, e.g. clouds padded by repetition) overflow the
// candidate list; they take a bisection on the 48-bit key instead (48 counting rounds, no atomics), which finds the
// K-th key exactly, and rejoin at step 2.
namespace {

constexpr int KP_T = 256;            // threads per node
constexpr int KP_CAP = 1024;         // candidate slots in LDS

__device__ __forceinline__ unsigned kp_wave_sorted_pick(unsigned v, int lane, int pick)
{
    // ascending bitonic sort of one value per lane across the wave; returns the value lane `pick` ends up with
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1)
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            const unsigned o = __shfl_xor(v, j);
            const bool up = ((lane & k) == 0);               // this block sorts ascending
            const bool lower = ((lane & j) == 0);            // this lane keeps the smaller of the pair when ascending
            const unsigned mn = v < o ? v : o, mx = v < o ? o : v;
            v = (up == lower) ? mn : mx;
        }
    return __shfl(v, pick);
}

template <int J>
__global__ __launch_bounds__(KP_T) void knn_points_kernel(
    const float* __restrict__ node, const float* __restrict__ x, int32_t* __restrict__ out, int M, int N, int K)
{
    __shared__ unsigned long long cand[KP_CAP];
    __shared__ unsigned wtau[4];
    __shared__ unsigned wcnt[4];
    __shared__ unsigned ncand;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = blockIdx.x, b = blockIdx.y;
    const float* nb_ = node + (long long)b * 3 * M;
    const float* xb = x + (long long)b * 3 * N;
    const float qx = nb_[m], qy = nb_[M + m], qz = nb_[2 * M + m];

    unsigned d[J];
    unsigned tmin = 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int n = j * KP_T + tid;
        unsigned key = 0xFFFFFFFFu;                           // n >= N: never a candidate
        if (n < N) {
            const float v = usip_dist(qx, qy, qz, xb[n], xb[N + n], xb[2 * N + n]);
            key = __builtin_bit_cast(unsigned, v);            // v >= +0: the bit pattern orders like the value;
            if (!(v == v)) key = 0xFFC00000u;                 // NaN after +inf, as torch.sort places it
        }
        d[j] = key;
        tmin = key < tmin ? key : tmin;
    }
    if (tid == 0) ncand = 0;
    const int q = (K + 3) >> 2;                               // per-wave quota (K <= 256)
    const unsigned tw = kp_wave_sorted_pick(tmin, lane, q - 1);
    if (lane == 0) wtau[wave] = tw;
    __syncthreads();
    unsigned tau = wtau[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) tau = wtau[w] > tau ? wtau[w] : tau;

    auto compact = [&](unsigned long long bound) {            // candidates: key48 <= bound
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int n = j * KP_T + tid;
            const unsigned long long k48 = ((unsigned long long)d[j] << 16) | (unsigned)n;
            if (n < N && k48 <= bound) {
                const unsigned slot = atomicAdd(&ncand, 1u);
                if (slot < (unsigned)KP_CAP) cand[slot] = k48;
            }
        }
    };
    compact(((unsigned long long)tau << 16) | 0xFFFFull);
    __syncthreads();
    unsigned c = ncand;
    if (c > (unsigned)KP_CAP) {                               // uniform: degenerate input, exact K-th key by bisection
        unsigned long long lo = 0, hi = (1ull << 48) - 1;     // smallest v with #(key48 <= v) >= K
        while (lo < hi) {
            const unsigned long long mid = lo + ((hi - lo) >> 1);
            unsigned cnt = 0;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int n = j * KP_T + tid;
                const unsigned long long k48 = ((unsigned long long)d[j] << 16) | (unsigned)n;
                cnt += (n < N && k48 <= mid) ? 1u : 0u;
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
            __syncthreads();                                  // previous round's readers are done with wcnt
            if (lane == 0) wcnt[wave] = cnt;
            __syncthreads();
            const unsigned total = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
            if (total >= (unsigned)K) hi = mid; else lo = mid + 1;
        }
        __syncthreads();
        if (tid == 0) ncand = 0;
        __syncthreads();
        compact(lo);                                          // exactly K keys (they are unique)
        __syncthreads();
        c = ncand;
    }
    int32_t* o = out + ((long long)b * M + m) * K;
    for (unsigned i = tid; i < c; i += KP_T) {
        const unsigned long long mine = cand[i];
        unsigned rank = 0;
        for (unsigned t = 0; t < c; ++t) rank += (cand[t] < mine) ? 1u : 0u;
        if (rank < (unsigned)K) o[rank] = (int32_t)(mine & 0xFFFFull);
    }
}

}  // namespace

// node [B][3][M], x [B][3][N] -> idx int32 [B][M][K]: the K points nearest to every node, nearest first, ties towards
// the lower index (distances: the path-wide torch.norm arithmetic).  K <= min(N, 256), N <= 16384.
// Replaces models/networks.py:576-581 (torch.norm + torch.topk(sorted=False)) of RPN_Detector_KNN.
extern "C" int usip_knn_points_f32(const float* node, const float* x, int32_t* idx, int B, int M, int N, int K,
                                   void* stream)
{
    if (B < 0 || M < 0 || N < 1 || K < 1 || K > N || K > 256 || N > 64 * KP_T) return USIP_EINVAL;
    if ((long long)B * M == 0) return USIP_OK;
    if (!node || !x || !idx || B > 65535) return USIP_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)M, (unsigned)B), block(KP_T);
    const int per = usip_ceil_div(N, KP_T);
    if (per <= 8) USIP_LAUNCH(knn_points_kernel<8>, grid, block, 0, st, node, x, idx, M, N, K);
    else if (per <= 16) USIP_LAUNCH(knn_points_kernel<16>, grid, block, 0, st, node, x, idx, M, N, K);
    else if (per <= 32) USIP_LAUNCH(knn_points_kernel<32>, grid, block, 0, st, node, x, idx, M, N, K);
    else USIP_LAUNCH(knn_points_kernel<64>, grid, block, 0, st, node, x, idx, M, N, K);
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}
