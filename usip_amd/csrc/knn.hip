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

constexpr int CPL = 16;              // candidates per lane -> N <= 1024

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
    if (!query || !database || !idx || B > 65535 || N > 64 * CPL) return USIP_EINVAL;
    USIP_LAUNCH(knn_kernel, dim3(usip_ceil_div(M, 4), B), dim3(256), 0, (hipStream_t)stream,
                query, database, idx, M, N, K);
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}
