// usip_amd/csrc/segment.hip -- an index tensor sorted by destination (CSR), and the reductions that walk it (gfx950).
//
// Every "sum the things that point at me" on the path -- the backward of torch.gather in the KNN grouping
// (models/layers.py:422-426), of the SOM cluster broadcast (models/networks.py:119-125) and the cluster means
// themselves (networks.py:87-107) -- is a scatter-add in the reference (ATen float atomics).  Round 1 did them as
// LDS float atomics, which run at ~0.35 lane-adds per clock and CU when the cells collide (89-98 us for 8-17 M adds).
// Here the index tensor idx[b, p] in [0, N) is turned ONCE per step into
//     start[b, n]  (N + 1 entries)  and  perm[b, start[b,n] .. start[b,n+1])  = the positions p with idx[b,p] == n
// (a counting sort, one workgroup per cloud), after which every reduction is a GATHER: a workgroup stages a channel
// row of the source in LDS with coalesced 16-B loads, and the thread that owns destination n adds up its segment
// with plain LDS reads -- no atomics, every output written exactly once, a fixed summation order.
#include "common.h"

namespace {

constexpr int CSR_T = 512;                                   // threads of the sorting workgroup (8 waves)
constexpr int CSR_W = CSR_T / 64;

// One workgroup per cloud.  Wave w owns the w-th contiguous slice of the positions and counts into ITS OWN row of
// the table, so slots are handed out slice by slice and, inside a slice, in program order 64 positions at a time;
// the only order the hardware decides is that of lanes of ONE ds_add_rtn that hit the same cell (the LDS unit
// serialises them in a fixed lane order -- the segments come out the same on every run).
__global__ __launch_bounds__(CSR_T) void csr_build_kernel(
    const int32_t* __restrict__ idx, int32_t* __restrict__ start, int32_t* __restrict__ perm, int P, int N)
{
    extern __shared__ int cnt[];                             // [CSR_W][N] counts -> cursors, then [N + 1] scan
    int* tot = cnt + CSR_W * N;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int32_t* ib = idx + (long long)b * P;
    for (int i = tid; i < CSR_W * N; i += CSR_T) cnt[i] = 0;
    __syncthreads();
    const int per = ((P + CSR_W * 64 - 1) / (CSR_W * 64)) * 64;
    const int pbeg = min(P, wave * per), pend = min(P, pbeg + per);
    int* mine = cnt + wave * N;
    // (the loads of eight 64-position blocks are issued together: one at a time, their latency was the kernel)
    for (int p0 = pbeg + lane; p0 < pend; p0 += 8 * 64) {
        int v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (p0 + u * 64 < pend) ? ib[p0 + u * 64] : -1;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if ((unsigned)v[u] < (unsigned)N) atomicAdd(&mine[v[u]], 1);
    }
    __syncthreads();
    // per cell: exclusive scan over the waves; tot[n] = members of cell n
    for (int n = tid; n < N; n += CSR_T) {
        int s = 0;
#pragma unroll
        for (int w = 0; w < CSR_W; ++w) { const int c = cnt[w * N + n]; cnt[w * N + n] = s; s += c; }
        tot[n] = s;
    }
    __syncthreads();
    // exclusive scan of tot[0..N) in place (Hillis-Steele over LDS; N <= 2048, a handful of steps)
    for (int off = 1; off < N; off <<= 1) {
        int v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int n = tid + r * CSR_T; v[r] = (n < N && n >= off) ? tot[n - off] : 0; }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int n = tid + r * CSR_T; if (n < N) tot[n] += v[r]; }
        __syncthreads();
    }
    // tot is now the INCLUSIVE scan
    int32_t* sb = start + (long long)b * (N + 1);
    for (int n = tid; n < N; n += CSR_T) {
        const int s0 = n ? tot[n - 1] : 0;
        sb[n] = s0;
        if (n == N - 1) sb[N] = tot[n];
#pragma unroll
        for (int w = 0; w < CSR_W; ++w) cnt[w * N + n] += s0;
    }
    __syncthreads();
    int32_t* pb = perm + (long long)b * P;
    for (int p0 = pbeg + lane; p0 < pend; p0 += 8 * 64) {
        int v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (p0 + u * 64 < pend) ? ib[p0 + u * 64] : -1;
#pragma unroll
        for (int u = 0; u < 8; ++u)                                      // ascending 64-position blocks, as counted
            if ((unsigned)v[u] < (unsigned)N) pb[atomicAdd(&mine[v[u]], 1)] = p0 + u * 64;
    }
}

// dx[b, c, n] = sum over the segment of n of src[b, coff + c, p].  A workgroup stages CPB rows of one cloud.
// (Several lanes per destination with a shuffle tree at the end -- to shorten the chain of dependent perm loads of a
// long segment -- measured SLOWER: 49 vs 43 us on the SOM broadcast gradient, 28 vs 22 us on the KNN one.)
template <int CPB, bool VEC>
__global__ __launch_bounds__(512) void segment_sum_kernel(
    const float* __restrict__ src, const int32_t* __restrict__ start, const int32_t* __restrict__ perm,
    float* __restrict__ dx, int C, int N, int P, int Ctot, int coff)
{
    extern __shared__ __attribute__((aligned(16))) float rows[];        // [CPB][P]
    const int b = blockIdx.y, c0 = blockIdx.x * CPB, tid = threadIdx.x;
    const float* sb = src + ((long long)b * Ctot + coff + c0) * P;
#pragma unroll
    for (int c = 0; c < CPB; ++c) {
        const float* r = sb + (long long)min(c, C - 1 - c0) * P;
        if (VEC) {
            for (int p = tid * 4; p < P; p += 512 * 4)
                *reinterpret_cast<float4*>(rows + c * P + p) = usip_load_stream4(r + p);
        } else {
            for (int p = tid; p < P; p += 512) rows[c * P + p] = r[p];
        }
    }
    __syncthreads();
    const int32_t* st = start + (long long)b * (N + 1);
    const int32_t* pm = perm + (long long)b * P;
    float* xb = dx + ((long long)b * C + c0) * N;
    for (int n = tid; n < N; n += 512) {
        const int s0 = st[n], s1 = st[n + 1];
        float acc[CPB];
#pragma unroll
        for (int c = 0; c < CPB; ++c) acc[c] = 0.f;
        int j = s0;
        for (; j + 3 < s1; j += 4) {                                    // four positions in flight
            const int p0 = pm[j], p1 = pm[j + 1], p2 = pm[j + 2], p3 = pm[j + 3];
#pragma unroll
            for (int c = 0; c < CPB; ++c) {
                const float* r = rows + c * P;
                acc[c] = (((acc[c] + r[p0]) + r[p1]) + r[p2]) + r[p3];  // segment order
            }
        }
        for (; j < s1; ++j) {
            const int p = pm[j];
#pragma unroll
            for (int c = 0; c < CPB; ++c) acc[c] += rows[c * P + p];
        }
#pragma unroll
        for (int c = 0; c < CPB; ++c)
            if (c0 + c < C) xb[(long long)c * N + n] = acc[c];
    }
}

}  // namespace

extern "C" int usip_csr_by_index_i32(const int32_t* idx, int32_t* start, int32_t* perm, int B, int P, int N,
                                     void* stream)
{
    if (B < 0 || P < 0 || N < 1 || N > 4 * CSR_T) return USIP_EINVAL;
    if (B == 0) return USIP_OK;
    if (!start || !perm || (P > 0 && !idx)) return USIP_EINVAL;
    const size_t lds = (size_t)(CSR_W + 1) * N * sizeof(int);
    if (lds > 65536) return USIP_EINVAL;
    USIP_LAUNCH(csr_build_kernel, dim3(B), dim3(CSR_T), lds, (hipStream_t)stream, idx, start, perm, P, N);
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}

extern "C" int usip_segment_sum_supported(int N, int P) { return (P >= 1 && N >= 1 && (long long)P * 4 <= 65536) ? 1 : 0; }

extern "C" int usip_segment_sum_f32(const float* src, const int32_t* start, const int32_t* perm, float* dx,
                                    int B, int C, int N, int P, int Ctot, int coff, void* stream)
{
    if (B < 0 || C < 1 || N < 1 || P < 1 || coff < 0 || coff + C > Ctot) return USIP_EINVAL;
    if (B == 0) return USIP_OK;
    if (!src || !start || !perm || !dx || B > 65535 || !usip_segment_sum_supported(N, P)) return USIP_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const bool vec = (P % 4 == 0) && ((reinterpret_cast<uintptr_t>(src) & 15u) == 0);
    // two rows per workgroup when they fit 64 KiB of LDS and still leave >= 512 workgroups
    const bool two = (long long)P * 8 <= 65536 && (long long)B * ((C + 1) / 2) >= 512;
#define USIP_SEG(CPB_, VEC_)                                                                                    \
    USIP_LAUNCH((segment_sum_kernel<CPB_, VEC_>), dim3(usip_ceil_div(C, CPB_), B), dim3(512),                   \
                (size_t)CPB_ * P * sizeof(float), st, src, start, perm, dx, C, N, P, Ctot, coff)
    if (two) { if (vec) USIP_SEG(2, true); else USIP_SEG(2, false); }
    else     { if (vec) USIP_SEG(1, true); else USIP_SEG(1, false); }
#undef USIP_SEG
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}
