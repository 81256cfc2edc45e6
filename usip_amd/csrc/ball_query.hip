// usip_amd/csrc/ball_query.hip -- ball_query on gfx950 (MI355X).
//
// Semantics: models/ball_query_ext/ball_query_cuda.cu:22-46 of the reference (first K
// indices n, ascending, with dist[b,m,n] <= radius; cyclic padding; empty ball -> zeros).
//
// The reference launches M blocks x B threads, each thread walking one row serially with a
// stride of M*N floats between neighbouring threads.  Here a row is scanned by whole
// wavefronts: every lane loads 16 B (a float4) per step so one wave instruction covers 1 KiB
// of the row, several steps are in flight per wave, and the rare hits are compacted IN ORDER
// with wave ballots (no atomics, so the result is exactly the serial scan's).  The kernel is
// HBM-bound: it has to look at prefix_len(b,m) floats per row (up to the K-th hit, or all N)
// and writes K ints.  SPLIT waves share a row when there are too few rows to fill 256 CUs.
#include "common.h"

namespace {

constexpr int UNROLL = 4;   // float4 loads in flight per lane per iteration (4 KiB per wave)

// Append, in index order, the hits of one 256-element step to the wave's LDS list.
// `bits` holds this lane's 4 hit flags (element lane*4+j of the step -> bit j).
__device__ __forceinline__ int append_hits(int* __restrict__ list, int count, int K,
                                           unsigned bits, int first_index)
{
    unsigned long long m0 = __ballot(bits & 1u), m1 = __ballot(bits & 2u);
    unsigned long long m2 = __ballot(bits & 4u), m3 = __ballot(bits & 8u);
    if ((m0 | m1 | m2 | m3) == 0ull) return count;
    int p = count + usip_mbcnt(m0) + usip_mbcnt(m1) + usip_mbcnt(m2) + usip_mbcnt(m3);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (bits & (1u << j)) {
            if (p < K) list[p] = first_index + j;
            ++p;
        }
    }
    return count + __popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3);
}

// Scan [begin, end) of one row with one wave; returns min(#hits, >=K) and fills list[0..).
// VEC: row base 16-B aligned and begin/end multiples of 4.
template <bool VEC>
__device__ __forceinline__ int scan_segment(const float* __restrict__ row, int begin, int end,
                                            float radius, int K, int* __restrict__ list, int lane)
{
    int count = 0;
    if (VEC) {
        for (int base = begin; base < end && count < K; base += 256 * UNROLL) {
            float4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                int i = base + u * 256 + lane * 4;
                v[u] = (i < end) ? usip_load_stream4(row + i)
                                 : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            unsigned bits[UNROLL];
            unsigned any = 0;
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                int i = base + u * 256 + lane * 4;
                unsigned b = (v[u].x <= radius ? 1u : 0u) | (v[u].y <= radius ? 2u : 0u) |
                             (v[u].z <= radius ? 4u : 0u) | (v[u].w <= radius ? 8u : 0u);
                bits[u] = (i < end) ? b : 0u;
                any |= bits[u];
            }
            if (__ballot(any != 0u) == 0ull) continue;      // the common case: no hit in 4 KiB
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                if (count < K)
                    count = append_hits(list, count, K, bits[u], base + u * 256 + lane * 4);
            }
        }
    } else {
        for (int base = begin; base < end && count < K; base += 64) {
            int i = base + lane;
            bool hit = (i < end) && (row[i] <= radius);
            unsigned long long m = __ballot(hit);
            if (m == 0ull) continue;
            int p = count + usip_mbcnt(m);
            if (hit && p < K) list[p] = i;
            count += __popcll(m);
        }
    }
    return count;
}

// One workgroup = SPLIT waves = one row.  LDS: SPLIT lists of K ints + SPLIT counts.
template <int SPLIT, bool VEC>
__global__ __launch_bounds__(SPLIT * 64) void ball_query_kernel(
    const float* __restrict__ dist, int32_t* __restrict__ out, float radius, int K, int N)
{
    extern __shared__ __attribute__((aligned(16))) int smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long rowid = blockIdx.x;
    const float* row = dist + rowid * (long long)N;
    int* list = smem + wave * K;
    int* counts = smem + SPLIT * K;

    // segment of this wave: contiguous, multiple of 4 long when VEC
    int seg = (N + SPLIT - 1) / SPLIT;
    if (VEC) seg = (seg + 3) & ~3;
    int begin = wave * seg;
    int end = min(N, begin + seg);
    if (begin > N) begin = N;

    int count = scan_segment<VEC>(row, begin, end, radius, K, list, lane);
    if (SPLIT > 1) {
        if (lane == 0) counts[wave] = min(count, K);
    }
    __syncthreads();

    int32_t* orow = out + rowid * (long long)K;
    if (SPLIT == 1) {
        const int u = min(count, K);
        for (int j = lane; j < K; j += 64) orow[j] = (u > 0) ? list[j % u] : 0;
    } else {
        int total = 0;
#pragma unroll
        for (int s = 0; s < SPLIT; ++s) total += counts[s];
        const int u = min(total, K);
        for (int j = threadIdx.x; j < K; j += SPLIT * 64) {
            int val = 0;
            if (u > 0) {
                int r = j % u;                // r-th hit of the row, concatenating the segments
#pragma unroll
                for (int s = 0; s < SPLIT; ++s) {
                    int c = counts[s];
                    if (r >= 0 && r < c) val = smem[s * K + r];
                    r -= c;
                }
            }
            orow[j] = val;
        }
    }
}

template <int SPLIT>
int launch(const float* dist, int32_t* out, float radius, int K, long long rows, int N, hipStream_t st)
{
    const bool vec = (N % 4 == 0) && ((reinterpret_cast<uintptr_t>(dist) & 15u) == 0);
    const size_t lds = (size_t)(SPLIT * K + SPLIT) * sizeof(int);
    dim3 grid((unsigned)rows), block(SPLIT * 64);
    if (vec)
        USIP_LAUNCH((ball_query_kernel<SPLIT, true>), grid, block, lds, st, dist, out, radius, K, N);
    else
        USIP_LAUNCH((ball_query_kernel<SPLIT, false>), grid, block, lds, st, dist, out, radius, K, N);
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}

}  // namespace

extern "C" int usip_ball_query_f32(const float* dist, int32_t* out_idx, float radius, int K,
                                   int B, int M, int N, void* stream)
{
    if (B < 0 || M < 0 || N < 0 || K < 0) return USIP_EINVAL;
    if ((long long)B * M == 0 || K == 0) return USIP_OK;
    if (!dist || !out_idx) return USIP_EINVAL;
    if (K > 8192) return USIP_EINVAL;                         // LDS list: SPLIT*K*4 B <= 64 KiB
    const long long rows = (long long)B * M;
    if (rows > 0x7fffffffLL) return USIP_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    // 256 CUs x 32 wave slots: one wave per row fills the chip from ~8k rows; below that,
    // split rows over several waves (a long row per wave also limits bytes in flight).
    if (rows >= 4096 || N < 2048 || K > 2048) return launch<1>(dist, out_idx, radius, K, rows, N, st);
    if (rows >= 1024 || N < 8192) return launch<4>(dist, out_idx, radius, K, rows, N, st);
    return launch<8>(dist, out_idx, radius, K, rows, N, st);
}
