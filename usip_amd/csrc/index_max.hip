// usip_amd/csrc/index_max.hip -- SOM-node arg-max (index_max) on gfx950 (MI355X).
//
// Semantics: models/index_max_ext/index_max_cuda.cu:29-61 / index_max.cpp:98-109 of the
// reference: per (b,c,k) the LOWEST n among the points assigned to node k (index[b,n]==k)
// whose value is the maximum, if that maximum is strictly above -1000; otherwise 0.
//
// The reference gives each (b,c) row to ONE thread that walks N points serially (C blocks x
// B threads, uncoalesced).  Here a workgroup owns CH channel rows of one cloud: all 256 lanes
// stream the rows with 16-B loads, U steps deep (coalesced, the index row is read once for the CH rows)
// and fold every point into a per-node table in LDS with ONE 64-bit ds_max per point:
//     key = (order-preserving bits of the value) << 32 | ~n
// so the LDS atomic max implements "greater value wins, then lower n wins" exactly, with no
// ordering dependence between lanes -- bit-identical to the serial loop.  The table starts at
// key(-1000, n = none); values <= -1000 and NaN are filtered before the atomic (strict >).
// HBM-bound by design: data + index in, B*C*K ints out.
#include "common.h"

namespace {

__device__ __forceinline__ unsigned ordered_bits(float v)
{
    unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);       // monotone: a < b  <=>  ord(a) < ord(b)
}

constexpr float FLOOR = -1000.0f;                               // index_max_cuda.cu:37

// (Tried and removed: reading the value half of the entry first and skipping the atomic for points below the node's
// current maximum.  It made the kernel SLOWER -- 23 vs 17 us at C=64 -- because the dependent LDS read serialises
// what the fire-and-forget atomics overlap; the kernel is not bound by LDS-atomic throughput but by the per-CU
// streaming rate (~25 GB/s per CU, the same rate ball_query reaches), of which the index row re-read by every
// channel group is a fixed tax: 1/3 of the bytes at two rows per workgroup.)
__device__ __forceinline__ void fold(unsigned long long* table, int k, float v, int n)
{
    if (v > FLOOR) {                                             // false for NaN as well
        v += 0.0f;                                               // -0.0 -> +0.0: they compare equal
        unsigned long long key = ((unsigned long long)ordered_bits(v) << 32) | (unsigned)(~n);
        atomicMax(&table[k], key);                               // LDS ds_max_u64, no return
    }
}

// T = threads per workgroup (256 / 512 / 1024).  U = prefetch depth: the loads of U consecutive 4*T-point steps (U index vectors + U*CH value vectors per lane)
// are all issued before the first LDS atomic, so a workgroup keeps U*(CH+1)*4 KiB in flight instead of (CH+1)*4.
// (An atomic in the loop body stops the compiler from hoisting the next step's loads on its own.)
template <int CH, int U, bool VEC, int T>
__global__ __launch_bounds__(T) void index_max_kernel(
    const float* __restrict__ data, const int32_t* __restrict__ index, int32_t* __restrict__ out,
    int C, int N, int K)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long table[];   // [CH][K]
    const int cgroups = C / CH;
    const int b = blockIdx.x / cgroups;
    const int c0 = (blockIdx.x % cgroups) * CH;
    const unsigned long long init = ((unsigned long long)ordered_bits(FLOOR) << 32) | 0xffffffffull;
    for (int i = threadIdx.x; i < CH * K; i += T) table[i] = init;
    __syncthreads();

    const int32_t* idx = index + (long long)b * N;
    const float* rows = data + ((long long)b * C + c0) * N;
    if (VEC) {
        constexpr int STEP = 4 * T;                          // points per workgroup step
        for (int n0 = threadIdx.x * 4; n0 < N; n0 += STEP * U) {
            int4 k4[U];
            float4 v[U][CH];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                // past the end: re-read the lane's first vector (always valid) and drop it below
                const int n = (n0 + u * STEP < N) ? n0 + u * STEP : n0;
                k4[u] = *reinterpret_cast<const int4*>(idx + n);
#pragma unroll
                for (int c = 0; c < CH; ++c)
                    v[u][c] = usip_load_stream4(rows + (long long)c * N + n);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int n = n0 + u * STEP;
                if (n < N) {
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        unsigned long long* t = table + c * K;
                        fold(t, k4[u].x, v[u][c].x, n);
                        fold(t, k4[u].y, v[u][c].y, n + 1);
                        fold(t, k4[u].z, v[u][c].z, n + 2);
                        fold(t, k4[u].w, v[u][c].w, n + 3);
                    }
                }
            }
        }
    } else {
        for (int n = threadIdx.x; n < N; n += T) {
            int k = idx[n];
#pragma unroll
            for (int c = 0; c < CH; ++c) fold(table + c * K, k, rows[(long long)c * N + n], n);
        }
    }
    __syncthreads();
    int32_t* o = out + ((long long)b * C + c0) * K;
    for (int i = threadIdx.x; i < CH * K; i += T) {
        unsigned long long key = table[i];
        o[i] = (key == init) ? 0 : (int32_t)(~(unsigned)key);
    }
}

template <int CH, int U, int T>
int launch(const float* data, const int32_t* index, int32_t* out, int B, int C, int N, int K, hipStream_t st)
{
    const bool vec = (N % 4 == 0) && ((reinterpret_cast<uintptr_t>(data) & 15u) == 0) &&
                     ((reinterpret_cast<uintptr_t>(index) & 15u) == 0);
    const size_t lds = (size_t)CH * K * sizeof(unsigned long long);
    dim3 grid((unsigned)(B * (C / CH))), block(T);
    if (vec)
        USIP_LAUNCH((index_max_kernel<CH, U, true, T>), grid, block, lds, st, data, index, out, C, N, K);
    else
        USIP_LAUNCH((index_max_kernel<CH, 1, false, 256>), grid, dim3(256), lds, st, data, index, out, C, N, K);
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}

template <int CH>
int launch_u(int U, int T, const float* data, const int32_t* index, int32_t* out, int B, int C, int N, int K, hipStream_t st)
{
    if (T >= 1024) return U >= 2 ? launch<CH, 2, 1024>(data, index, out, B, C, N, K, st)
                                 : launch<CH, 1, 1024>(data, index, out, B, C, N, K, st);
    if (T >= 512) return U >= 2 ? launch<CH, 2, 512>(data, index, out, B, C, N, K, st)
                                : launch<CH, 1, 512>(data, index, out, B, C, N, K, st);
    if (U >= 4 && CH <= 4) return launch<CH, 4, 256>(data, index, out, B, C, N, K, st);
    if (U >= 2) return launch<CH, 2, 256>(data, index, out, B, C, N, K, st);
    return launch<CH, 1, 256>(data, index, out, B, C, N, K, st);
}

}  // namespace

extern "C" int usip_index_max_f32(const float* data, const int32_t* index, int32_t* max_idx,
                                  int B, int C, int N, int K, void* stream)
{
    if (B < 0 || C < 0 || N < 0 || K < 0) return USIP_EINVAL;
    if ((long long)B * C * K == 0) return USIP_OK;
    if (!max_idx || (N > 0 && (!data || !index))) return USIP_EINVAL;
    if ((long long)B * C > 0x7fffffffLL) return USIP_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (K > 8192) return USIP_EINVAL;                            // 64 KiB table
    // A workgroup owns CH channel rows of one cloud (the index row is read once for CH value rows) and keeps
    // U*(CH+1)*16 B per lane in flight.  Measured at B'=16, N=16384, K=512 (tools/index_max_sweep.py, ring of
    // inputs larger than the Infinity Cache): two rows per workgroup is the best trade at C=64 and C=128 --
    // more rows mean fewer, fatter workgroups that finish unevenly (C=64: 17 us at CH=2, 21 at 4, 35 at 8).
    const long long rows = (long long)B * C;
    int ch = (C % 2 == 0 && rows / 2 >= 512 && (long long)2 * K * 8 <= 65536) ? 2 : 1;
    int u = (N >= 4096) ? 2 : 1;
    int t = 256;
    const int tch = usip_tuning_value(USIP_TUNE_INDEX_MAX_CH), tu = usip_tuning_value(USIP_TUNE_INDEX_MAX_UNROLL);
    const int tt = usip_tuning_value(USIP_TUNE_INDEX_MAX_THREADS);
    if (tch > 0 && C % tch == 0 && (long long)tch * K * 8 <= 65536) ch = tch;
    if (tu > 0) u = tu;
    if (tt == 512 || tt == 1024) t = tt;
    switch (ch) {
    case 8: return launch_u<8>(u, t, data, index, max_idx, B, C, N, K, st);
    case 4: return launch_u<4>(u, t, data, index, max_idx, B, C, N, K, st);
    case 2: return launch_u<2>(u, t, data, index, max_idx, B, C, N, K, st);
    default: return launch_u<1>(u, t, data, index, max_idx, B, C, N, K, st);
    }
}
