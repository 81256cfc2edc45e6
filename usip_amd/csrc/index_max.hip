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

__device__ __forceinline__ void fold(unsigned long long* table, int k, float v, int n)
{
    if (v > FLOOR) {                                             // false for NaN as well
        v += 0.0f;                                               // -0.0 -> +0.0: they compare equal
        unsigned long long key = ((unsigned long long)ordered_bits(v) << 32) | (unsigned)(~n);
        atomicMax(&table[k], key);                               // LDS ds_max_u64, no return
    }
}

// U = prefetch depth: the loads of U consecutive 1024-point steps (U index vectors + U*CH value vectors per lane)
// are all issued before the first LDS atomic, so a workgroup keeps U*(CH+1)*4 KiB in flight instead of (CH+1)*4.
// (An atomic in the loop body stops the compiler from hoisting the next step's loads on its own.)
template <int CH, int U, bool VEC>
__global__ __launch_bounds__(256) void index_max_kernel(
    const float* __restrict__ data, const int32_t* __restrict__ index, int32_t* __restrict__ out,
    int C, int N, int K)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long table[];   // [CH][K]
    const int cgroups = C / CH;
    const int b = blockIdx.x / cgroups;
    const int c0 = (blockIdx.x % cgroups) * CH;
    const unsigned long long init = ((unsigned long long)ordered_bits(FLOOR) << 32) | 0xffffffffull;
    for (int i = threadIdx.x; i < CH * K; i += 256) table[i] = init;
    __syncthreads();

    const int32_t* idx = index + (long long)b * N;
    const float* rows = data + ((long long)b * C + c0) * N;
    if (VEC) {
        for (int n0 = threadIdx.x * 4; n0 < N; n0 += 1024 * U) {
            int4 k4[U];
            float4 v[U][CH];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                // past the end: re-read the lane's first vector (always valid) and drop it below
                const int n = (n0 + u * 1024 < N) ? n0 + u * 1024 : n0;
                k4[u] = *reinterpret_cast<const int4*>(idx + n);
#pragma unroll
                for (int c = 0; c < CH; ++c)
                    v[u][c] = usip_load_stream4(rows + (long long)c * N + n);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int n = n0 + u * 1024;
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
        for (int n = threadIdx.x; n < N; n += 256) {
            int k = idx[n];
#pragma unroll
            for (int c = 0; c < CH; ++c) fold(table + c * K, k, rows[(long long)c * N + n], n);
        }
    }
    __syncthreads();
    int32_t* o = out + ((long long)b * C + c0) * K;
    for (int i = threadIdx.x; i < CH * K; i += 256) {
        unsigned long long key = table[i];
        o[i] = (key == init) ? 0 : (int32_t)(~(unsigned)key);
    }
}

template <int CH, int U>
int launch(const float* data, const int32_t* index, int32_t* out, int B, int C, int N, int K, hipStream_t st)
{
    const bool vec = (N % 4 == 0) && ((reinterpret_cast<uintptr_t>(data) & 15u) == 0) &&
                     ((reinterpret_cast<uintptr_t>(index) & 15u) == 0);
    const size_t lds = (size_t)CH * K * sizeof(unsigned long long);
    dim3 grid((unsigned)(B * (C / CH))), block(256);
    if (vec)
        USIP_LAUNCH((index_max_kernel<CH, U, true>), grid, block, lds, st, data, index, out, C, N, K);
    else
        USIP_LAUNCH((index_max_kernel<CH, 1, false>), grid, block, lds, st, data, index, out, C, N, K);
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}

template <int CH>
int launch_u(int U, const float* data, const int32_t* index, int32_t* out, int B, int C, int N, int K, hipStream_t st)
{
    if (U >= 4 && CH <= 4) return launch<CH, 4>(data, index, out, B, C, N, K, st);
    if (U >= 2) return launch<CH, 2>(data, index, out, B, C, N, K, st);
    return launch<CH, 1>(data, index, out, B, C, N, K, st);
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
    // A workgroup owns CH channel rows of one cloud: the index row is read once for CH value rows, and with the
    // prefetch depth U it keeps U*(CH+1)*4 KiB of loads in flight.  Largest CH that still leaves two workgroups
    // per CU (fewer, fatter workgroups finish unevenly) and fits the 64 KiB LDS table.
    const long long rows = (long long)B * C;
    int ch = 1;
    for (int cand = 8; cand > 1; cand >>= 1)
        if (C % cand == 0 && rows / cand >= 512 && (long long)cand * K * 8 <= 65536) { ch = cand; break; }
    int u = (N >= 4096) ? ((ch <= 4) ? 4 : 2) : 1;
    const int tch = usip_tuning_value(USIP_TUNE_INDEX_MAX_CH), tu = usip_tuning_value(USIP_TUNE_INDEX_MAX_UNROLL);
    if (tch > 0 && C % tch == 0 && (long long)tch * K * 8 <= 65536) ch = tch;
    if (tu > 0) u = tu;
    switch (ch) {
    case 8: return launch_u<8>(u, data, index, max_idx, B, C, N, K, st);
    case 4: return launch_u<4>(u, data, index, max_idx, B, C, N, K, st);
    case 2: return launch_u<2>(u, data, index, max_idx, B, C, N, K, st);
    default: return launch_u<1>(u, data, index, max_idx, B, C, N, K, st);
    }
}
