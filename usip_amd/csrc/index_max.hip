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

__device__ __forceinline__ float from_ordered_bits(unsigned o)
{
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
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
    int C, int N, int K, float* __restrict__ vals, const int32_t* __restrict__ count, int Ctot)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long table[];   // [CH][K]
    const int cgroups = C / CH;
    const int b = blockIdx.x / cgroups;
    const int c0 = (blockIdx.x % cgroups) * CH;
    const unsigned long long init = ((unsigned long long)ordered_bits(FLOOR) << 32) | 0xffffffffull;
    for (int i = threadIdx.x; i < CH * K; i += T) table[i] = init;
    __syncthreads();

    const int32_t* idx = index + (long long)b * N;
    const float* rows = data + ((long long)b * Ctot + c0) * N;
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
    float* ov = vals ? vals + ((long long)b * C + c0) * K : nullptr;
    for (int i = threadIdx.x; i < CH * K; i += T) {
        unsigned long long key = table[i];
        o[i] = (key == init) ? 0 : (int32_t)(~(unsigned)key);
        if (ov) {
            // data[b, c, max_idx] * (count > 0) -- what the reference forms with torch.gather and mask_row_max
            // (networks.py:117-118): the winning key holds the value; a node whose members are all <= -1000 reports
            // index 0 and therefore the value at n = 0
            const int c = i / K, k = i - c * K;
            const bool has = count ? count[(long long)b * K + k] > 0 : true;
            float v = 0.f;
            if (has) v = (key == init) ? (N > 0 ? rows[(long long)c * N] : 0.f) : from_ordered_bits((unsigned)(key >> 32));
            ov[i] = v;
        }
    }
}

// ddata[b, coff + c, max_idx[b,c,k]] += g[b,c,k] for the nodes with members: the backward of the gather + mask above,
// added INTO a dense gradient that already exists (B*C*K touched elements instead of a zero fill, a scatter and a dense
// add).  Two nodes share a target only in the all-below-floor case (both report n = 0), hence the atomic.
__global__ __launch_bounds__(256) void index_max_values_bwd_add_kernel(
    const float* __restrict__ g, const int32_t* __restrict__ max_idx, const int32_t* __restrict__ count,
    float* __restrict__ ddata, long long total, int C, int Ctot, int coff, int N, int K)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int k = (int)(i % K);
    const long long bc = i / K;
    const int c = (int)(bc % C);
    const long long b = bc / C;
    if (count && count[b * K + k] <= 0) return;
    atomicAdd(&ddata[(b * Ctot + coff + c) * N + max_idx[i]], g[i]);
}

// The same backward as ONE dense pass, for callers that need the gradient as a contiguous tensor anyway:
//     dz[b,c,n] = (src ? src[b, soff + c, n] : 0) + (n is the reported arg-max of its node in channel c ? g[b,c,node] : 0)
// A workgroup owns one (b, c) row: the node tables (g, max_idx; -1 for unpopulated nodes) sit in LDS, the assignment
// row and src stream through with 16-B accesses, every output is written once -- no zero fill, no scattered
// read-modify-write (1 M random 4-B updates cost 40 us; this pass runs at the streaming rate).  Position 0 is special:
// it also receives the nodes whose members are all below the floor (they report index 0), whichever node it is in.
template <bool VEC>
__global__ __launch_bounds__(256) void index_max_values_bwd_dense_kernel(
    const float* __restrict__ g, const int32_t* __restrict__ max_idx, const int32_t* __restrict__ count,
    const int32_t* __restrict__ index, const float* __restrict__ src, float* __restrict__ dz,
    int C, int Csrc, int soff, int N, int K)
{
    extern __shared__ __attribute__((aligned(16))) float gs[];            // [K] gradients, then [K] arg-max (int)
    int* is = reinterpret_cast<int*>(gs + K);
    const int b = blockIdx.x / C, c = blockIdx.x - b * C;
    const long long row = (long long)b * C + c;
    for (int k = threadIdx.x; k < K; k += 256) {
        const bool has = count ? count[(long long)b * K + k] > 0 : true;
        gs[k] = g[row * K + k];
        is[k] = has ? max_idx[row * K + k] : -1;
    }
    __syncthreads();
    const int32_t* ib = index + (long long)b * N;
    const float* sr = src ? src + ((long long)b * Csrc + soff + c) * N : nullptr;
    float* o = dz + row * N;
    auto term = [&](int k, int n) { return ((unsigned)k < (unsigned)K && is[k] == n && n != 0) ? gs[k] : 0.f; };
    if (VEC) {
        for (int n = threadIdx.x * 4; n < N; n += 1024) {
            const int4 k4 = *reinterpret_cast<const int4*>(ib + n);
            float4 v = sr ? usip_load_stream4(sr + n) : make_float4(0.f, 0.f, 0.f, 0.f);
            v.x += term(k4.x, n); v.y += term(k4.y, n + 1); v.z += term(k4.z, n + 2); v.w += term(k4.w, n + 3);
            *reinterpret_cast<float4*>(o + n) = v;
        }
    } else {
        for (int n = threadIdx.x; n < N; n += 256) o[n] = (sr ? sr[n] : 0.f) + term(ib[n], n);
    }
    __syncthreads();                                                      // o[0] was written (without its term) above
    if (threadIdx.x < 64 && N > 0) {
        float a = 0.f;
        for (int k = threadIdx.x; k < K; k += 64) a += (is[k] == 0) ? gs[k] : 0.f;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off);
        if (threadIdx.x == 0) o[0] = (sr ? sr[0] : 0.f) + a;
    }
}

struct Extra { float* vals; const int32_t* count; int Ctot; };

template <int CH, int U, int T>
int launch(const float* data, const int32_t* index, int32_t* out, int B, int C, int N, int K, hipStream_t st, Extra x)
{
    const bool vec = (N % 4 == 0) && ((reinterpret_cast<uintptr_t>(data) & 15u) == 0) &&
                     ((reinterpret_cast<uintptr_t>(index) & 15u) == 0);
    float* vals = x.vals;
    const int32_t* count = x.count;
    const int Ctot = x.Ctot;
    const size_t lds = (size_t)CH * K * sizeof(unsigned long long);
    dim3 grid((unsigned)(B * (C / CH))), block(T);
    if (vec)
        USIP_LAUNCH((index_max_kernel<CH, U, true, T>), grid, block, lds, st, data, index, out, C, N, K, vals, count, Ctot);
    else
        USIP_LAUNCH((index_max_kernel<CH, 1, false, 256>), grid, dim3(256), lds, st, data, index, out, C, N, K, vals, count, Ctot);
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}

template <int CH>
int launch_u(int U, int T, const float* data, const int32_t* index, int32_t* out, int B, int C, int N, int K, hipStream_t st, Extra x)
{
    if (T >= 1024) return U >= 2 ? launch<CH, 2, 1024>(data, index, out, B, C, N, K, st, x)
                                 : launch<CH, 1, 1024>(data, index, out, B, C, N, K, st, x);
    if (T >= 512) return U >= 2 ? launch<CH, 2, 512>(data, index, out, B, C, N, K, st, x)
                                : launch<CH, 1, 512>(data, index, out, B, C, N, K, st, x);
    if (U >= 4 && CH <= 4) return launch<CH, 4, 256>(data, index, out, B, C, N, K, st, x);
    if (U >= 2) return launch<CH, 2, 256>(data, index, out, B, C, N, K, st, x);
    return launch<CH, 1, 256>(data, index, out, B, C, N, K, st, x);
}

}  // namespace

static int index_max_impl(const float* data, const int32_t* index, int32_t* max_idx, int B, int C, int N, int K,
                          void* stream, Extra x)
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
    case 8: return launch_u<8>(u, t, data, index, max_idx, B, C, N, K, st, x);
    case 4: return launch_u<4>(u, t, data, index, max_idx, B, C, N, K, st, x);
    case 2: return launch_u<2>(u, t, data, index, max_idx, B, C, N, K, st, x);
    default: return launch_u<1>(u, t, data, index, max_idx, B, C, N, K, st, x);
    }
}

extern "C" int usip_index_max_f32(const float* data, const int32_t* index, int32_t* max_idx,
                                  int B, int C, int N, int K, void* stream)
{
    return index_max_impl(data, index, max_idx, B, C, N, K, stream, Extra{nullptr, nullptr, C});
}

extern "C" int usip_index_max_values_f32(const float* data, const int32_t* index, const int32_t* count,
                                         int32_t* max_idx, float* max_val, int B, int C, int Ctot, int N, int K,
                                         void* stream)
{
    if (Ctot < C || ((long long)B * C * K > 0 && !max_val)) return USIP_EINVAL;
    if ((long long)B * Ctot > 0x7fffffffLL) return USIP_EINVAL;
    return index_max_impl(data, index, max_idx, B, C, N, K, stream, Extra{max_val, count, Ctot});
}

extern "C" int usip_index_max_values_backward_add_f32(const float* g, const int32_t* max_idx, const int32_t* count,
                                                      float* ddata, int B, int C, int Ctot, int coff, int N, int K,
                                                      void* stream)
{
    if (B < 0 || C < 0 || N < 0 || K < 0 || coff < 0 || coff + C > Ctot) return USIP_EINVAL;
    const long long total = (long long)B * C * K;
    if (total == 0 || N == 0) return USIP_OK;
    if (!g || !max_idx || !ddata) return USIP_EINVAL;
    const long long blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffLL) return USIP_EINVAL;
    USIP_LAUNCH(index_max_values_bwd_add_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                g, max_idx, count, ddata, total, C, Ctot, coff, N, K);
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}

extern "C" int usip_index_max_values_backward_f32(const float* g, const int32_t* max_idx, const int32_t* count,
                                                  const int32_t* index, const float* src, int Csrc, int soff,
                                                  float* dz, int B, int C, int N, int K, void* stream)
{
    if (B < 0 || C < 0 || N < 0 || K < 0) return USIP_EINVAL;
    if ((long long)B * C * N == 0) return USIP_OK;
    if (!g || !max_idx || !index || !dz || (src && (soff < 0 || soff + C > Csrc))) return USIP_EINVAL;
    if ((long long)B * C > 0x7fffffffLL || (size_t)K * 8 > 65536) return USIP_EINVAL;
    const bool vec = (N % 4 == 0) && ((reinterpret_cast<uintptr_t>(dz) & 15u) == 0) &&
                     ((reinterpret_cast<uintptr_t>(index) & 15u) == 0) && (!src || (reinterpret_cast<uintptr_t>(src) & 15u) == 0);
    dim3 grid((unsigned)(B * C)), block(256);
    const size_t lds = (size_t)K * 8;
    if (vec)
        USIP_LAUNCH((index_max_values_bwd_dense_kernel<true>), grid, block, lds, (hipStream_t)stream,
                    g, max_idx, count, index, src, dz, C, Csrc, soff, N, K);
    else
        USIP_LAUNCH((index_max_values_bwd_dense_kernel<false>), grid, block, lds, (hipStream_t)stream,
                    g, max_idx, count, index, src, dz, C, Csrc, soff, N, K);
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}
