// usip_amd/csrc/group.hip -- neighbourhood grouping and pooling on gfx950.
//
// Replaces the index-expansion + torch.gather + in-place decentering + torch.cat + torch.max chains
// of the reference (models/networks.py:699-710, models/layers.py:422-438, models/operations.py:
// 271-287): there the int64 index tensor is expanded to B x C x (M*K) (8 B per gathered float),
// gathered, decentered in a second pass, concatenated in a third, and max-pooled with a fourth
// pass that also materialises int64 arg-max indices.
//
//   group_gather      out[b, coff+c, m, k] = x[b, c, idx[b,m,k]] - (c < nsub ? sub[b,c,m] : 0)
//                     written straight into a channel slice of the (pre-concatenated) output.
//   group_gather_bwd  dx[b,c,n] += sum_{(m,k): idx = n} dout[b, coff+c, m, k]   (float atomics, as
//                     ATen's scatter_add; the only non-deterministic summation order on the path)
//   group_max         pooled[b,c,m] = max_k z[b,c,m,k], arg = first k attaining it
//   group_max_bwd     dz[b,c,m,k] = (k == arg[b,c,m]) ? dpooled[b,c,m] : 0
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void group_gather_kernel(
    const float* __restrict__ x, const int32_t* __restrict__ idx, const float* __restrict__ sub,
    float* __restrict__ out, int C, int N, int M, int K, int nsub, int Ctot, int coff)
{
    const int b = blockIdx.y;
    const int P = M * K;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const int n = idx[(long long)b * P + p];
    const int m = p / K;
    const float* xb = x + (long long)b * C * N;
    float* ob = out + ((long long)b * Ctot + coff) * P;
    for (int c = 0; c < C; ++c) {
        float v = xb[(long long)c * N + n];
        if (c < nsub) v -= sub[((long long)b * nsub + c) * M + m];
        ob[(long long)c * P + p] = v;
    }
}

__global__ __launch_bounds__(256) void group_gather_bwd_kernel(
    const float* __restrict__ dout, const int32_t* __restrict__ idx, float* __restrict__ dx,
    int C, int N, int P, int Ctot, int coff)
{
    const int b = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const int n = idx[(long long)b * P + p];
    const float* gb = dout + ((long long)b * Ctot + coff) * P;
    float* xb = dx + (long long)b * C * N;
    for (int c = 0; c < C; ++c) atomicAdd(&xb[(long long)c * N + n], gb[(long long)c * P + p]);
}

// L lanes (a power of two, <= 64) cooperate on one row of K values; rows = B*C*M.
template <int L>
__global__ __launch_bounds__(256) void group_max_kernel(
    const float* __restrict__ z, float* __restrict__ pooled, int32_t* __restrict__ arg, long long rows, int K)
{
    constexpr int RPB = 256 / L;                    // rows per block
    const int sub = threadIdx.x % L;
    const long long row = (long long)blockIdx.x * RPB + threadIdx.x / L;
    float best = -__builtin_inff();
    int bk = 0x7fffffff;
    if (row < rows) {
        const float* zr = z + row * K;
        for (int k = sub; k < K; k += L) {
            const float v = zr[k];
            if (bk == 0x7fffffff || v > best) { best = v; bk = k; }   // first k wins this lane's ties
        }
    }
#pragma unroll
    for (int off = L / 2; off > 0; off >>= 1) {
        const float ov = __shfl_xor(best, off);
        const int ok = __shfl_xor(bk, off);
        if (ov > best || (ov == best && ok < bk)) { best = ov; bk = ok; }
    }
    if (sub == 0 && row < rows) { pooled[row] = best; arg[row] = bk; }
}

// K % 4 == 0 and L = K/4 a power of two <= 64: every lane owns 4 consecutive neighbours (one 16-B load).
// coef != null: z is the producer's pre-BatchNorm output and the activation relu?(z*coef[0][c]+coef[1][c])
// is applied on the fly (the activated tensor is never written): rows are (b, c, m), c = (row / M) % C.
// Every thread group handles RPT rows whose 16-B loads are issued together (non-temporal: the tensor is next read
// in backward): 4 KiB in flight per wave is what moved ball_query from ~5 to 6.5 TB/s.
// (r03: rows handled in 32-bit arithmetic -- the channel of a row was a 64-bit division per row -- and the reduction over
// the L lanes of a row through DPP lane permutations (quad_perm / row_half_mirror / row_mirror: pure VALU) instead of
// three ds_bpermute per step.  Neither changed the time (70 us for 268 MB at K = 64, tools/group_max_bench.py): the pass
// is bound by its read rate, not by instructions; eight rows per thread group instead of four halved the workgroups
// and made the smaller pools slower (36 -> 50 us), so four it stays.)
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false); }

template <int L, int CTRL>
__device__ __forceinline__ void argmax_step(float& best, int& bk, float& braw)
{
    const float ov = dpp_f<CTRL>(best);
    const int ok = dpp_i<CTRL>(bk);
    const float orw = dpp_f<CTRL>(braw);
    if (ov > best || (ov == best && ok < bk)) { best = ov; bk = ok; braw = orw; }
}

// A workgroup walks `nbatch` consecutive batches (one batch = RPT rows per thread group = 4 KiB per wave) and requests
// batch j + 1 before it reduces batch j.  nbatch = 1 (one load latency per wave, 16384 workgroups per launch) is what the
// library uses: round 5 built the loop expecting ball_query's gain (its waves loop over a 64 KiB row and read 6.7 TB/s
// against 4.2-5.2 here) and measured it SLOWER inside the step, same box, alternating: 54.4-55.4 us per launch against
// 46.8-47.8 (profiles/r05b_forms_ab.txt) -- many short workgroups spread this pass's reads over the chip better than
// fewer long ones.  Kept behind knob r5_forms bit 3 as the measured form of that idea.  Same comparisons in the same
// order: same results either way.
template <int L>
__global__ __launch_bounds__(256) void group_max4_kernel(
    const float* __restrict__ z, float* __restrict__ pooled, int32_t* __restrict__ arg, long long rows,
    const float* __restrict__ coef, int relu, int C, int M, float* __restrict__ zarg, int nbatch)
{
    constexpr int RPB = 256 / L, RPT = 4;
    const int sub = threadIdx.x % L;
    const bool small = rows < (1LL << 31);
    const long long first = (long long)blockIdx.x * nbatch;             // this workgroup's first batch
    auto row_of = [&](long long batch, int j) { return batch * RPB * RPT + threadIdx.x / L + (long long)j * RPB; };
    float4 v[RPT], nv[RPT];
#pragma unroll
    for (int j = 0; j < RPT; ++j) v[j] = usip_load_stream4(z + (min(row_of(first, j), rows - 1) * L + sub) * 4);
    for (int bi = 0; bi < nbatch; ++bi) {
        const long long batch = first + bi;
        if (bi + 1 < nbatch) {                                           // (uniform) the next batch: clamped, branch-free
#pragma unroll
            for (int j = 0; j < RPT; ++j) nv[j] = usip_load_stream4(z + (min(row_of(batch + 1, j), rows - 1) * L + sub) * 4);
        }
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            const long long row = row_of(batch, j);
            float4 w = v[j];
            if (coef) {
                const int ch = small ? (int)(((unsigned)min(row, rows - 1) / (unsigned)M) % (unsigned)C)
                                     : (int)((min(row, rows - 1) / M) % C);
                const float s0 = coef[ch], s1 = coef[C + ch];
                w.x = __builtin_fmaf(w.x, s0, s1); w.y = __builtin_fmaf(w.y, s0, s1);
                w.z = __builtin_fmaf(w.z, s0, s1); w.w = __builtin_fmaf(w.w, s0, s1);
                if (relu) { w.x = fmaxf(w.x, 0.f); w.y = fmaxf(w.y, 0.f); w.z = fmaxf(w.z, 0.f); w.w = fmaxf(w.w, 0.f); }
            }
            float best = w.x, braw = v[j].x;              // braw: the INPUT value at the arg-max (the pre-BN output when
            int bk = sub * 4;                             // coef is given): the pooled layer's backward needs exactly it
            if (w.y > best) { best = w.y; bk = sub * 4 + 1; braw = v[j].y; }
            if (w.z > best) { best = w.z; bk = sub * 4 + 2; braw = v[j].z; }
            if (w.w > best) { best = w.w; bk = sub * 4 + 3; braw = v[j].w; }
            if constexpr (L == 4 || L == 16) {
                // every lane of the row ends up with the row's (max, first arg-max): the comparison is symmetric, so any
                // pairing of lanes that covers the row works -- xor 1, xor 2 inside a quad, then the two mirror steps
                argmax_step<L, 0xB1>(best, bk, braw);      // quad_perm [1,0,3,2]
                argmax_step<L, 0x4E>(best, bk, braw);      // quad_perm [2,3,0,1]
                if constexpr (L == 16) {
                    argmax_step<L, 0x141>(best, bk, braw); // row_half_mirror
                    argmax_step<L, 0x140>(best, bk, braw); // row_mirror
                }
            } else {
#pragma unroll
                for (int off = L / 2; off > 0; off >>= 1) {
                    const float ov = __shfl_xor(best, off);
                    const int ok = __shfl_xor(bk, off);
                    const float orw = __shfl_xor(braw, off);
                    if (ov > best || (ov == best && ok < bk)) { best = ov; bk = ok; braw = orw; }
                }
            }
            if (sub == 0 && row < rows) {
                pooled[row] = best;
                arg[row] = bk;
                if (zarg) zarg[row] = braw;
            }
        }
#pragma unroll
        for (int j = 0; j < RPT; ++j) v[j] = nv[j];
    }
}

// batches per workgroup for group_max4_kernel: 1; with the knob as many as leave >= 8 workgroups per CU (2048), at most 8
static int group_max4_batches(long long batches)
{
    if (!(usip_tuning_value(USIP_TUNE_R5_FORMS) & 8)) return 1;
    int nb = 1;
    while (nb < 8 && batches / (nb * 2) >= 2048) nb *= 2;
    return nb;
}

__global__ __launch_bounds__(256) void group_max_bwd4_kernel(
    const float* __restrict__ dpooled, const int32_t* __restrict__ arg, float* __restrict__ dz,
    long long total4, int K4)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const long long row = i / K4;
    const int k = (int)(i - row * K4) * 4;
    const int a = arg[row] - k;
    const float g = dpooled[row];
    *reinterpret_cast<float4*>(dz + i * 4) =
        make_float4(a == 0 ? g : 0.f, a == 1 ? g : 0.f, a == 2 ? g : 0.f, a == 3 ? g : 0.f);
}

// dz[row][arg[row]] += dpooled[row]: the pooling gradient folded into a dense gradient that already exists
// (the pooled tensor's source also fed a layer directly), touching rows elements instead of writing a dense
// rows x K tensor and adding two dense tensors.
__global__ __launch_bounds__(256) void group_max_bwd_add_kernel(
    const float* __restrict__ dpooled, const int32_t* __restrict__ arg, float* __restrict__ dz, long long rows, int K)
{
    const long long row = (long long)blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    dz[row * K + arg[row]] += dpooled[row];
}

// Scatter-add through LDS, reproducible.  A workgroup owns CPB channels of one cloud; each of its four waves
// accumulates ITS quarter of the positions into ITS OWN table [CPB][N] with ds_add_f32, and the four tables are
// combined in a fixed order at the end.  No two waves ever add into the same cell, so the order of additions
// into a cell is program order, and lanes of one instruction that hit the same cell are serialised by the LDS
// unit in its fixed lane order: the same bits on every run (tests/test_group_gpu.py launches it repeatedly) --
// the first version let four waves share one table, and their interleaving depended on timing.  Measured
// bound: the LDS float-add rate (~0.35 lane-adds per clock per CU with random cells; 16.8 M adds = 89 us for the
// KNN-feature gradient) -- neither prefetching the next batch of positions nor 1/2/4 channels per workgroup
// nor one wave per workgroup moved it.  4*N*CPB*4 B <= 64 KiB.
template <int CPB, bool VEC>
__global__ __launch_bounds__(256) void group_gather_bwd_lds_kernel(
    const float* __restrict__ dout, const int32_t* __restrict__ idx, float* __restrict__ dx,
    int C, int N, int P, int Ctot, int coff)
{
    extern __shared__ __attribute__((aligned(16))) float tables[];      // [4 waves][CPB][N]
    const int b = blockIdx.y, c0 = blockIdx.x * CPB, lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 4 * CPB * N; i += 256) tables[i] = 0.f;
    __syncthreads();
    float* table = tables + wave * CPB * N;
    const int32_t* ib = idx + (long long)b * P;
    const float* gb = dout + ((long long)b * Ctot + coff + c0) * P;
    // the wave's quarter of the positions, a multiple of 4 long (except the tail)
    const int per = ((P + 15) / 16) * 4;
    const int pbeg = min(P, wave * per), pend = min(P, pbeg + per);
    if (VEC) {
        auto fetch = [&](int p, int4& n, float4 (&g)[CPB]) {           // clamped address; masked by the loop bound
            const int pc = min(p, P - 4);
            n = *reinterpret_cast<const int4*>(ib + pc);
#pragma unroll
            for (int c = 0; c < CPB; ++c)
                g[c] = *reinterpret_cast<const float4*>(gb + (long long)min(c, C - 1 - c0) * P + pc);
        };
        int4 n, nn;
        float4 g[CPB], gn[CPB];
        fetch(pbeg + lane * 4, n, g);
        for (int p = pbeg + lane * 4; p < pend; p += 256) {
            fetch(p + 256, nn, gn);
#pragma unroll
            for (int c = 0; c < CPB; ++c) {                              // position order p, p+1, p+2, p+3
                if (c0 + c >= C) break;
                atomicAdd(&table[c * N + n.x], g[c].x);
                atomicAdd(&table[c * N + n.y], g[c].y);
                atomicAdd(&table[c * N + n.z], g[c].z);
                atomicAdd(&table[c * N + n.w], g[c].w);
            }
            n = nn;
#pragma unroll
            for (int c = 0; c < CPB; ++c) g[c] = gn[c];
        }
    } else {
        for (int p = pbeg + lane; p < pend; p += 64) {
            const int n = ib[p];
#pragma unroll
            for (int c = 0; c < CPB; ++c)
                if (c0 + c < C) atomicAdd(&table[c * N + n], gb[(long long)c * P + p]);
        }
    }
    __syncthreads();
    float* xb = dx + ((long long)b * C + c0) * N;
    const int sz = CPB * N;
    for (int i = threadIdx.x; i < sz; i += 256)
        if (c0 + i / N < C) xb[i] = (tables[i] + tables[sz + i]) + (tables[2 * sz + i] + tables[3 * sz + i]);
}

__global__ __launch_bounds__(256) void group_max_bwd_kernel(
    const float* __restrict__ dpooled, const int32_t* __restrict__ arg, float* __restrict__ dz,
    long long total, int K)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long long row = i / K;
    const int k = (int)(i - row * K);
    dz[i] = (arg[row] == k) ? dpooled[row] : 0.0f;
}

}  // namespace

extern "C" int usip_group_gather_f32(const float* x, const int32_t* idx, const float* sub, float* out,
                                     int B, int C, int N, int M, int K, int nsub, int Ctot, int coff,
                                     void* stream)
{
    if (B < 0 || C < 1 || N < 1 || M < 0 || K < 0 || nsub < 0 || nsub > C || coff < 0 || coff + C > Ctot)
        return USIP_EINVAL;
    if ((long long)B * M * K == 0) return USIP_OK;
    if (!x || !idx || !out || (nsub > 0 && !sub) || B > 65535) return USIP_EINVAL;
    USIP_LAUNCH(group_gather_kernel, dim3(usip_ceil_div((long long)M * K, 256), B), dim3(256), 0,
                (hipStream_t)stream, x, idx, sub, out, C, N, M, K, nsub, Ctot, coff);
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}

extern "C" int usip_group_gather_backward_f32(const float* dout, const int32_t* idx, float* dx,
                                              int B, int C, int N, int M, int K, int Ctot, int coff,
                                              void* stream)
{
    if (B < 0 || C < 1 || N < 1 || M < 0 || K < 0 || coff < 0 || coff + C > Ctot) return USIP_EINVAL;
    if (!dx) return USIP_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if ((long long)B * M * K > 0 && dout && idx && B <= 65535 && (long long)N * 4 * 2 * 4 <= 65536) {
        // LDS path: 2 channels per workgroup, every dx element written exactly once (no memset)
        const long long P = (long long)M * K;
        const bool vec = (P % 4 == 0) && ((reinterpret_cast<uintptr_t>(dout) & 15u) == 0) &&
                         ((reinterpret_cast<uintptr_t>(idx) & 15u) == 0);
        const dim3 grid(usip_ceil_div(C, 2), B), block(256);
        const size_t lds = (size_t)4 * 2 * N * sizeof(float);
        if (vec)
            USIP_LAUNCH((group_gather_bwd_lds_kernel<2, true>), grid, block, lds, st, dout, idx, dx, C, N, M * K, Ctot, coff);
        else
            USIP_LAUNCH((group_gather_bwd_lds_kernel<2, false>), grid, block, lds, st, dout, idx, dx, C, N, M * K, Ctot, coff);
        USIP_LAUNCH_CHECK();
        return USIP_OK;
    }
    (void)hipGetLastError();
    hipError_t e = hipMemsetAsync(dx, 0, sizeof(float) * (size_t)B * C * N, st);
    if (e != hipSuccess) return (int)e;
    if ((long long)B * M * K == 0) return USIP_OK;
    if (!dout || !idx || B > 65535) return USIP_EINVAL;
    USIP_LAUNCH(group_gather_bwd_kernel, dim3(usip_ceil_div((long long)M * K, 256), B), dim3(256), 0, st,
                dout, idx, dx, C, N, M * K, Ctot, coff);
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}

extern "C" int usip_group_max_act_f32(const float* y, const float* coef, int relu, float* pooled, int32_t* arg,
                                      float* yarg, int B, int C, int M, int K, void* stream)
{
    const long long rows = (long long)B * C * M;
    const int L4 = K / 4;
    if (B < 0 || C < 1 || M < 0 || K < 4 || K % 4 != 0 || (L4 & (L4 - 1)) != 0 || L4 > 64) return USIP_EINVAL;
    if (rows == 0) return USIP_OK;
    if (!y || !coef || !pooled || !arg || (reinterpret_cast<uintptr_t>(y) & 15u)) return USIP_EINVAL;
    hipStream_t st = (hipStream_t)stream;
#define USIP_GM4(L_)                                                                             \
    if (L4 == L_) {                                                                              \
        const long long batches = (rows + 4 * (256 / L_) - 1) / (4 * (256 / L_));                \
        const int nbatch = group_max4_batches(batches);                                          \
        const long long blocks = (batches + nbatch - 1) / nbatch;                                \
        if (blocks > 0x7fffffffLL) return USIP_EINVAL;                                           \
        USIP_LAUNCH((group_max4_kernel<L_>), dim3((unsigned)blocks), dim3(256), 0, st, y, pooled, arg, rows, \
                    coef, relu, C, M, yarg, nbatch);                                             \
    }
    USIP_GM4(1) USIP_GM4(2) USIP_GM4(4) USIP_GM4(8) USIP_GM4(16) USIP_GM4(32) USIP_GM4(64)
#undef USIP_GM4
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}

extern "C" int usip_group_max_f32(const float* z, float* pooled, int32_t* arg, long long rows, int K,
                                  void* stream)
{
    if (rows < 0 || K < 1) return USIP_EINVAL;
    if (rows == 0) return USIP_OK;
    if (!z || !pooled || !arg) return USIP_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int L4 = K / 4;
    if (K % 4 == 0 && (L4 & (L4 - 1)) == 0 && L4 <= 64 && (reinterpret_cast<uintptr_t>(z) & 15u) == 0) {
#define USIP_GM4(L_)                                                                             \
        if (L4 == L_) {                                                                          \
            const long long batches = (rows + 4 * (256 / L_) - 1) / (4 * (256 / L_));            \
            const int nbatch = group_max4_batches(batches);                                      \
            const long long blocks = (batches + nbatch - 1) / nbatch;                            \
            if (blocks > 0x7fffffffLL) return USIP_EINVAL;                                       \
            USIP_LAUNCH((group_max4_kernel<L_>), dim3((unsigned)blocks), dim3(256), 0, st, z, pooled, arg, rows, \
                        (const float*)nullptr, 0, 1, 1, (float*)nullptr, nbatch);                \
        }
        USIP_GM4(1) USIP_GM4(2) USIP_GM4(4) USIP_GM4(8) USIP_GM4(16) USIP_GM4(32) USIP_GM4(64)
#undef USIP_GM4
        USIP_LAUNCH_CHECK();
        return USIP_OK;
    }
    int L = 1;
    while (L < K && L < 64) L <<= 1;
#define USIP_GM(L_)                                                                              \
    if (L == L_) {                                                                               \
        const long long blocks = (rows + (256 / L_) - 1) / (256 / L_);                           \
        if (blocks > 0x7fffffffLL) return USIP_EINVAL;                                           \
        USIP_LAUNCH((group_max_kernel<L_>), dim3((unsigned)blocks), dim3(256), 0, st, z, pooled, arg, rows, K); \
    }
    USIP_GM(1) USIP_GM(2) USIP_GM(4) USIP_GM(8) USIP_GM(16) USIP_GM(32) USIP_GM(64)
#undef USIP_GM
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}

extern "C" int usip_group_max_backward_f32(const float* dpooled, const int32_t* arg, float* dz,
                                           long long rows, int K, void* stream)
{
    if (rows < 0 || K < 1) return USIP_EINVAL;
    if (rows == 0) return USIP_OK;
    if (!dpooled || !arg || !dz) return USIP_EINVAL;
    const long long total = rows * K;
    if (K % 4 == 0 && (reinterpret_cast<uintptr_t>(dz) & 15u) == 0) {
        const long long total4 = total / 4, blocks4 = (total4 + 255) / 256;
        if (blocks4 > 0x7fffffffLL) return USIP_EINVAL;
        USIP_LAUNCH(group_max_bwd4_kernel, dim3((unsigned)blocks4), dim3(256), 0, (hipStream_t)stream,
                    dpooled, arg, dz, total4, K / 4);
        USIP_LAUNCH_CHECK();
        return USIP_OK;
    }
    const long long blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffLL) return USIP_EINVAL;
    USIP_LAUNCH(group_max_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                dpooled, arg, dz, total, K);
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}

extern "C" int usip_group_max_backward_add_f32(const float* dpooled, const int32_t* arg, float* dz,
                                               long long rows, int K, void* stream)
{
    if (rows < 0 || K < 1) return USIP_EINVAL;
    if (rows == 0) return USIP_OK;
    if (!dpooled || !arg || !dz) return USIP_EINVAL;
    const long long blocks = (rows + 255) / 256;
    if (blocks > 0x7fffffffLL) return USIP_EINVAL;
    USIP_LAUNCH(group_max_bwd_add_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                dpooled, arg, dz, rows, K);
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}
