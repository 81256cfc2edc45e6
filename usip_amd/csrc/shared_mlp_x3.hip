// usip_amd/csrc/shared_mlp_x3.hip -- the matrix-bound shared-MLP GEMMs (forward and data gradient of the 128..640-
// wide layers: models/layers.py:208-216, :293-303, :401-440) as fp32-ACCURATE products on the bf16 matrix cores.
//
// Arithmetic ("f32x3", see shared_mlp_bf16.hip for the derivation): every fp32 operand is split exactly into three
// bf16 planes, the six plane pairs of weight >= 2^-18 are accumulated in fp32 by v_mfma_f32_32x32x16_bf16.
//
// What this file adds over the NS = 3 instantiation of gemm_bf16_kernel is the instruction diet the counters asked
// for.  That kernel spends 230 VALU + 150 SALU instructions per 24 MFMAs (SQ_INSTS_*; MFMA pipe 35 % busy): it is
// issue-bound on the operand preparation, not on memory and not on the matrix pipe.  Here
//   * the WEIGHT operand is split ONCE per step by split3_tiles_kernel into the exact LDS image of every
//     (128-row, 16-k) stage -- three 4 KiB planes, contiguous -- so a stage of A is three coalesced 16-B loads and
//     three ds_write_b128 per thread, no arithmetic (was: 8 dword loads, 8 three-way splits, 8 selects);
//   * the streamed operand is converted in PAIRS (v_cvt_pk_bf16_f32 takes two floats), unpacked with one shift and
//     one mask per pair, and is never masked per lane: rows of k beyond K meet zero weights (the planes are zero
//     padded), positions beyond P are never stored or counted, so clamped loads suffice;
//   * prologue coefficients come from LDS (loaded once per workgroup) instead of 16 scalar loads per stage whose
//     s_waitcnt lgkmcnt(0) also drained the LDS queue;
//   * LDS rows are 32 B (16 bf16) with the two 16-B halves swapped in every other block of 8 rows: conflict-free
//     ds_read_b128 without padding, 48 KiB for both stages of both operands (two workgroups per CU).
#include "mlp_common.h"
#include "split_common.h"
#include <type_traits>

using namespace usip_mlp;

namespace {

// byte offset of (row, 16-B half) inside a [rows][16 bf16] plane: halves swapped in every other block of 8 rows
__device__ __forceinline__ int lds_off(int row, int half) { return row * 32 + ((half ^ (row >> 3)) & 1) * 16; }

// Scale of a weight operand for the two-plane fp16 image: 2^e with max|A| 2^e in [2^13, 2^14).  Two levels without
// atomics: X2H_PARTS workgroups per operand each leave the maximum of their share behind the image (floats
// [4, 4 + X2H_PARTS) of the trailer); the split kernel combines them (every workgroup the same way) and the operand's
// first workgroup writes the scale into trailer float 0 for the GEMM.
constexpr int X2H_PARTS = 32;

__device__ __forceinline__ float* x2h_trailer(const usip_split3_desc& d)
{
    const int bm = d.tile_rows, ksteps = (d.K + XBK - 1) / XBK;
    return reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(d.planes) +
                                    (long long)((d.M + bm - 1) / bm) * ksteps * 2 * bm * 32);
}

__device__ __forceinline__ void absmax_part(const usip_split3_desc& d, int part)
{
    __shared__ float red[4];
    const long long total = (long long)d.M * d.K, per = (total + X2H_PARTS - 1) / X2H_PARTS;
    const long long i0 = part * per, i1 = i0 + per < total ? i0 + per : total;
    float mx = 0.f;
    for (long long i = i0 + threadIdx.x; i < i1; i += 256) {
        const int k = (int)(i / d.M), m = (int)(i % d.M);
        mx = fmaxf(mx, fabsf(d.At[(long long)k * d.lda + m]));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_down(mx, off));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) x2h_trailer(d)[4 + part] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

__device__ __forceinline__ float x2h_weight_scale(const usip_split3_desc& d)
{
    const float* t = x2h_trailer(d);
    float mx = 0.f;
    for (int i = 0; i < X2H_PARTS; ++i) mx = fmaxf(mx, t[4 + i]);
    return pow2_scale(mx, X2H_TOP - 1);
}

__global__ __launch_bounds__(256) void absmax_multi_kernel(const usip_split3_desc* __restrict__ descs, int n)
{
    const usip_split3_desc d = descs[blockIdx.x / X2H_PARTS];
    if (d.reserved != 2) return;
    absmax_part(d, blockIdx.x % X2H_PARTS);
    (void)n;
}

__global__ __launch_bounds__(256) void absmax_one_kernel(usip_split3_desc d) { absmax_part(d, blockIdx.x); }

// ------------------------------------------------------------------------------------------------
// Weight operand -> per-stage LDS images.  At is the K-major operand of usip_mlp_gemm_f32 (A[m][k] = At[k*lda + m]);
// out[((mt * ksteps + ks) * 3 + plane) * bm*2 + lds_off(row, half) / 16] (16-B chunks) with the chunk holding
// k = ks*16 + half*8 .. +7 of row mt*bm + row, zero outside M x K.  bm (128 or 256) = rows of the GEMM's tile.
// One workgroup of 2*bm threads per (mt, ks) stage.
__global__ __launch_bounds__(512) void split3_tiles_kernel(const float* __restrict__ At, int lda, int M, int K,
                                                           uint4* __restrict__ out, int ksteps, int bm, int npl)
{
    float wscale = 1.0f;
    if (npl == 2) {
        const usip_split3_desc d{At, out, lda, M, K, 0, bm, 2};
        wscale = x2h_weight_scale(d);
        if (blockIdx.x == 0 && threadIdx.x == 0) x2h_trailer(d)[0] = wscale;
    }
    const int ks = blockIdx.x % ksteps, mt = blockIdx.x / ksteps;
    const int row = threadIdx.x % bm, half = threadIdx.x / bm;
    const int m = mt * bm + row;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int k = ks * XBK + half * 8 + i;
        v[i] = (m < M && k < K) ? At[(long long)k * lda + m] : 0.0f;
    }
    unsigned p[3][4];
    if (npl == 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) split_pair_h(v[2 * j] * wscale, v[2 * j + 1] * wscale, p[0][j], p[1][j]);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) split_pair(v[2 * j], v[2 * j + 1], p[0][j], p[1][j], p[2][j]);
    }
    uint4* stage = out + (long long)blockIdx.x * (npl * bm * 2);
    for (int s = 0; s < npl; ++s)        // chunk position = its LDS position (lds_off / 16): the GEMM copies linearly
        stage[s * (bm * 2) + row * 2 + ((half ^ (row >> 3)) & 1)] = make_uint4(p[s][0], p[s][1], p[s][2], p[s][3]);
}

// The same for ALL weight operands of a step in one launch (a dozen 5-us launches otherwise): descs[i] names operand i
// and the first workgroup of its range.
__global__ __launch_bounds__(512) void split3_multi_kernel(const usip_split3_desc* __restrict__ descs, int n)
{
    int i = 0;
    while (i + 1 < n && (int)blockIdx.x >= descs[i + 1].first_block) ++i;
    const usip_split3_desc d = descs[i];
    const int bm = d.tile_rows;
    if ((int)threadIdx.x >= 2 * bm) return;
    const int ksteps = (d.K + XBK - 1) / XBK, blk = blockIdx.x - d.first_block;
    const int ks = blk % ksteps, mt = blk / ksteps;
    const int row = threadIdx.x % bm, half = threadIdx.x / bm;
    const int m = mt * bm + row;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = ks * XBK + half * 8 + j;
        v[j] = (m < d.M && k < d.K) ? d.At[(long long)k * d.lda + m] : 0.0f;
    }
    unsigned p[3][4];
    const int npl = d.reserved == 2 ? 2 : 3;
    if (npl == 2) {
        // the operand's scale, from the partial maxima absmax_multi_kernel left behind the two-plane image
        const float ws = x2h_weight_scale(d);
        if (blk == 0 && threadIdx.x == 0) x2h_trailer(d)[0] = ws;
#pragma unroll
        for (int j = 0; j < 4; ++j) split_pair_h(v[2 * j] * ws, v[2 * j + 1] * ws, p[0][j], p[1][j]);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) split_pair(v[2 * j], v[2 * j + 1], p[0][j], p[1][j], p[2][j]);
    }
    uint4* stage = reinterpret_cast<uint4*>(d.planes) + (long long)blk * (npl * bm * 2);
    for (int s = 0; s < npl; ++s)
        stage[s * (bm * 2) + row * 2 + ((half ^ (row >> 3)) & 1)] = make_uint4(p[s][0], p[s][1], p[s][2], p[s][3]);
}

// ------------------------------------------------------------------------------------------------
// amdgpu_waves_per_eu(2, 2): LDS already limits the kernel to two workgroups per CU; telling the register allocator
// so stops it from aiming at three waves per SIMD and spilling the staged A planes to scratch around the MFMAs.
// Tile: 2 x WN waves, each TM x 2 MFMA tiles of 32 x 32: XBM = 64*TM channels x XBN = 64*WN positions.
//   (TM, WN) = (2, 2): 128 x 128, 256 threads, two workgroups per CU   (layers with 128 output channels)
//              (4, 2): 256 x 128, 256 threads, two workgroups per CU   (the wide layers: every streamed element is
//                      prepared for 256 channels instead of 128 -- half the preparation per MFMA, half the L2 reads)
//              (4, 4): 256 x 256, 512 threads, one workgroup per CU
//   ASLOTS = 3 (round 5, 128-row tiles only: 12 KB more LDS still leave two workgroups per CU): the weight stage is
//   requested TWO stages ahead into a three-slot ring.  The M-sized launches of the second stage and the head (8192
//   positions: 128-320 workgroups, one per CU, a lone wave per SIMD) walk K = 256..640 serially and were paced by the
//   one-stage-ahead DMA's L2 round trip (~1.3 us per 16-k stage for ~0.5 us of work).
template <int PRO, int EPI, int TM, int WN, int NPL = 3, int ASLOTS = 2>
__global__ __launch_bounds__(128 * WN) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_x3p_kernel(const GemmArgs a, const uint4* __restrict__ planes)
{
    static_assert(ASLOTS == 2 || (ASLOTS == 3 && TM == 2), "the three-slot weight ring exists for the 128-row tile");
    constexpr int XBM = 64 * TM, XBN = 64 * WN, NT = 128 * WN;
    constexpr int APL = XBM * 32, BPL = XBN * 32;              // bytes of one plane of one stage
    constexpr int ASTAGE = NPL * APL, BSTAGE = NPL * BPL;
    constexpr int NA = XBM * 2 / NT;                           // 16-B chunks of an A plane per thread: 1 or 2
    constexpr bool POOL = (PRO == PRO_BN_BWD_POOL);
    constexpr bool TWO = (PRO == PRO_BN_BWD) || POOL;
    constexpr int NCOEF = (PRO == PRO_NONE) ? 0 : (TWO ? 4 : 2);
    constexpr int OPER_BYTES = ASLOTS * ASTAGE + 2 * BSTAGE;   // two (three) stages of the weights, two of the streamed operand
    // widest contraction of the path: 640 inputs (mlp1) forward, 512 outputs backward; 4 x 512 floats keep two
    // workgroups of the 256 x 128 tile inside the CU's 160 KiB
    constexpr int KMAX = (NCOEF == 4) ? 512 : 640;
    __shared__ __attribute__((aligned(16))) unsigned char smem[OPER_BYTES + (NCOEF ? NCOEF : 1) * KMAX * 4];
    unsigned char* As = smem;                                  // [stage][plane][row][16 k]
    unsigned char* Bs = smem + ASLOTS * ASTAGE;
    float* cf = reinterpret_cast<float*>(smem + OPER_BYTES);   // [NCOEF][KMAX]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // same logical tile order and XCD remap as the fp32 kernel
    const int tpc = (a.P + XBN - 1) / XBN, nmt = (a.M + XBM - 1) / XBM;
    const int total = a.nb * tpc * nmt;
    int L = blockIdx.x;
    if ((total & 7) == 0) L = (blockIdx.x & 7) * (total >> 3) + (blockIdx.x >> 3);
    const int mt = L % nmt, tn = L / nmt;
    const int b = tn / tpc, pt = tn % tpc;
    const int m0 = mt * XBM, p0 = pt * XBN;
    const int nk = (a.K + XBK - 1) / XBK;

    // Two fp16 planes: both operands are scaled by powers of two (exact) so that neither leaves the fp16 range.  The
    // weights carry their scale behind their image (absmax_multi_kernel); the streamed operand's comes from a rigorous
    // bound of what the prologue can produce -- forward: |gamma| sqrt(n) + |beta| for a BatchNorm output over the n
    // samples of THIS launch; backward: the bound usip_bn_backward_reduce_f32 left in row 4 of coef4 -- and is folded
    // into the prologue coefficients (relu(fma(x, s c0, s c1)) = s relu(fma(x, c0, c1)) exactly for s = 2^e > 0).
    float xs = 1.0f, out_scale = 1.0f;
    if (NPL == 2) {
        float* redm = reinterpret_cast<float*>(Bs);            // free until the first stage is written (a barrier later)
        float bnd = 0.f;
        if (PRO == PRO_AFFINE_RELU) {
            const float rn = sqrtf((float)a.nb * (float)a.P);
            for (int k = tid; k < a.K; k += NT) {
                const float c0 = a.coef[k], c1 = a.coef[a.K + k], mu = a.coef[2 * a.K + k], is = a.coef[3 * a.K + k];
                bnd = fmaxf(bnd, fabsf(c0) / is * rn + fabsf(__builtin_fmaf(mu, c0, c1)));
            }
        } else {
            for (int i = tid; i < (a.K + 63) / 64; i += NT) bnd = fmaxf(bnd, a.coef[4 * a.K + i]);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) bnd = fmaxf(bnd, __shfl_xor(bnd, off));
        if (lane == 0) redm[wave] = bnd;
        __syncthreads();
        bnd = redm[0];
#pragma unroll
        for (int w = 1; w < NT / 64; ++w) bnd = fmaxf(bnd, redm[w]);
        xs = pow2_scale(bnd, X2H_TOP);
        const float ws = __uint_as_float(planes[(long long)nmt * nk * (ASTAGE / 16)].x);
        out_scale = 1.0f / (xs * ws);
    }
    if (NCOEF) {
        for (int i = tid; i < NCOEF * a.K; i += NT) cf[(i / a.K) * KMAX + i % a.K] = a.coef[i] * xs;
    }

    // A: the global image IS the LDS image, so a stage of A is copied by LDS-DMA (global_load_lds_dwordx4: every lane
    // sends 16 B, the wave's 1 KiB lands at a wave-uniform LDS base) -- no staging registers, no ds_write
    const uint4* Ag = planes + (long long)mt * nk * (ASTAGE / 16) + tid;
    const int a_wave = wave * 64 * 16;                         // byte offset of this wave's first chunk in a plane
    // X: thread -> position p = tid % XBN, k-group = tid / XBN (wave-uniform): 8 consecutive k of one position
    const int xp = tid & (XBN - 1);
    const int xkg = __builtin_amdgcn_readfirstlane(tid / XBN);
    const unsigned xpc = (unsigned)min(p0 + xp, a.P - 1);
    const int x_lds = lds_off(xp, xkg);
    // The streamed operand is read through buffer descriptors built from wave-uniform values (the cloud's base, its
    // size): a load is `buffer_load_dword v, v_off, s[rsrc], s_row offen` -- one constant VGPR offset per lane, the row
    // as a scalar byte offset -- instead of a 64-bit vector add per load (12-20 v_lshl_add_u64 per stage in the r02
    // loop, which after halving the matrix work was a third of the loop's vector instructions).
    const int pgrp = POOL ? a.P / a.pool_group : 0;
    const unsigned cloud_bytes = (unsigned)a.K * (unsigned)a.P * 4u, pool_bytes = (unsigned)a.K * (unsigned)pgrp * 4u;
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((POOL ? a.X2 : a.X) + (long long)b * a.K * a.P), 0, cloud_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rX2 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((TWO ? a.X2 : a.X) + (long long)b * a.K * a.P), 0, cloud_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rPd = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(POOL ? a.pool_dp + (long long)b * a.K * pgrp : a.X), 0, POOL ? pool_bytes : 4u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rPa = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(POOL ? (const float*)(a.pool_arg + (long long)b * a.K * pgrp) : a.X), 0, POOL ? pool_bytes : 4u, 0x00020000);
    const int xoff = (int)(xpc * 4u);
    const int goff = POOL ? (int)((xpc / (unsigned)a.pool_group) * 4u) : 0;
    const int xkin = POOL ? (int)(xpc % (unsigned)a.pool_group) : 0;

    f32x16 acc[TM][2];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    float rx[8], ry[TWO ? 8 : 1];
    int rarg[POOL ? 8 : 1];

    auto dma_stage = [&](int buf, int kt) {
#pragma unroll
        for (int s = 0; s < NPL; ++s)
#pragma unroll
            for (int j = 0; j < NA; ++j)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(Ag + (long long)kt * (ASTAGE / 16) + s * (APL / 16) + j * NT),
                    (__attribute__((address_space(3))) void*)(As + buf * ASTAGE + s * APL + j * NT * 16 + a_wave), 16, 0, 0);
    };
    auto load_stage = [&](int kt) {
        const int kb = kt * XBK + xkg * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int kc = min(kb + i, a.K - 1);               // scalar: the k-group is wave-uniform
            if (POOL) {
                rx[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rPd, goff, kc * pgrp * 4, 0));
                rarg[i] = (int)__builtin_amdgcn_raw_buffer_load_b32(rPa, goff, kc * pgrp * 4, 0);
            } else {
                rx[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rX, xoff, kc * a.P * 4, st_aux<(PRO >= PRO_BN_BWD) ? 0 : LD_X3P_FWD>()));
            }
            if (TWO) ry[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rX2, xoff, kc * a.P * 4, 0));
        }
    };
    auto store_stage = [&](int buf, int kt) {
        const int kb = kt * XBK + xkg * 8;
        float v[8];
        if (TWO) {
            // dY = fma(c0, [fma(y, c0, c1) > 0 ? dZ : 0], fma(c2, y, c3)) on PAIRS of consecutive k, written with
            // 2-vectors in their natural order: packed fp32 FMAs without op_sel (the library is built without the SLP
            // vectoriser, usip_amd/build.py; scalar, this prologue made the pooled 512 x 512 data gradient 13 % slower).
            // The same operations in the same order as pro_apply<PRO_BN_BWD>: bit-identical.
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k0 = min(kb + 2 * j, a.K - 1), k1 = min(kb + 2 * j + 1, a.K - 1);
                float x0 = rx[2 * j], x1 = rx[2 * j + 1];
                if (POOL) {
                    x0 = (rarg[POOL ? 2 * j : 0] == xkin) ? x0 : 0.f;
                    x1 = (rarg[POOL ? 2 * j + 1 : 0] == xkin) ? x1 : 0.f;
                }
                const f32x2 W = {ry[TWO ? 2 * j : 0], ry[TWO ? 2 * j + 1 : 0]};
                const f32x2 C0 = {cf[k0], cf[k1]}, C1 = {cf[KMAX + k0], cf[KMAX + k1]};
                const f32x2 C2 = {cf[2 * KMAX + k0], cf[2 * KMAX + k1]}, C3 = {cf[3 * KMAX + k0], cf[3 * KMAX + k1]};
                const f32x2 T = __builtin_elementwise_fma(W, C0, C1);
                const f32x2 D = {T[0] > 0.0f ? x0 : 0.0f, T[1] > 0.0f ? x1 : 0.0f};
                const f32x2 O = __builtin_elementwise_fma(C0, D, __builtin_elementwise_fma(C2, W, C3));
                v[2 * j] = O[0];
                v[2 * j + 1] = O[1];
            }
        } else
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float x = rx[i];
            if (PRO != PRO_NONE) {
                const int kc = min(kb + i, a.K - 1);
                const float c0 = cf[kc], c1 = cf[KMAX + kc];
                float c2 = 0.f, c3 = 0.f, w = x;
                if (TWO) { c2 = cf[2 * KMAX + kc]; c3 = cf[3 * KMAX + kc]; w = ry[i]; }
                if (POOL) x = (rarg[POOL ? i : 0] == xkin) ? x : 0.f;
                constexpr int PA = POOL ? PRO_BN_BWD : PRO;
                x = pro_apply<PA>(x, w, c0, c1, c2, c3);
            }
            v[i] = x;
        }
        // k >= K (the tail of a 131-wide layer): the loads were clamped to row K-1 -- finite data -- and meet the
        // zero padding of the weight planes, so no masking is needed (a branch here would also split the loop body
        // and with it the MFMA / preparation interleave)
        unsigned p[3][4];
        if (NPL == 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) split_pair_h(v[2 * j], v[2 * j + 1], p[0][j], p[1][j]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) split_pair(v[2 * j], v[2 * j + 1], p[0][j], p[1][j], p[2][j]);
        }
#pragma unroll
        for (int s = 0; s < NPL; ++s)
            *reinterpret_cast<uint4*>(Bs + buf * BSTAGE + s * BPL + x_lds) =
                make_uint4(p[s][0], p[s][1], p[s][2], p[s][3]);
    };

    // Software pipeline, one barrier per stage:
    //   registers hold the RAW operands of stage kt+1 (loaded during stage kt-1);
    //   during the 24 MFMAs of stage kt the wave also (a) runs prologue + split on those registers and writes the
    //   LDS image of stage kt+1 into the other buffer, (b) re-issues the loads for stage kt+2 into the same
    //   registers.  In-order issue lets ~5 other instructions slip between two 32-cycle MFMAs, so the operand
    //   preparation overlaps the matrix pipe; hipcc interleaves the two streams on its own (pinning the order with
    //   sched_group_barrier was measured 8-25 % slower and removed; s_setprio(1) around the MFMAs: no change).  (Measured by
    //   ablation before this structure: MFMA + fragment reads 290 us, + global loads 145 us, + split/LDS writes
    //   110 us -- the three added up, 562-634 us for the 512 x 512 forward.)
    const int c = lane & 31, kh = lane >> 5;
    int fa_off[TM], fb_off[2];
#pragma unroll
    for (int t = 0; t < TM; ++t) fa_off[t] = lds_off((wm * TM + t) * 32 + c, kh);
#pragma unroll
    for (int t = 0; t < 2; ++t) fb_off[t] = lds_off(wn * 64 + t * 32 + c, kh);
    constexpr int NXL = POOL ? 24 : (TWO ? 16 : 8);            // register loads of one stage of the streamed operand
    dma_stage(0, 0);
    if (ASLOTS == 3) dma_stage(1, min(1, nk - 1));
    load_stage(0);
    __syncthreads();                                           // prologue coefficients are in LDS
    store_stage(0, 0);
    load_stage(min(1, nk - 1));
    __syncthreads();
    int cur = 0;
    int sa = 0;                                                // ASLOTS == 3: the ring slot that holds stage kt's weights
    // the six plane pairs, smallest terms first; operands swapped: D'[position][channel], see gemm_epilogue
#define USIP_X3_PRODUCT(PA_, PB_)                                                                                  \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                                           \
            if constexpr (NPL == 2) {                                                                              \
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fb[PB_][0]), __builtin_bit_cast(f16x8, fa[PA_][i]), acc[i][0], 0, 0, 0); \
                acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fb[PB_][1]), __builtin_bit_cast(f16x8, fa[PA_][i]), acc[i][1], 0, 0, 0); \
            } else {                                                                                               \
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[PB_][0], fa[PA_][i], acc[i][0], 0, 0, 0);    \
                acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[PB_][1], fa[PA_][i], acc[i][1], 0, 0, 0);    \
            }                                                                                                      \
        }
#define USIP_X3_READ_FRAGS()                                                                                       \
        bf16x8 fa[NPL][TM], fb[NPL][2];                                                                            \
        _Pragma("unroll") for (int s = 0; s < NPL; ++s) {                                                          \
            _Pragma("unroll") for (int t = 0; t < TM; ++t)                                                         \
                fa[s][t] = *reinterpret_cast<const bf16x8*>(As + (ASLOTS == 3 ? sa : cur) * ASTAGE + s * APL + fa_off[t]); \
            _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                          \
                fb[s][t] = *reinterpret_cast<const bf16x8*>(Bs + cur * BSTAGE + s * BPL + fb_off[t]);              \
        }
#define USIP_X3_FIRST_HALF()                                                                                       \
        if constexpr (NPL == 2) { USIP_X3_PRODUCT(1, 0) USIP_X3_PRODUCT(0, 1) }                                    \
        else { USIP_X3_PRODUCT(2, 0) USIP_X3_PRODUCT(0, 2) USIP_X3_PRODUCT(1, 1) }
#define USIP_X3_SECOND_HALF()                                                                                      \
        if constexpr (NPL == 2) { USIP_X3_PRODUCT(0, 0) }                                                          \
        else { USIP_X3_PRODUCT(1, 0) USIP_X3_PRODUCT(0, 1) USIP_X3_PRODUCT(0, 0) }
    for (int kt = 0; kt + 1 < nk; ++kt) {
        // A of stage kt+1: memory -> LDS (the other buffer is free); three slots: stage kt+2 into the slot stage kt-1 left
        // (in the last iterations a harmless repeat of the last stage, so that the counted wait below always has the same
        // instructions in front of it)
        // -- and it is issued BEHIND store_stage (below), not here: hipcc waits vmcnt(0) at the first use of the raw
        // operand registers inside store_stage whenever an LDS-DMA is in flight, i.e. for every DMA issued before that
        // point.  At the top of the stage that is a wait for a request made a few hundred cycles earlier (the two-slot
        // form: one L2 round trip per stage, exposed); behind store_stage the same drain comes one whole stage later.
        if (ASLOTS == 2) dma_stage(cur ^ 1, kt + 1);
        // no memory instruction may cross: the counted wait below relies on "DMA first, then the NXL register loads"
        // (ALU, MFMA and LDS instructions may still be scheduled across).  Measured and not kept (r03, f32x2 512 x 512
        // forward, same box): the DMA issued after store_stage, so that hipcc's vmcnt(0) at the first use of the
        // register loads does not also wait for it (284 vs 258 us: the compiler sinks it to the end of the stage);
        // three scheduling regions pinned with sched_barrier(0), 16 + 8 or 8 + 16 MFMAs around the memory
        // instructions (265 / 263 vs 254 us); the register loads as inline-asm buffer loads with a hand-counted
        // vmcnt(DMA count) in front of their first use instead of hipcc's vmcnt(0) (252 vs 260 us, inside the noise).
        // What the counters say instead (profiles/r03_pmc_x2_gemm.txt): the loop moves 48 KB per stage pair into the
        // CU -- A planes from L2, the streamed operand from HBM -- at ~25 GB/s per CU, the rate every streaming kernel
        // of this library tops out at; the matrix pipe is 38 % busy, the vector L1 has requests pending 60 % of the
        // time.  Larger tiles move fewer bytes per flop: 256 x 256 (one 8-wave workgroup per CU) is 8-14 % faster
        // stand-alone and 0.5 % SLOWER inside the step (operands then partly sit in the Infinity Cache).
        __builtin_amdgcn_sched_barrier(0x38F);
        USIP_X3_READ_FRAGS()
        USIP_X3_FIRST_HALF()
        store_stage(cur ^ 1, kt + 1);                          // X of stage kt+1: registers -> LDS
        if (ASLOTS == 3) {
            __builtin_amdgcn_sched_barrier(0);                 // (the DMA stays behind the first use of the operand registers)
            dma_stage(sa == 0 ? 2 : sa - 1, min(kt + 2, nk - 1));
        }
        load_stage(min(kt + 2, nk - 1));                       // X of stage kt+2: memory -> registers (last: a harmless repeat)
        USIP_X3_SECOND_HALF()
        // The DMA (issued before the register loads of stage kt+2, loads retire in order) must have landed and this
        // wave's LDS writes must be done before anyone reads the other buffer; the register loads stay in flight
        // across the barrier -- __syncthreads() would drain them (it waits vmcnt(0) while an LDS-DMA is pending).
        // (three slots: what may stay in flight is this stage's DMA -- stage kt+2 -- AND its register loads; everything
        // older, the DMA of stage kt+1 included, has then landed)
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(ASLOTS == 3 ? NPL * NA + NXL : NXL) : "memory");
        __builtin_amdgcn_s_barrier();
        cur ^= 1;
        sa = (sa == 2) ? 0 : sa + 1;
    }
    {
        USIP_X3_READ_FRAGS()
        USIP_X3_FIRST_HALF()
        USIP_X3_SECOND_HALF()
        __syncthreads();
    }
#undef USIP_X3_PRODUCT
#undef USIP_X3_READ_FRAGS
#undef USIP_X3_FIRST_HALF
#undef USIP_X3_SECOND_HALF
    if (NPL == 2) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] *= out_scale;
    }
    gemm_epilogue<2, WN, EPI, TM>(a, acc, reinterpret_cast<float*>(smem), OPER_BYTES / 4, b, m0, p0, tn, tpc);
}

// ------------------------------------------------------------------------------------------------
// Weight gradient dW[m][n] = sum_p pro(G)[m][p] * X[n][p] with f32x3 products.  Both operands are streamed
// activations (prologue + split on the fly), so what can be saved is how often an element is prepared: a 256 x 256
// output tile per workgroup (8 waves, each 128 x 64 = 4 x 2 MFMA tiles) prepares every element for 256 partners
// instead of 128 -- 16 elements per thread per 48 MFMAs (the 128 x 128 kernel: 16 per 24).  Same two-stage
// software pipeline as the GEMM; positions are the contraction index, 16 per stage, contiguous in memory, so a
// thread's float4 becomes 8 B of its LDS row directly.
template <int PRO, bool XPRO, int NPL = 3>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void wgrad_x3_kernel(const WgradArgs a)
{
    constexpr int TM = 4, TN = 2, WN = 4, BM = 256, BN = 256, NT = 512;
    constexpr int PL = BM * 32, STAGE = NPL * PL;              // bytes: one plane, one operand stage (BM == BN)
    constexpr bool POOL = (PRO == PRO_BN_BWD_POOL);
    constexpr bool TWO = (PRO == PRO_BN_BWD) || POOL;
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * STAGE];       // [stage][G | X][plane][row][16 p]
    unsigned char* Gs = smem;
    unsigned char* Xs = smem + 2 * STAGE;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int nmt = (a.M + BM - 1) / BM, nnt = (a.N + BN - 1) / BN;
    const int total = gridDim.x;
    int L = blockIdx.x;
    if ((total & 7) == 0) L = (blockIdx.x & 7) * (total >> 3) + (blockIdx.x >> 3);
    const int tile = L % (nmt * nnt), slice = L / (nmt * nnt);
    const int m0 = (tile / nnt) * BM, n0 = (tile % nnt) * BN;
    const int b = slice / a.segs, seg = slice % a.segs;
    const int pbeg = seg * a.seglen, pend = min(a.P, pbeg + a.seglen);
    const int nst = (pend - pbeg + 15) / 16;

    // thread -> two (row, 4 positions) pieces per operand and stage: f = tid + i*512, row = f / 4, kq = (f % 4) * 4
    int grow[2], xrow[2], lds[2];
    bool gok[2];
    const float* gp[2];
    const float* g2p[2];
    const float* xp[2];
    const float* pdp[2];
    const int* pap[2];
    const int pgrp = POOL ? a.P / a.pool_group : 0;
    const int kq = (tid & 3) * 4;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (tid + i * NT) >> 2;
        grow[i] = min(m0 + row, a.M - 1);
        xrow[i] = min(n0 + row, a.N - 1);
        gok[i] = m0 + row < a.M;
        lds[i] = lds_off(row, kq >> 3) + (kq & 7) * 2;
        gp[i] = POOL ? nullptr : a.G + ((long long)b * a.M + grow[i]) * a.P;
        g2p[i] = TWO ? a.G2 + ((long long)b * a.M + grow[i]) * a.P : nullptr;
        xp[i] = a.X + ((long long)b * a.N + xrow[i]) * a.P;
        pdp[i] = POOL ? a.pool_dp + ((long long)b * a.M + grow[i]) * pgrp : nullptr;
        pap[i] = POOL ? a.pool_arg + ((long long)b * a.M + grow[i]) * pgrp : nullptr;
    }
    // Two fp16 planes (NPL == 2, see gemm_x3p_kernel): G is scaled by 2^e from the bound in row 4 of coef4, X by 2^e
    // from |gamma| sqrt(n) + |beta| of its BatchNorm (xcoef = [4][N] batch statistics over the nb * P samples); both
    // folded into the per-row prologue coefficients, the partial tile is multiplied by 2^-(e_g + e_x) on the way out.
    float gscale = 1.0f, xscale = 1.0f;
    if (NPL == 2) {
        float* redm = reinterpret_cast<float*>(smem);          // free until the first stage is written (barrier below)
        float gb = 0.f, xb = 0.f;
        for (int i = tid; i < (a.M + 63) / 64; i += NT) gb = fmaxf(gb, a.coef[4 * a.M + i]);
        const float rn = sqrtf((float)a.nb * (float)a.P);
        for (int k = tid; k < a.N; k += NT) {
            const float c0 = a.xcoef[k], c1 = a.xcoef[a.N + k], mu = a.xcoef[2 * a.N + k], is = a.xcoef[3 * a.N + k];
            xb = fmaxf(xb, fabsf(c0) / is * rn + fabsf(__builtin_fmaf(mu, c0, c1)));
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { gb = fmaxf(gb, __shfl_xor(gb, off)); xb = fmaxf(xb, __shfl_xor(xb, off)); }
        if (lane == 0) { redm[wave] = gb; redm[8 + wave] = xb; }
        __syncthreads();
        gb = redm[0]; xb = redm[8];
#pragma unroll
        for (int w = 1; w < 8; ++w) { gb = fmaxf(gb, redm[w]); xb = fmaxf(xb, redm[8 + w]); }
        gscale = pow2_scale(gb, X2H_TOP);
        xscale = pow2_scale(xb, X2H_TOP);
        __syncthreads();                                       // everyone has read redm before the stages overwrite it
    }
    float gc[TWO ? 2 : 1][4], xc[XPRO ? 2 : 1][2];
    if (TWO) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) gc[i][j] = a.coef[j * a.M + grow[i]] * gscale;
    }
    if (XPRO) {
#pragma unroll
        for (int i = 0; i < 2; ++i) { xc[i][0] = a.xcoef[xrow[i]] * xscale; xc[i][1] = a.xcoef[a.N + xrow[i]] * xscale; }
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    float4 rg[2], rg2[TWO ? 2 : 1], rx[2];
    auto load_stage = [&](int st) {
        const int pc = min(pbeg + st * 16 + kq, a.P - 4);      // clamped: always a valid, aligned float4
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (POOL) {
                const int g = pc / a.pool_group;
                rg[i] = make_float4(pdp[i][g], __int_as_float(pap[i][g]), __int_as_float(pc % a.pool_group), 0.f);
            } else {
                rg[i] = ld_in4<LD_WGRAD3_G>(gp[i] + pc);
            }
            if (TWO) rg2[i] = ld_in4<LD_WGRAD3_G>(g2p[i] + pc);
            rx[i] = *reinterpret_cast<const float4*>(xp[i] + pc);
        }
    };
    auto store_stage = [&](int buf, int st) {
        const bool pok = pbeg + st * 16 + kq < pend;           // float4 granularity: P % 4 == 0, segments start at 32 p
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float4 v = rg[i];
            if (TWO) {
                const float4 w = rg2[i];
                if (POOL) {
                    const int hit = __float_as_int(v.y) - __float_as_int(v.z);
                    const float g = v.x;
                    v = make_float4(hit == 0 ? g : 0.f, hit == 1 ? g : 0.f, hit == 2 ? g : 0.f, hit == 3 ? g : 0.f);
                }
                const float c0 = gc[TWO ? i : 0][0], c1 = gc[TWO ? i : 0][1], c2 = gc[TWO ? i : 0][2], c3 = gc[TWO ? i : 0][3];
                v.x = pro_apply<PRO_BN_BWD>(v.x, w.x, c0, c1, c2, c3);
                v.y = pro_apply<PRO_BN_BWD>(v.y, w.y, c0, c1, c2, c3);
                v.z = pro_apply<PRO_BN_BWD>(v.z, w.z, c0, c1, c2, c3);
                v.w = pro_apply<PRO_BN_BWD>(v.w, w.w, c0, c1, c2, c3);
            }
            // only G is masked (rows beyond M, positions beyond the segment): whatever finite value the clamped X
            // load returned there meets a zero
            if (!(gok[i] && pok)) v = make_float4(0.f, 0.f, 0.f, 0.f);
            unsigned p0[2], p1[2], p2[2];
            if (NPL == 2) {
                split_pair_h(v.x, v.y, p0[0], p1[0]);
                split_pair_h(v.z, v.w, p0[1], p1[1]);
            } else {
                split_pair(v.x, v.y, p0[0], p1[0], p2[0]);
                split_pair(v.z, v.w, p0[1], p1[1], p2[1]);
            }
            *reinterpret_cast<uint2*>(Gs + buf * STAGE + 0 * PL + lds[i]) = make_uint2(p0[0], p0[1]);
            *reinterpret_cast<uint2*>(Gs + buf * STAGE + 1 * PL + lds[i]) = make_uint2(p1[0], p1[1]);
            if (NPL == 3) *reinterpret_cast<uint2*>(Gs + buf * STAGE + 2 * PL + lds[i]) = make_uint2(p2[0], p2[1]);
            float4 x = rx[i];
            if (XPRO) {
                const float s0 = xc[XPRO ? i : 0][0], s1 = xc[XPRO ? i : 0][1];
                x.x = fmaxf(__builtin_fmaf(x.x, s0, s1), 0.f); x.y = fmaxf(__builtin_fmaf(x.y, s0, s1), 0.f);
                x.z = fmaxf(__builtin_fmaf(x.z, s0, s1), 0.f); x.w = fmaxf(__builtin_fmaf(x.w, s0, s1), 0.f);
            }
            if (NPL == 2) {
                split_pair_h(x.x, x.y, p0[0], p1[0]);
                split_pair_h(x.z, x.w, p0[1], p1[1]);
            } else {
                split_pair(x.x, x.y, p0[0], p1[0], p2[0]);
                split_pair(x.z, x.w, p0[1], p1[1], p2[1]);
            }
            *reinterpret_cast<uint2*>(Xs + buf * STAGE + 0 * PL + lds[i]) = make_uint2(p0[0], p0[1]);
            *reinterpret_cast<uint2*>(Xs + buf * STAGE + 1 * PL + lds[i]) = make_uint2(p1[0], p1[1]);
            if (NPL == 3) *reinterpret_cast<uint2*>(Xs + buf * STAGE + 2 * PL + lds[i]) = make_uint2(p2[0], p2[1]);
        }
    };

    const int c = lane & 31, kh = lane >> 5;
    int fa_off[TM], fb_off[TN];
#pragma unroll
    for (int t = 0; t < TM; ++t) fa_off[t] = lds_off((wm * TM + t) * 32 + c, kh);
#pragma unroll
    for (int t = 0; t < TN; ++t) fb_off[t] = lds_off((wn * TN + t) * 32 + c, kh);
#define USIP_X3W_PRODUCT(PA_, PB_)                                                                                 \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                             \
            _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                       \
                if constexpr (NPL == 2)                                                                            \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[PA_][i]), __builtin_bit_cast(f16x8, fb[PB_][j]), acc[i][j], 0, 0, 0); \
                else                                                                                               \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA_][i], fb[PB_][j], acc[i][j], 0, 0, 0); \
            }
#define USIP_X3W_FIRST_HALF()                                                                                      \
        if constexpr (NPL == 2) { USIP_X3W_PRODUCT(1, 0) USIP_X3W_PRODUCT(0, 1) }                                  \
        else { USIP_X3W_PRODUCT(2, 0) USIP_X3W_PRODUCT(0, 2) USIP_X3W_PRODUCT(1, 1) }
#define USIP_X3W_SECOND_HALF()                                                                                     \
        if constexpr (NPL == 2) { USIP_X3W_PRODUCT(0, 0) }                                                         \
        else { USIP_X3W_PRODUCT(1, 0) USIP_X3W_PRODUCT(0, 1) USIP_X3W_PRODUCT(0, 0) }
#define USIP_X3W_READ_FRAGS()                                                                                      \
        bf16x8 fa[NPL][TM], fb[NPL][TN];                                                                           \
        _Pragma("unroll") for (int s = 0; s < NPL; ++s) {                                                          \
            _Pragma("unroll") for (int t = 0; t < TM; ++t)                                                         \
                fa[s][t] = *reinterpret_cast<const bf16x8*>(Gs + cur * STAGE + s * PL + fa_off[t]);                \
            _Pragma("unroll") for (int t = 0; t < TN; ++t)                                                         \
                fb[s][t] = *reinterpret_cast<const bf16x8*>(Xs + cur * STAGE + s * PL + fb_off[t]);                \
        }
    int cur = 0;
    if (nst > 0) {
        load_stage(0);
        store_stage(0, 0);
        load_stage(min(1, nst - 1));
        __syncthreads();
        for (int st = 0; st + 1 < nst; ++st) {
            USIP_X3W_READ_FRAGS()
            USIP_X3W_FIRST_HALF()
            store_stage(cur ^ 1, st + 1);                      // stage st+1: registers -> LDS
            load_stage(min(st + 2, nst - 1));                  // stage st+2: memory -> registers
            USIP_X3W_SECOND_HALF()
            __syncthreads();
            cur ^= 1;
        }
        USIP_X3W_READ_FRAGS()
        USIP_X3W_FIRST_HALF()
        USIP_X3W_SECOND_HALF()
    }
#undef USIP_X3W_PRODUCT
#undef USIP_X3W_READ_FRAGS
#undef USIP_X3W_FIRST_HALF
#undef USIP_X3W_SECOND_HALF
    if (NPL == 2) {
        const float out_scale = 1.0f / (gscale * xscale);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] *= out_scale;
    }
    wgrad_store_partial<TM, TN>(a, acc, slice, m0, n0, wm, wn, lane);
}


// ------------------------------------------------------------------------------------------------
// The f32x2 weight gradient (wgrad_x3_kernel<PRO, true, 2>) with FULL-LINE loads.  Round 5.
// wgrad_x3_kernel fetches, per 16-position stage, 64 B of each of a tile's 256 + 256 (+ 256) operand rows: half a
// 128-B line per row and stage, the other half one stage later -- by which time the 32 workgroups of an XCD have
// touched 3 MB of lines in a 4 MB L2.  rocprofv3 FETCH_SIZE of the round-4 step: 641 MB per launch against 492 MB of
// operands (256 x 256 and 512 x 256 launches, 5.6 TB/s of counted traffic: the launch IS at what HBM delivers, a
// quarter of it fetched twice), 827 against 537 MB for the pooled 512 x 512 launch.  Round 4 tried "both halves back to
// back" by loading two stages per event and lost a stage of prefetch distance (25 % fewer bytes, 15 % slower).
// Here the 16-position stages, their order, their MFMAs and the registers per thread stay exactly as they are --
// partial tiles are BIT-IDENTICAL to wgrad_x3_kernel's -- and only WHICH 16 KB per operand a load event fetches
// changes: event 2j covers rows [0, 128) x the 32 positions of stage pair j, event 2j+1 rows [128, 256) x the same 32
// positions: eight consecutive lanes read one whole line.  A pair of stages is complete in LDS when both its events
// are stored, so LDS holds two PAIRS (8 stage images = 128 KiB; one workgroup per CU as before), every event is
// still issued one stage before it is stored, and the stores of a pair's first stage and the reads of its second
// touch different buffers: ONE barrier per 32 positions instead of two.
template <int PRO>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void wgrad_x2l_kernel(const WgradArgs a)
{
    constexpr int TM = 4, TN = 2, WN = 4, BM = 256, BN = 256, NT = 512;
    constexpr int PL = BM * 32, STAGE = 2 * PL;                // bytes: one plane, one operand stage (two fp16 planes)
    constexpr bool POOL = (PRO == PRO_BN_BWD_POOL);
    static_assert(PRO == PRO_BN_BWD || PRO == PRO_BN_BWD_POOL, "two-plane weight gradient: both operands carry a bound");
    // [G | X][pair buffer][stage of the pair][plane][row][16 p].  The image of a pair's SECOND stage starts SHIFT = 64 B
    // late: a 16-lane group of a store instruction covers two rows x both stages of the pair, and with the images exactly
    // 16 KiB apart the two stages of a row met in the same eight banks (SQ_LDS_BANK_CONFLICT = 28 % of the LDS cycles in
    // the first build, profiles/r05d_pmc_wgrad.txt); 16 banks apart the group's 16 lanes cover all 32 banks once.
    constexpr int SHIFT = 64, PAIR = 2 * (STAGE + SHIFT);      // bytes of one pair buffer: image 0 at 0, image 1 at STAGE + SHIFT
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * PAIR];
    unsigned char* Gs = smem;
    unsigned char* Xs = smem + 2 * PAIR;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int nmt = (a.M + BM - 1) / BM, nnt = (a.N + BN - 1) / BN;
    const int total = gridDim.x;
    int L = blockIdx.x;
    if ((total & 7) == 0) L = (blockIdx.x & 7) * (total >> 3) + (blockIdx.x >> 3);
    const int tile = L % (nmt * nnt), slice = L / (nmt * nnt);
    const int m0 = (tile / nnt) * BM, n0 = (tile % nnt) * BN;
    const int b = slice / a.segs, seg = slice % a.segs;
    const int pbeg = seg * a.seglen, pend = min(a.P, pbeg + a.seglen);
    const int nst = (pend - pbeg + 15) / 16;
    const int npair = (nst + 1) / 2;                           // an odd last stage is multiplied with G = 0: adds +-0

    // thread -> per event two (row, 4 positions) pieces per operand: rows h * 128 + i * 64 + tid / 8 (h = the event's
    // row half), positions kq .. kq + 3 of the pair's 32: lanes 8 r .. 8 r + 7 cover one 128-B line of row r
    const int rl = tid >> 3;
    const int kq = (tid & 7) * 4, sub = kq >> 4, kk = kq & 15;
    // (row + 64) >> 3 and (row + 128) >> 3 keep the parity of row >> 3: the swizzle of lds_off is that of row rl
    const int lds0 = lds_off(rl, kk >> 3) + (kk & 7) * 2 + sub * (STAGE + SHIFT);
    const int pgrp = POOL ? a.P / a.pool_group : 0;
    const float* const Gb = POOL ? nullptr : a.G + (long long)b * a.M * a.P;
    const float* const G2b = a.G2 + (long long)b * a.M * a.P;
    const float* const Xb = a.X + (long long)b * a.N * a.P;
    const float* const Pdb = POOL ? a.pool_dp + (long long)b * a.M * pgrp : nullptr;
    const int* const Pab = POOL ? a.pool_arg + (long long)b * a.M * pgrp : nullptr;
    int goff[2][2], xoff[2][2], pgoff[2][2];                   // element offsets inside the cloud (max(M, N) * P < 2^30: launcher)
    bool gok[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = h * 128 + i * 64 + rl;
            const int grow = min(m0 + row, a.M - 1), xrow = min(n0 + row, a.N - 1);
            gok[h][i] = m0 + row < a.M;
            goff[h][i] = grow * a.P;
            xoff[h][i] = xrow * a.P;
            pgoff[h][i] = POOL ? grow * pgrp : 0;
        }
    // operand scales: as in wgrad_x3_kernel<.., 2>
    float gscale, xscale;
    {
        float* redm = reinterpret_cast<float*>(smem);          // free until the first stage is written (barrier below)
        float gb = 0.f, xb = 0.f;
        for (int i = tid; i < (a.M + 63) / 64; i += NT) gb = fmaxf(gb, a.coef[4 * a.M + i]);
        const float rn = sqrtf((float)a.nb * (float)a.P);
        for (int k = tid; k < a.N; k += NT) {
            const float c0 = a.xcoef[k], c1 = a.xcoef[a.N + k], mu = a.xcoef[2 * a.N + k], is = a.xcoef[3 * a.N + k];
            xb = fmaxf(xb, fabsf(c0) / is * rn + fabsf(__builtin_fmaf(mu, c0, c1)));
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { gb = fmaxf(gb, __shfl_xor(gb, off)); xb = fmaxf(xb, __shfl_xor(xb, off)); }
        if (lane == 0) { redm[wave] = gb; redm[8 + wave] = xb; }
        __syncthreads();
        gb = redm[0]; xb = redm[8];
#pragma unroll
        for (int w = 1; w < 8; ++w) { gb = fmaxf(gb, redm[w]); xb = fmaxf(xb, redm[8 + w]); }
        gscale = pow2_scale(gb, X2H_TOP);
        xscale = pow2_scale(xb, X2H_TOP);
        __syncthreads();                                       // everyone has read redm before the stages overwrite it
    }
    float gc[2][2][4], xc[2][2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int grow = min(m0 + h * 128 + i * 64 + rl, a.M - 1), xrow = min(n0 + h * 128 + i * 64 + rl, a.N - 1);
#pragma unroll
            for (int j = 0; j < 4; ++j) gc[h][i][j] = a.coef[j * a.M + grow] * gscale;
            xc[h][i][0] = a.xcoef[xrow] * xscale;
            xc[h][i][1] = a.xcoef[a.N + xrow] * xscale;
        }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    float4 rg[2], rg2[2], rx[2];
    auto load_event = [&](auto hc, int pair) {
        constexpr int H = decltype(hc)::value;
        const int pc = min(pbeg + pair * 32 + kq, a.P - 4);    // clamped: always a valid, aligned float4
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (POOL) {
                const int g = pc / a.pool_group;
                rg[i] = make_float4(Pdb[pgoff[H][i] + g], __int_as_float(Pab[pgoff[H][i] + g]),
                                    __int_as_float(pc % a.pool_group), 0.f);
            } else {
                rg[i] = ld_in4<LD_WGRAD_G>(Gb + goff[H][i] + pc);
            }
            rg2[i] = ld_in4<LD_WGRAD_G>(G2b + goff[H][i] + pc);
            rx[i] = ld_in4<LD_WGRAD_X>(Xb + xoff[H][i] + pc);
        }
    };
    auto store_event = [&](auto hc, int pbuf, int pair) {
        constexpr int H = decltype(hc)::value;
        const bool pok = pbeg + pair * 32 + kq < pend;         // float4 granularity: P % 4 == 0, segments start at 32 p
        unsigned char* const gdst = Gs + pbuf * PAIR + lds0 + H * 4096;
        unsigned char* const xdst = Xs + pbuf * PAIR + lds0 + H * 4096;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float4 v = rg[i];
            const float4 w = rg2[i];
            if (POOL) {
                const int hit = __float_as_int(v.y) - __float_as_int(v.z);
                const float g = v.x;
                v = make_float4(hit == 0 ? g : 0.f, hit == 1 ? g : 0.f, hit == 2 ? g : 0.f, hit == 3 ? g : 0.f);
            }
            const float c0 = gc[H][i][0], c1 = gc[H][i][1], c2 = gc[H][i][2], c3 = gc[H][i][3];
            v.x = pro_apply<PRO_BN_BWD>(v.x, w.x, c0, c1, c2, c3);
            v.y = pro_apply<PRO_BN_BWD>(v.y, w.y, c0, c1, c2, c3);
            v.z = pro_apply<PRO_BN_BWD>(v.z, w.z, c0, c1, c2, c3);
            v.w = pro_apply<PRO_BN_BWD>(v.w, w.w, c0, c1, c2, c3);
            if (!(gok[H][i] && pok)) v = make_float4(0.f, 0.f, 0.f, 0.f);
            unsigned p0[2], p1[2];
            split_pair_h(v.x, v.y, p0[0], p1[0]);
            split_pair_h(v.z, v.w, p0[1], p1[1]);
            *reinterpret_cast<uint2*>(gdst + 0 * PL + i * 2048) = make_uint2(p0[0], p0[1]);
            *reinterpret_cast<uint2*>(gdst + 1 * PL + i * 2048) = make_uint2(p1[0], p1[1]);
            float4 x = rx[i];
            const float s0 = xc[H][i][0], s1 = xc[H][i][1];
            x.x = fmaxf(__builtin_fmaf(x.x, s0, s1), 0.f); x.y = fmaxf(__builtin_fmaf(x.y, s0, s1), 0.f);
            x.z = fmaxf(__builtin_fmaf(x.z, s0, s1), 0.f); x.w = fmaxf(__builtin_fmaf(x.w, s0, s1), 0.f);
            split_pair_h(x.x, x.y, p0[0], p1[0]);
            split_pair_h(x.z, x.w, p0[1], p1[1]);
            *reinterpret_cast<uint2*>(xdst + 0 * PL + i * 2048) = make_uint2(p0[0], p0[1]);
            *reinterpret_cast<uint2*>(xdst + 1 * PL + i * 2048) = make_uint2(p1[0], p1[1]);
        }
    };

    const int c = lane & 31, kh = lane >> 5;
    int fa_off[TM], fb_off[TN];
#pragma unroll
    for (int t = 0; t < TM; ++t) fa_off[t] = lds_off((wm * TM + t) * 32 + c, kh);
#pragma unroll
    for (int t = 0; t < TN; ++t) fb_off[t] = lds_off((wn * TN + t) * 32 + c, kh);
#define USIP_X2L_PRODUCT(PA_, PB_)                                                                                 \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                             \
            _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                         \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[PA_][i]), __builtin_bit_cast(f16x8, fb[PB_][j]), acc[i][j], 0, 0, 0);
#define USIP_X2L_READ_FRAGS(SB_)                                                                                   \
        bf16x8 fa[2][TM], fb[2][TN];                                                                               \
        _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                            \
            _Pragma("unroll") for (int t = 0; t < TM; ++t)                                                         \
                fa[s][t] = *reinterpret_cast<const bf16x8*>(Gs + ((SB_) >> 1) * PAIR + ((SB_) & 1) * (STAGE + SHIFT) + s * PL + fa_off[t]);              \
            _Pragma("unroll") for (int t = 0; t < TN; ++t)                                                         \
                fb[s][t] = *reinterpret_cast<const bf16x8*>(Xs + ((SB_) >> 1) * PAIR + ((SB_) & 1) * (STAGE + SHIFT) + s * PL + fb_off[t]);              \
        }
    using H0 = std::integral_constant<int, 0>;
    using H1 = std::integral_constant<int, 1>;
    if (npair > 0) {
        load_event(H0{}, 0);
        store_event(H0{}, 0, 0);
        load_event(H1{}, 0);
        store_event(H1{}, 0, 0);
        load_event(H0{}, min(1, npair - 1));
        __syncthreads();
        int pb = 0;
        for (int pr = 0; pr + 1 < npair; ++pr) {
            {   // first stage of pair pr; rows [0, 128) of pair pr + 1 go to LDS, rows [128, 256) are requested
                USIP_X2L_READ_FRAGS(pb * 2)
                USIP_X2L_PRODUCT(1, 0) USIP_X2L_PRODUCT(0, 1)
                store_event(H0{}, pb ^ 1, pr + 1);
                load_event(H1{}, pr + 1);
                USIP_X2L_PRODUCT(0, 0)
            }
            {   // second stage (its buffers are not written in this iteration: no barrier in between)
                USIP_X2L_READ_FRAGS(pb * 2 + 1)
                USIP_X2L_PRODUCT(1, 0) USIP_X2L_PRODUCT(0, 1)
                store_event(H1{}, pb ^ 1, pr + 1);
                load_event(H0{}, min(pr + 2, npair - 1));      // (last: a harmless repeat)
                USIP_X2L_PRODUCT(0, 0)
            }
            __syncthreads();
            pb ^= 1;
        }
        {
            USIP_X2L_READ_FRAGS(pb * 2)
            USIP_X2L_PRODUCT(1, 0) USIP_X2L_PRODUCT(0, 1) USIP_X2L_PRODUCT(0, 0)
        }
        {
            USIP_X2L_READ_FRAGS(pb * 2 + 1)
            USIP_X2L_PRODUCT(1, 0) USIP_X2L_PRODUCT(0, 1) USIP_X2L_PRODUCT(0, 0)
        }
    }
#undef USIP_X2L_PRODUCT
#undef USIP_X2L_READ_FRAGS
    {
        const float out_scale = 1.0f / (gscale * xscale);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] *= out_scale;
    }
    wgrad_store_partial<TM, TN>(a, acc, slice, m0, n0, wm, wn, lane);
}


// ------------------------------------------------------------------------------------------------
// f32x2 GEMM for the 128-wide layers (M <= 128 outputs, K <= 128 inputs: conv5 of RPN_Detector_Ball, the second
// PointNet of RPN_Detector, the descriptor's layers) with the WEIGHT FRAGMENTS RESIDENT IN REGISTERS.
// The tile kernel above re-reads the layer's 64 KB of weight planes from L2 for every 128-position tile: at K = 128
// that is as many bytes as the streamed operand itself, and these layers are bound by what a CU's memory path carries
// (~25 GB/s, L2 hits included).  Here a wave owns 32 output channels for the kernel's lifetime and keeps
// their fragments of both planes for all K in 64 VGPRs; persistent workgroups (two per CU) walk 64-position tiles,
// so a CU moves the streamed operand in and the output out -- 2/3 of the bytes.
//   workgroup = 4 waves = 128 channels x 64 positions; a tile is processed in slabs of 64 k (one barrier per slab, 24
//   MFMAs per wave between barriers); thread -> (position tid % 64, 16 consecutive k of the slab): loads 16 (32, 48)
//   values with buffer loads, applies the prologue, splits, writes two 16-B chunks per plane into the slab's LDS image
//   ([plane][position][64 k], 128-B rows, 16-B chunks XOR-swizzled by the position: conflict-free ds_read_b128);
//   the slab of step q+1 is prepared and the loads of step q+2 are in flight while step q is multiplied.
//   The accumulator is D[channel][position] (weights as the MFMA A operand, as in narrow_fwd.hip): a lane holds ONE
//   position and 16 channels per 32 x 32 tile, so every store instruction writes two full 128-B lines (the first
//   version used the tile kernel's D'[position][channel] epilogue -- 16-B pieces of 32 different rows per store
//   instruction -- and ran no faster than the tile kernel, 144 vs 160 us for conv5's forward: the store path, not
//   the loads, was what the CU was short of).  BatchNorm statistics are per-lane running sums over all the
//   workgroup's tiles, reduced over the 32 positions of a half-wave once at the end: one partial per channel and
//   WORKGROUP, no LDS, no barrier in the epilogue.
template <int PRO, int EPI, bool RB = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_x2r_kernel(const GemmArgs a, const uint4* __restrict__ planes)
{
    constexpr bool POOL = (PRO == PRO_BN_BWD_POOL);
    constexpr bool TWO = (PRO == PRO_BN_BWD) || POOL;
    constexpr int NCOEF = TWO ? 4 : 2;
    constexpr int BN = 64, SLAB = 64, PLB = BN * SLAB * 2;     // bytes of one plane of one slab: 8 KiB
    __shared__ __attribute__((aligned(16))) unsigned char Bs[2][2][PLB];   // [buffer][plane]
    __shared__ float cf[NCOEF][128];
    __shared__ float redm[4];
    // RB: the tile's row bias (rowbias[b][channel][position / rb_group], rb_group a multiple of 32: one value per channel
    // and 32-position half of the tile), fetched with the tile's first slab, double-buffered by the tile's parity
    __shared__ float rbs[RB ? 2 : 1][RB ? 128 : 1][2];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, kh = lane >> 5;
    const int nk = (a.K + XBK - 1) / XBK;                      // 16-k steps (<= 8)
    const int nslab = (a.K + SLAB - 1) / SLAB;                 // 1 or 2
    const int tpc = (a.P + BN - 1) / BN, total = a.nb * tpc;

    // operand scales (see gemm_x3p_kernel)
    float bnd = 0.f;
    if (PRO == PRO_AFFINE_RELU) {
        const float rn = sqrtf((float)a.nb * (float)a.P);
        for (int k = tid; k < a.K; k += 256) {
            const float c0 = a.coef[k], c1 = a.coef[a.K + k], mu = a.coef[2 * a.K + k], is = a.coef[3 * a.K + k];
            bnd = fmaxf(bnd, fabsf(c0) / is * rn + fabsf(__builtin_fmaf(mu, c0, c1)));
        }
    } else {
        for (int i = tid; i < (a.K + 63) / 64; i += 256) bnd = fmaxf(bnd, a.coef[4 * a.K + i]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) bnd = fmaxf(bnd, __shfl_xor(bnd, off));
    if (lane == 0) redm[wave] = bnd;
    __syncthreads();
    const float xs = pow2_scale(fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3])), X2H_TOP);
    const float ws = __uint_as_float(planes[(long long)nk * (2 * 128 * 32 / 16)].x);      // behind the image (one M tile)
    const float out_scale = 1.0f / (xs * ws);
    for (int i = tid; i < NCOEF * 128; i += 256) {
        const int r = i / 128, k = i % 128;
        cf[r][k] = (k < a.K) ? a.coef[r * a.K + k] * xs : 0.f;
    }

    // resident weight fragments: step s, plane pl -> 8 consecutive k (half kh of the step) of channel 32 wave + c
    bf16x8 fa[8][2];
    {
        const int row = wave * 32 + c;
        const int chunk = row * 2 + ((kh ^ (row >> 3)) & 1);
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (s8 < nk) v = planes[(long long)(s8 * 2 + pl) * 256 + chunk];
                fa[s8][pl] = __builtin_bit_cast(bf16x8, v);
            }
    }
    __shared__ float bsh[128];
    if (tid < 128) bsh[tid] = (a.bias && tid < a.M) ? a.bias[tid] : 0.f;     // visible after the barriers below
    float s1[EPI == EPI_STATS ? 16 : 1], s2[EPI == EPI_STATS ? 16 : 1];
    if (EPI == EPI_STATS) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { s1[r] = 0.f; s2[r] = 0.f; }
    }

    // streamed operand: thread -> position xp of the tile, 16 consecutive k (k-group xkg, wave-uniform) of a slab
    const int xp = tid & 63;
    const int xkg = wave;
    const int pgrp = POOL ? a.P / a.pool_group : 0;
    const unsigned cloud_bytes = (unsigned)a.K * (unsigned)a.P * 4u, pool_bytes = (unsigned)a.K * (unsigned)pgrp * 4u;

    // raw operands of one slab step: TWO register sets, so that the loads of step q+3 are issued while step q is
    // multiplied (with one set -- 4 KiB in flight per wave, 32 KiB per CU at two 4-wave workgroups -- the kernel ran at
    // the 8 B/clk per CU that 32 KiB cover at ~2 us of loaded latency: 144 us for conv5's forward)
    struct Raw { float x[16]; float y[TWO ? 16 : 1]; int arg[POOL ? 16 : 1]; int xkin; float rb; };
    const int ngrp = RB ? a.P / a.rb_group : 0;
    auto load_slab = [&](int tile, int sl, Raw& r) {
        const int b = tile / tpc, p0 = (tile - b * tpc) * BN;
        if (RB && sl == 0) {                                   // thread -> (channel tid >> 1, half tid & 1)
            const int ch = min(tid >> 1, a.M - 1), pp = min(p0 + (tid & 1) * 32, a.P - 1);
            r.rb = a.rowbias[((long long)b * a.M + ch) * ngrp + pp / a.rb_group];
        }
        const unsigned xpc = (unsigned)min(p0 + xp, a.P - 1);
        const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(
            (void*)((POOL ? a.X2 : a.X) + (long long)b * a.K * a.P), 0, cloud_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rX2 = __builtin_amdgcn_make_buffer_rsrc(
            (void*)((TWO ? a.X2 : a.X) + (long long)b * a.K * a.P), 0, cloud_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rPd = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(POOL ? a.pool_dp + (long long)b * a.K * pgrp : a.X), 0, POOL ? pool_bytes : 4u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rPa = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(POOL ? (const float*)(a.pool_arg + (long long)b * a.K * pgrp) : a.X), 0, POOL ? pool_bytes : 4u, 0x00020000);
        const int xoff = (int)(xpc * 4u);
        const int goff = POOL ? (int)((xpc / (unsigned)a.pool_group) * 4u) : 0;
        r.xkin = POOL ? (int)(xpc % (unsigned)a.pool_group) : 0;
        const int kb = sl * SLAB + xkg * 16;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int kc = min(kb + i, a.K - 1);               // rows beyond K meet zero weight planes
            if (POOL) {
                r.x[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rPd, goff, kc * pgrp * 4, 0));
                r.arg[POOL ? i : 0] = (int)__builtin_amdgcn_raw_buffer_load_b32(rPa, goff, kc * pgrp * 4, 0);
            } else {
                r.x[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rX, xoff, kc * a.P * 4, st_aux<LD_X2R>()));
            }
            if (TWO) r.y[TWO ? i : 0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rX2, xoff, kc * a.P * 4, 0));
        }
    };
    auto store_slab = [&](int buf, int sl, const Raw& r, int par) {
        if (RB && sl == 0) rbs[RB ? par : 0][RB ? tid >> 1 : 0][tid & 1] = r.rb;
        const int kb = sl * SLAB + xkg * 16;
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int kc = min(kb + i, 127);
            const float c0 = cf[0][kc], c1 = cf[1][kc];
            float c2 = 0.f, c3 = 0.f, w = r.x[i], x = r.x[i];
            if (TWO) { c2 = cf[TWO ? 2 : 0][kc]; c3 = cf[TWO ? 3 : 0][kc]; w = r.y[TWO ? i : 0]; }
            if (POOL) x = (r.arg[POOL ? i : 0] == r.xkin) ? x : 0.f;
            constexpr int PA = POOL ? PRO_BN_BWD : PRO;
            v[i] = pro_apply<PA>(x, w, c0, c1, c2, c3);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {                          // the thread's two 8-k chunks of the slab
            unsigned p0[4], p1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) split_pair_h(v[h * 8 + 2 * j], v[h * 8 + 2 * j + 1], p0[j], p1[j]);
            const int chunk = (xkg * 2 + h) ^ (xp & 7);
            *reinterpret_cast<uint4*>(&Bs[buf][0][xp * 128 + chunk * 16]) = make_uint4(p0[0], p0[1], p0[2], p0[3]);
            *reinterpret_cast<uint4*>(&Bs[buf][1][xp * 128 + chunk * 16]) = make_uint4(p1[0], p1[1], p1[2], p1[3]);
        }
    };

    // linear sequence of steps q = (tile iteration, slab); persistent over tiles blockIdx.x, + gridDim.x, ...
    const int G = gridDim.x;
    const int first = blockIdx.x;
    if (first >= total) return;
    const int my_tiles = (total - first + G - 1) / G;
    const int nsteps = my_tiles * nslab;
    auto tile_of = [&](int q) { return first + (q / nslab) * G; };
    auto slab_of = [&](int q) { return q % nslab; };

    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    constexpr int DIST = POOL ? 1 : 2;                         // (the pooled form has three arrays per set: one set)
    Raw rA, rB;
    load_slab(tile_of(0), slab_of(0), rA);
    if (DIST == 2 && nsteps > 1) load_slab(tile_of(1), slab_of(1), rB);
    __syncthreads();                                           // cf visible
    store_slab(0, slab_of(0), rA, 0);
    if (nsteps > DIST) load_slab(tile_of(DIST), slab_of(DIST), rA);
    __syncthreads();
    int cur = 0;
    // one step: multiply slab q (LDS buffer cur), turn the registers of step q+1 (`nxt`) into the other buffer, refill
    // them with step q+3
    auto step = [&](int q, Raw& nxt) {
        const int sl = slab_of(q);
        // fragments of the slab's four 16-k steps: position tile j, step s, plane pl
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            bf16x8 fb[2][2];
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int pos = j * 32 + c;
                    fb[pl][j] = *reinterpret_cast<const bf16x8*>(&Bs[cur][pl][pos * 128 + (((s4 * 2 + kh) ^ (pos & 7)) * 16)]);
                }
#pragma unroll
            for (int half = 0; half < 2; ++half) {             // slab 0 -> steps 0..3, slab 1 -> steps 4..7 (static indices)
                if (half == sl) {
                    const bf16x8 w0 = fa[half * 4 + s4][0], w1 = fa[half * 4 + s4][1];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w0), __builtin_bit_cast(f16x8, fb[1][j]), acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w1), __builtin_bit_cast(f16x8, fb[0][j]), acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w0), __builtin_bit_cast(f16x8, fb[0][j]), acc[j], 0, 0, 0);
                    }
                }
            }
        }
        if (q + 1 < nsteps) {
            store_slab(cur ^ 1, slab_of(q + 1), nxt, ((q + 1) / nslab) & 1);
            if (q + 1 + DIST < nsteps) load_slab(tile_of(q + 1 + DIST), slab_of(q + 1 + DIST), nxt);
        }
        if (sl == nslab - 1) {
            // tile finished: + bias, store, statistics.  lane = position j*32 + c of the tile; register r of position tile j
            // is channel 32 wave + 8 (r >> 2) + 4 kh + (r & 3): one store instruction = 32 consecutive positions of two rows
            const int tile = tile_of(q);
            const int b = tile / tpc, pt = tile - b * tpc, p0 = pt * BN;
            float* yb = a.Y + (long long)b * a.y_rows * a.P;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int pp = p0 + j * 32 + c;
                const bool pok = pp < a.P;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ch = wave * 32 + 8 * (r >> 2) + 4 * kh + (r & 3);
                    float v = __builtin_fmaf(acc[j][r], out_scale, bsh[ch]);
                    if (RB) v += rbs[RB ? (q / nslab) & 1 : 0][RB ? ch : 0][j];
                    if (pok && ch < a.M) {
                        st_out<RB ? ST_X2R : ST_X2R_PLAIN>(yb + (long long)ch * a.P + pp, v);
                        if (EPI == EPI_STATS) { s1[EPI == EPI_STATS ? r : 0] += v; s2[EPI == EPI_STATS ? r : 0] = __builtin_fmaf(v, v, s2[EPI == EPI_STATS ? r : 0]); }
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                          // register loads stay in flight across it
        cur ^= 1;
    };
    for (int q = 0; q < nsteps; q += 2) {
        step(q, DIST == 2 ? rB : rA);                          // step q+1's registers are set B, q+2's set A, ...
        if (q + 1 < nsteps) step(q + 1, rA);
    }
    if (EPI == EPI_STATS) {
        // per channel: the 32 positions of a half-wave (two 16-lane DPP row sums and one exchange) -> lane c == 0
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float u = usip_row16_sum(s1[EPI == EPI_STATS ? r : 0]), v = usip_row16_sum(s2[EPI == EPI_STATS ? r : 0]);
            u += __shfl_xor(u, 16);
            v += __shfl_xor(v, 16);
            const int ch = wave * 32 + 8 * (r >> 2) + 4 * kh + (r & 3);
            if (c == 0 && ch < a.M) {
                a.stats[(long long)ch * G + blockIdx.x] = u;
                a.stats[(long long)G * a.M + (long long)ch * G + blockIdx.x] = v;
            }
        }
    }
}

}  // namespace

// Rows of the tile usip_mlp_gemm_x3p_f32 / _x2h_f32 use for an M-row operand in a launch over nb clouds of P positions
// (= rows per block of the split image the launch expects).  256-row tiles halve the operand preparation per MFMA, but
// the launch still has to fill the chip: the second-stage and head layers (512 positions per cloud) are 128-192
// workgroups of 256 x 128 -- with 128-row tiles 640 -> 512 over 16 x 512 positions runs in 36 instead of 46 us (fp32
// MFMA kernel: 73).
extern "C" int usip_mlp_x3p_tile_rows(int M, int P, int nb)
{
    const int t = usip_tuning_value(USIP_TUNE_X3_GEMM_TILE);        // measurement: 1 = always 128-row tiles
    if (M <= 128 || t == 1) return 128;
    const long long wide = (long long)((M + 255) / 256) * nb * ((P + 127) / 128);
    return wide >= 256 ? 256 : 128;
}


// Bytes of the split weight image usip_mlp_split3_f32 writes for an M x K operand.
extern "C" long long usip_mlp_split3_bytes(int M, int K)
{
    if (M < 1 || K < 1) return 0;
    const int bm = M > 128 ? 256 : 128;                        // covers either tiling of the rows
    return (long long)((M + bm - 1) / bm) * ((K + XBK - 1) / XBK) * 3 * bm * 32;      // sized for three planes
}

extern "C" int usip_mlp_split3_f32(const float* At, int lda, int M, int K, int tile_rows, void* planes, void* stream)
{
    if (!At || !planes || M < 1 || K < 1 || lda < M || (reinterpret_cast<uintptr_t>(planes) & 15u)) return USIP_EINVAL;
    if (tile_rows != 128 && tile_rows != 256) return USIP_EINVAL;
    const int bm = tile_rows;
    const int ksteps = (K + XBK - 1) / XBK, mts = (M + bm - 1) / bm;
    USIP_LAUNCH(split3_tiles_kernel, dim3((unsigned)(mts * ksteps)), dim3(2 * bm), 0, (hipStream_t)stream, At, lda, M, K,
                reinterpret_cast<uint4*>(planes), ksteps, bm, 3);
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}

// The two-plane fp16 image of the same operand for usip_mlp_gemm_x2h_f32 (same buffer size: usip_mlp_split3_bytes):
// per stage two planes, and behind the image one float = the power of two the operand was multiplied by.
extern "C" int usip_mlp_split2h_f32(const float* At, int lda, int M, int K, int tile_rows, void* planes, void* stream)
{
    if (!At || !planes || M < 1 || K < 1 || lda < M || (reinterpret_cast<uintptr_t>(planes) & 15u)) return USIP_EINVAL;
    if (tile_rows != 128 && tile_rows != 256) return USIP_EINVAL;
    const int bm = tile_rows;
    const int ksteps = (K + XBK - 1) / XBK, mts = (M + bm - 1) / bm;
    usip_split3_desc d{At, planes, lda, M, K, 0, bm, 2};
    USIP_LAUNCH(absmax_one_kernel, dim3(X2H_PARTS), dim3(256), 0, (hipStream_t)stream, d);
    USIP_LAUNCH_CHECK();
    USIP_LAUNCH(split3_tiles_kernel, dim3((unsigned)(mts * ksteps)), dim3(2 * bm), 0, (hipStream_t)stream, At, lda, M, K,
                reinterpret_cast<uint4*>(planes), ksteps, bm, 2);
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}

extern "C" int usip_mlp_split3_blocks(int M, int K, int tile_rows)
{
    if (M < 1 || K < 1 || (tile_rows != 128 && tile_rows != 256)) return 0;
    const int bm = tile_rows;
    return ((M + bm - 1) / bm) * ((K + XBK - 1) / XBK);
}

extern "C" int usip_mlp_split3_multi_f32(const usip_split3_desc* descs_device, int n, int total_blocks, void* stream)
{
    if (n < 0 || total_blocks < 0) return USIP_EINVAL;
    if (n == 0 || total_blocks == 0) return USIP_OK;
    if (!descs_device) return USIP_EINVAL;
    // operands asked for as two fp16 planes (reserved == 2) get their scale first; the others return at once
    USIP_LAUNCH(absmax_multi_kernel, dim3((unsigned)(n * X2H_PARTS)), dim3(256), 0, (hipStream_t)stream, descs_device, n);
    USIP_LAUNCH_CHECK();
    USIP_LAUNCH(split3_multi_kernel, dim3((unsigned)total_blocks), dim3(512), 0, (hipStream_t)stream, descs_device, n);
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}

template <int TM, int WN, int NPL = 3>
static int launch_x3p(const GemmArgs& a, const uint4* pl, int pro, hipStream_t st)
{
    constexpr int BM = 64 * TM, BN = 64 * WN;
    const int tpc = (a.P + BN - 1) / BN, nmt = (a.M + BM - 1) / BM;
    const long long total = (long long)a.nb * tpc * nmt;
    if (total > 0x7fffffffLL) return USIP_EINVAL;
    const int epi = a.stats ? EPI_STATS : EPI_NONE;
    dim3 grid((unsigned)total), block(128 * WN);
    // three-slot weight ring: every 128-row-tile launch (knob r5_forms bit 5 = 32: never).  Same box, alternating
    // (profiles/r05j_deep_ring_ab.txt): the six M-sized launches of the second stage and the head 258 -> 228 us together
    // (512 x 640 forward 56 -> 49, 256 x 512 pooled data gradient 44 -> 37, ...), the chip-filling 128 x 256 data gradient
    // 77 -> 70-75, the step 4.68-4.71 -> 4.65 ms
    const bool deep = TM == 2 && !(usip_tuning_value(USIP_TUNE_R5_FORMS) & 32);
#define USIP_X3P_CASE(P_, E_)                                                                 \
    if (pro == P_ && epi == E_) {                                                             \
        if constexpr (TM == 2) {                                                              \
            if (deep) {                                                                       \
                USIP_LAUNCH((gemm_x3p_kernel<P_, E_, TM, WN, NPL, 3>), grid, block, 0, st, a, pl); \
                USIP_LAUNCH_CHECK();                                                          \
                return USIP_OK;                                                               \
            }                                                                                 \
        }                                                                                     \
        USIP_LAUNCH((gemm_x3p_kernel<P_, E_, TM, WN, NPL>), grid, block, 0, st, a, pl);       \
        USIP_LAUNCH_CHECK();                                                                  \
        return USIP_OK;                                                                       \
    }
    if constexpr (NPL == 3) {
        USIP_X3P_CASE(PRO_NONE, EPI_STATS)
        USIP_X3P_CASE(PRO_NONE, EPI_NONE)
    }
    USIP_X3P_CASE(PRO_AFFINE_RELU, EPI_STATS)
    USIP_X3P_CASE(PRO_AFFINE_RELU, EPI_NONE)
    USIP_X3P_CASE(PRO_BN_BWD, EPI_NONE)
    USIP_X3P_CASE(PRO_BN_BWD_POOL, EPI_NONE)
#undef USIP_X3P_CASE
    return USIP_EINVAL;
}

// Y[b] = A . pro(X[b]) + bias (+ rowbias) with A given as the split image of usip_mlp_split3_f32 (same M, K).
// Contract of usip_mlp_gemm_f32 otherwise; K <= 640, P % 4 == 0 not required.
extern "C" int usip_mlp_x3p_tile_cols(int M, int P, int nb, int pro, int with_stats)
{
    if (usip_mlp_x3p_tile_rows(M, P, nb) == 128 || with_stats) return 128;
    const int knob = usip_tuning_value(USIP_TUNE_X3_GEMM_TILE);
    const long long wide_tiles = (long long)((M + 255) / 256) * ((P + 255) / 256) * nb;
    return (knob == 4 || (knob == 5 && pro >= 2 && wide_tiles >= 256)) ? 256 : 128;
}

extern "C" int usip_mlp_gemm_x3p_f32(const void* planes, const float* X, const float* X2, const float* coef, int pro,
                                     const float* bias, const float* rowbias, int rb_group, const float* pool_dp,
                                     const int32_t* pool_arg, int pool_group, float* Y, int y_rows, float* stats,
                                     int M, int K, int P, int nb, void* stream)
{
    if (M < 1 || K < 1 || K > 640 || P < 0 || nb < 0) return USIP_EINVAL;
    if ((pro == PRO_BN_BWD || pro == PRO_BN_BWD_POOL) && K > 512) return USIP_EINVAL;
    if ((long long)P * nb == 0) return USIP_OK;
    if (!planes || !Y || pro < 0 || pro > 3 || (reinterpret_cast<uintptr_t>(planes) & 15u)) return USIP_EINVAL;
    if (pro != PRO_BN_BWD_POOL && !X) return USIP_EINVAL;
    if (pro != PRO_NONE && !coef) return USIP_EINVAL;
    if ((pro == PRO_BN_BWD || pro == PRO_BN_BWD_POOL) && (!X2 || stats)) return USIP_EINVAL;
    if (pro == PRO_BN_BWD_POOL && (!pool_dp || !pool_arg || pool_group < 1 || P % pool_group != 0)) return USIP_EINVAL;
    if (rowbias && (rb_group < 1 || P % rb_group != 0)) return USIP_EINVAL;
    if (y_rows == 0) y_rows = M;
    if (y_rows < M) return USIP_EINVAL;
    GemmArgs a{nullptr, 0, X, X2, coef, bias, Y, stats, M, K, P, nb, rowbias, rb_group, pool_dp, pool_arg, pool_group,
               0, y_rows, (P % 4 == 0 && (reinterpret_cast<uintptr_t>(Y) & 15u) == 0) ? 1 : 0};
    hipStream_t st = (hipStream_t)stream;
    const uint4* pl = reinterpret_cast<const uint4*>(planes);
    if (usip_mlp_x3p_tile_rows(M, P, nb) == 128) return launch_x3p<2, 2>(a, pl, pro, st);
    // 256 x 256 tiles (8 waves, one workgroup per CU) are a measurement option (knob 4: every launch without
    // statistics -- their partial layout is per 128 positions; 5: data-gradient launches only).  r02r: alone on the
    // GPU the data-gradient launches gain 8 % (123 -> 113 us at 256 x 256 x 131072), inside the step the same choice
    // costs 1 % (6.50 vs 6.43 ms, same box, twice) -- half as many, longer workgroups drain worse behind the
    // neighbouring launches.  8-wave 256 x 128 and 128 x 256 variants (four waves per SIMD) measured equal to this one.
    if (usip_mlp_x3p_tile_cols(M, P, nb, pro, stats != nullptr) == 256) return launch_x3p<4, 4>(a, pl, pro, st);
    return launch_x3p<4, 2>(a, pl, pro, st);
}

// The same product from TWO fp16 planes per operand and THREE plane products (half the matrix work of f32x3, error at
// the same level): `planes` from usip_mlp_split2h_f32 (or a usip_split3_desc with reserved = 2).  Only for launches
// whose streamed operand has a known bound: pro 1 with coef = the [4][K] (scale, shift, mean, invstd) of a BatchNorm
// over exactly the nb * P samples of this launch (training mode), pro 2 / 3 with coef = the [5][K] array
// usip_bn_backward_reduce_f32 / usip_bn_pool_backward_reduce_f32 write with want_bound.  Everything else as
// usip_mlp_gemm_x3p_f32.
extern "C" int usip_mlp_gemm_x2h_f32(const void* planes, const float* X, const float* X2, const float* coef, int pro,
                                     const float* bias, const float* rowbias, int rb_group, const float* pool_dp,
                                     const int32_t* pool_arg, int pool_group, float* Y, int y_rows, float* stats,
                                     int M, int K, int P, int nb, void* stream)
{
    if (M < 1 || K < 1 || K > 640 || P < 0 || nb < 0) return USIP_EINVAL;
    if (pro != PRO_AFFINE_RELU && pro != PRO_BN_BWD && pro != PRO_BN_BWD_POOL) return USIP_EINVAL;
    if ((pro == PRO_BN_BWD || pro == PRO_BN_BWD_POOL) && K > 512) return USIP_EINVAL;
    if ((long long)P * nb == 0) return USIP_OK;
    if (!planes || !Y || !coef || (reinterpret_cast<uintptr_t>(planes) & 15u)) return USIP_EINVAL;
    if (pro != PRO_BN_BWD_POOL && !X) return USIP_EINVAL;
    if ((pro == PRO_BN_BWD || pro == PRO_BN_BWD_POOL) && (!X2 || stats)) return USIP_EINVAL;
    if (pro == PRO_BN_BWD_POOL && (!pool_dp || !pool_arg || pool_group < 1 || P % pool_group != 0)) return USIP_EINVAL;
    if (rowbias && (rb_group < 1 || P % rb_group != 0)) return USIP_EINVAL;
    if (y_rows == 0) y_rows = M;
    if (y_rows < M) return USIP_EINVAL;
    GemmArgs a{nullptr, 0, X, X2, coef, bias, Y, stats, M, K, P, nb, rowbias, rb_group, pool_dp, pool_arg, pool_group,
               0, y_rows, (P % 4 == 0 && (reinterpret_cast<uintptr_t>(Y) & 15u) == 0) ? 1 : 0};
    hipStream_t st = (hipStream_t)stream;
    const uint4* pl = reinterpret_cast<const uint4*>(planes);
    if (usip_mlp_x3p_tile_rows(M, P, nb) == 128) return launch_x3p<2, 2, 2>(a, pl, pro, st);
    if (usip_mlp_x3p_tile_cols(M, P, nb, pro, stats != nullptr) == 256) return launch_x3p<4, 4, 2>(a, pl, pro, st);
    // round 4: the streamed operand goes global -> registers -> MFMA (gemm_x2d.hip); knob x2_direct = 1 restores the
    // LDS-staged kernel for A/B runs
    if ((usip_tuning_value(USIP_TUNE_X2_DIRECT) & 15) != 1 && (long long)K * P * 4 < (1LL << 31))
        return launch_gemm_x2d(a, pl, pro, st);
    return launch_x3p<4, 2, 2>(a, pl, pro, st);
}

// Data-gradient launches of the direct kernel (gemm_x2d.hip) that also leave the BatchNorm-backward partial sums of the
// layer that PRODUCED the activation dX is the gradient of (pre-BN output red_y [nb][M][P], coefficients red_coef [4][M]:
// scale, shift, mean, invstd).  usip_mlp_gemm_x2d_red_tiles: the number of position tiles (= rows of the partial sums) of
// such a launch, 0 when the launch would not take that path (then call usip_mlp_gemm_x2h_f32 and the stand-alone
// reduction).  red_out: [2][tiles][M] sums, then [tiles * M / 256] maxima; red_gsum (optional, red_group 16 or 32):
// [2][nb * M][P / red_group].  Everything else as usip_mlp_gemm_x2h_f32 with pro 2 or 3.
extern "C" int usip_mlp_gemm_x2d_red_tiles(int M, int K, int P, int nb, int red_group)
{
    if (M < 256 || M % 256 || P < 128 || P % 128 || nb < 1 || K < 1 || K > 512) return 0;
    if (red_group != 0 && red_group != 16 && red_group != 32) return 0;
    if (usip_mlp_x3p_tile_rows(M, P, nb) != 256 || usip_mlp_x3p_tile_cols(M, P, nb, 2, 0) != 128) return 0;
    if ((usip_tuning_value(USIP_TUNE_X2_DIRECT) & 15) == 1 || (usip_tuning_value(USIP_TUNE_X2_DIRECT) & 15) == 8) return 0;
    if ((long long)K * P * 4 >= (1LL << 31) || (long long)M * P * 4 >= (1LL << 31)) return 0;
    return nb * (P / 128);
}

// Would usip_mlp_gemm_x2h_f32 hand a launch of this shape that reaches the direct kernel (256-row tiles, 128-position
// tiles: see the dispatcher above) to the round-6 form csrc/gemm_x2f.hip (one wave per SIMD, 256 x 256 tiles)?  Profiling aid:
// usip_amd/ops.py names the kernel a launch runs.  has_* = the pointer is given; rb_group 0 = no row bias; 16-B aligned
// operands assumed (the entry point checks the real pointers).
extern "C" int usip_mlp_gemm_x2f_used(int M, int K, int P, int nb, int pro, int has_stats, int has_bias, int rb_group,
                                      int pool_group, int y_rows)
{
    static const float dummy[4] __attribute__((aligned(16))) = {0.f, 0.f, 0.f, 0.f};
    GemmArgs a{nullptr, 0, dummy, dummy, dummy, has_bias ? dummy : nullptr, const_cast<float*>(dummy),
               has_stats ? const_cast<float*>(dummy) : nullptr, M, K, P, nb, rb_group ? dummy : nullptr, rb_group ? rb_group : 1,
               nullptr, nullptr, pool_group, 0, y_rows ? y_rows : M, (P % 4 == 0) ? 1 : 0};
    return gemm_x2f_takes(a, pro) ? 1 : 0;
}

extern "C" int usip_mlp_gemm_x2h_red_f32(const void* planes, const float* X, const float* X2, const float* coef, int pro,
                                         const float* pool_dp, const int32_t* pool_arg, int pool_group, float* Y,
                                         const float* red_y, const float* red_coef, float* red_out, float* red_gsum,
                                         int red_group, int M, int K, int P, int nb, void* stream)
{
    if (pro != PRO_BN_BWD && pro != PRO_BN_BWD_POOL) return USIP_EINVAL;
    if (!usip_mlp_gemm_x2d_red_tiles(M, K, P, nb, red_gsum ? red_group : 0)) return USIP_EINVAL;
    if (!planes || !Y || !coef || !X2 || !red_y || !red_coef || !red_out || (reinterpret_cast<uintptr_t>(planes) & 15u))
        return USIP_EINVAL;
    if ((reinterpret_cast<uintptr_t>(Y) | reinterpret_cast<uintptr_t>(red_y)) & 15u) return USIP_EINVAL;
    if (pro == PRO_BN_BWD && !X) return USIP_EINVAL;
    if (pro == PRO_BN_BWD_POOL && (!pool_dp || !pool_arg || pool_group < 1 || P % pool_group != 0)) return USIP_EINVAL;
    GemmArgs a{nullptr, 0, X, X2, coef, nullptr, Y, nullptr, M, K, P, nb, nullptr, 1, pool_dp, pool_arg, pool_group,
               0, M, 1, red_y, red_coef, red_out, red_gsum, red_gsum ? red_group : 0};
    return launch_gemm_x2d(a, reinterpret_cast<const uint4*>(planes), pro, (hipStream_t)stream);
}

// f32x2 for the 128-wide layers with register-resident weight fragments (gemm_x2r_kernel): M <= 128, K <= 128, a row
// bias only with pro 1 and rb_group a multiple of 32; `planes` = the usip_mlp_split2h_f32 image; stats: [2][M][usip_mlp_gemm_x2r_tiles(P, nb)] (one partial per workgroup).
// Otherwise the contract of usip_mlp_gemm_x2h_f32.
extern "C" int usip_mlp_gemm_x2r_tiles(int P, int nb)
{
    const long long total = (long long)nb * ((P + 63) / 64);
    return (int)(total < 512 ? total : 512);                   // = workgroups: one statistics partial per channel and workgroup
}

extern "C" int usip_mlp_gemm_x2r_f32(const void* planes, const float* X, const float* X2, const float* coef, int pro,
                                     const float* bias, const float* rowbias, int rb_group, const float* pool_dp,
                                     const int32_t* pool_arg, int pool_group, float* Y, int y_rows, float* stats, int M,
                                     int K, int P, int nb, void* stream)
{
    if (rowbias && (pro != PRO_AFFINE_RELU || rb_group < 32 || rb_group % 32 != 0 || P % rb_group != 0)) return USIP_EINVAL;
    if (M < 1 || M > 128 || K < 1 || K > 128 || P < 0 || nb < 0) return USIP_EINVAL;
    if (pro != PRO_AFFINE_RELU && pro != PRO_BN_BWD && pro != PRO_BN_BWD_POOL) return USIP_EINVAL;
    if ((long long)P * nb == 0) return USIP_OK;
    if (!planes || !Y || !coef || (reinterpret_cast<uintptr_t>(planes) & 15u)) return USIP_EINVAL;
    if (pro != PRO_BN_BWD_POOL && !X) return USIP_EINVAL;
    if ((pro == PRO_BN_BWD || pro == PRO_BN_BWD_POOL) && (!X2 || stats)) return USIP_EINVAL;
    if (pro == PRO_BN_BWD_POOL && (!pool_dp || !pool_arg || pool_group < 1 || P % pool_group != 0)) return USIP_EINVAL;
    if ((long long)K * P >= (1LL << 30)) return USIP_EINVAL;
    if (y_rows == 0) y_rows = M;
    if (y_rows < M) return USIP_EINVAL;
    GemmArgs a{nullptr, 0, X, X2, coef, bias, Y, stats, M, K, P, nb, rowbias, rowbias ? rb_group : 1, pool_dp, pool_arg,
               pool_group, 0, y_rows, (P % 4 == 0 && (reinterpret_cast<uintptr_t>(Y) & 15u) == 0) ? 1 : 0};
    const long long total = (long long)nb * ((P + 63) / 64);
    const unsigned grid = (unsigned)(total < 512 ? total : 512);             // two workgroups per CU, persistent
    hipStream_t st = (hipStream_t)stream;
    const uint4* pl = reinterpret_cast<const uint4*>(planes);
    if (pro == PRO_AFFINE_RELU && rowbias) {
        if (stats) USIP_LAUNCH((gemm_x2r_kernel<PRO_AFFINE_RELU, EPI_STATS, true>), dim3(grid), dim3(256), 0, st, a, pl);
        else USIP_LAUNCH((gemm_x2r_kernel<PRO_AFFINE_RELU, EPI_NONE, true>), dim3(grid), dim3(256), 0, st, a, pl);
    } else if (pro == PRO_AFFINE_RELU) {
        if (stats) USIP_LAUNCH((gemm_x2r_kernel<PRO_AFFINE_RELU, EPI_STATS>), dim3(grid), dim3(256), 0, st, a, pl);
        else USIP_LAUNCH((gemm_x2r_kernel<PRO_AFFINE_RELU, EPI_NONE>), dim3(grid), dim3(256), 0, st, a, pl);
    } else if (pro == PRO_BN_BWD) {
        USIP_LAUNCH((gemm_x2r_kernel<PRO_BN_BWD, EPI_NONE>), dim3(grid), dim3(256), 0, st, a, pl);
    } else {
        USIP_LAUNCH((gemm_x2r_kernel<PRO_BN_BWD_POOL, EPI_NONE>), dim3(grid), dim3(256), 0, st, a, pl);
    }
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}

// ---- weight gradient, 256 x 256 tiles -------------------------------------------------------------------------
namespace usip_mlp {

// Slicing of the 256 x 256-tile weight gradient: ~256 workgroups (one per CU, one round), position
// segments that are multiples of 32 and at least 512 long (a workgroup writes a 256 KiB partial tile; shorter
// segments would make that the dominant traffic).
void wgrad_x3_plan(int M, int N, int P, int nb, int* seglen, int* segs, int* tiles)
{
    *tiles = ((M + 255) / 256) * ((N + 255) / 256);
    // ONE round of workgroups (256, one per CU) since r03: two rounds (512) wrote and re-read twice the partial tiles
    // (134 MB for the 512 x 512 gradient) for no better balance -- 5.64 -> 5.60 ms per step, same box, three runs each
    // (measurement knob x3_wgrad_tile = 2 restores the two rounds)
    long long want = (usip_tuning_value(USIP_TUNE_X3_WGRAD_TILE) == 2 ? 512 : 256) / (*tiles);
    if (want < 1) want = 1;
    long long per_cloud = (want + nb - 1) / nb;
    if (per_cloud < 1) per_cloud = 1;
    long long sl = (P + per_cloud - 1) / per_cloud;
    sl = ((sl + 31) / 32) * 32;
    // >= 512 positions per segment (a workgroup writes a 256 KiB partial tile) -- unless that leaves the chip half
    // empty: the head's products (512 positions per cloud, 16 clouds x 6 tiles = 96 workgroups) take 256 or 128
    // (knob x3_wgrad_tile = 4: always 512)
    long long floor_sl = 512;
    if (usip_tuning_value(USIP_TUNE_X3_WGRAD_TILE) != 4)
        while (floor_sl > 128 && (long long)(*tiles) * nb * ((P + floor_sl - 1) / floor_sl) < 192) floor_sl /= 2;
    if (sl < floor_sl) sl = floor_sl;
    *seglen = (int)sl;
    *segs = (int)((P + sl - 1) / sl);
}

int launch_wgrad_x2h_256(const WgradArgs& a, int pro, unsigned blocks, hipStream_t st)
{
    dim3 grid(blocks), block(512);
    // round 5: full-line loads (wgrad_x2l_kernel; same partial tiles bit for bit).  Its row offsets are 32-bit element
    // offsets inside a cloud; knob r5_forms bit 0 keeps round 4's kernel for A/B runs
    const bool line = !(usip_tuning_value(USIP_TUNE_R5_FORMS) & 1) &&
                      (long long)(a.M > a.N ? a.M : a.N) * a.P < (1LL << 30);
    if (line && pro == PRO_BN_BWD) { USIP_LAUNCH((wgrad_x2l_kernel<PRO_BN_BWD>), grid, block, 0, st, a); USIP_LAUNCH_CHECK(); return USIP_OK; }
    if (line && pro == PRO_BN_BWD_POOL) { USIP_LAUNCH((wgrad_x2l_kernel<PRO_BN_BWD_POOL>), grid, block, 0, st, a); USIP_LAUNCH_CHECK(); return USIP_OK; }
    if (pro == PRO_BN_BWD) USIP_LAUNCH((wgrad_x3_kernel<PRO_BN_BWD, true, 2>), grid, block, 0, st, a);
    else if (pro == PRO_BN_BWD_POOL) USIP_LAUNCH((wgrad_x3_kernel<PRO_BN_BWD_POOL, true, 2>), grid, block, 0, st, a);
    else return USIP_EINVAL;
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}

int launch_wgrad_x3_256(const WgradArgs& a, int pro, bool xpro, unsigned blocks, hipStream_t st)
{
    dim3 grid(blocks), block(512);
#define USIP_X3W_CASE(P_, X_)                                                             \
    if (pro == P_ && xpro == X_) {                                                        \
        USIP_LAUNCH((wgrad_x3_kernel<P_, X_>), grid, block, 0, st, a);                    \
        USIP_LAUNCH_CHECK();                                                              \
        return USIP_OK;                                                                   \
    }
    USIP_X3W_CASE(PRO_NONE, false) USIP_X3W_CASE(PRO_NONE, true)
    USIP_X3W_CASE(PRO_BN_BWD, false) USIP_X3W_CASE(PRO_BN_BWD, true)
    USIP_X3W_CASE(PRO_BN_BWD_POOL, false) USIP_X3W_CASE(PRO_BN_BWD_POOL, true)
#undef USIP_X3W_CASE
    return USIP_EINVAL;
}

}  // namespace usip_mlp
