// usip_amd/csrc/gemm_x2d.hip -- the f32x2 forward / data-gradient GEMM of the 256..640-wide shared-MLP layers
// (models/layers.py:208-216, :293-303, :401-440) with the STREAMED operand going global memory -> registers -> MFMA,
// never through LDS ("direct").  Round 4; replaces gemm_x3p_kernel<.., 4, 2, 2> of shared_mlp_x3.hip on that path.
//
// Why.  The round-3 kernel staged the streamed operand through LDS: load -> BN/ReLU prologue -> fp16 hi/lo split ->
// ds_write -> barrier -> ds_read -> MFMA.  Its counters (profiles/r03_pmc_x2_gemm.txt) showed a wave parked 41 % of its
// cycles (s_waitcnt / s_barrier) with the matrix pipe 38 % busy: every 16-k stage put an LDS write -> barrier -> read
// round trip and a drain of the global loads in front of the next 24 MFMAs, and spent 2.3 VALU + 3 SALU instructions
// per MFMA on addresses, clamps, conversions back from fp16 and LDS coefficient lookups.
//
// What.  Tile 256 channels x 128 positions, four waves, two workgroups per CU -- but each wave owns 32 POSITIONS of
// ALL 256 channels (8 accumulator tiles of 32 x 32).  The MFMA operand layout of the streamed side (lane = position
// l & 31, eight consecutive k at 8 (l >> 5)) is then exactly what a lane can load for itself: 8 dword loads per
// stage, rows (k) at a scalar offset, positions across lanes (two 128-B segments per instruction).  Prologue and split
// run on those registers and the result IS the MFMA operand.  Consequences:
//   * no LDS write / barrier / read for the streamed operand; the only shared data is the weight image, which arrives
//     by LDS-DMA two stages ahead into a four-slot ring, so the per-stage barrier never has a memory wait behind it;
//   * the prologue of stage kt+1 runs while stage kt's MFMAs issue (the planes of the NEXT stage are prepared one
//     stage ahead into a second register set; the loop is unrolled by two so that no copies are needed);
//   * instruction diet: split = v_cvt_pk_f16_f32 + v_fma_mixlo/hi_f16 (3 instructions per PAIR instead of 6; the mix
//     form subtracts the fp16 high part from the fp32 value and rounds the difference to fp16 in one instruction),
//     coefficients as 16-B broadcast LDS reads ([k][c0, c1(, c2, c3)] interleaved), row offsets as scalar adds,
//     weight DMA through a buffer descriptor with scalar offsets (no 64-bit vector adds), no clamps unless K % 16 != 0.
// Arithmetic, scales, plane products and their order are those of shared_mlp_x3.hip (two fp16 planes per operand,
// three products, smallest first); results differ from it only by fp32 summation order in the statistics.
#include "mlp_common.h"
#include "split_common.h"
#include <type_traits>

using namespace usip_mlp;

namespace {

constexpr int DBM = 256, DBN = 128, DNT = 256, DSLOTS = 4;
constexpr int DPL = DBM * 32;                                  // bytes of one plane of one 16-k stage of the weights
constexpr int DSTAGE = 2 * DPL;                                // hi + lo

// byte offset of (row, 16-B half) inside a [rows][16 fp16] plane (the image usip_mlp_split2h_f32 writes)
__device__ __forceinline__ int d_lds_off(int row, int half) { return row * 32 + ((half ^ (row >> 3)) & 1) * 16; }

// two fp32 -> packed fp16 high parts and packed fp16 low parts (x = hi + lo up to 2^-22 |x|), 3 VALU instructions
__device__ __forceinline__ void split_pair_mix(float x, float y, unsigned& hi, unsigned& lo)
{
    const f32x2 v = {x, y};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));          // v_cvt_pk_f16_f32, RNE
    // lo.lo16 = f16(x - f32(hi.lo16)), lo.hi16 = f16(y - f32(hi.hi16)): the differences are exact in fp32
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(lo) : "v"(x), "v"(hi));
    asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(y), "v"(hi));
}

// Output epilogue of the direct kernel.  A wave holds 32 positions x 256 channels: lane = channel (l & 31) of each of the
// eight 32-channel tiles, positions 8g + 4 (l >> 5) + e.  Stored straight from that layout (mlp_common.h's epilogue) a
// store instruction writes 32 rows x 32 B -- measured: the epilogue of the 512 x 512 layer cost 70 of the launch's 240
// us (profiles/r04h_gemm_x2d_time_split.txt).  Here every channel tile goes through a 32 x 32 transposition in LDS (wave
// private, no barrier: a wave's DS instructions execute in order), after which a store instruction writes 8 rows x 128
// B: whole cache lines.  Bias, row bias, statistics and their summation order are those of gemm_epilogue<.., 8, 1>.
constexpr int TRS = 36;                                        // floats per transposition row (32 + 4: conflict-free b128 writes)
constexpr int EPI_SCRATCH_FLOATS = 4 * 32 * TRS + 2 * 4 * DBM; // transposition areas + statistics exchange

template <int EPI>
__device__ __forceinline__ void epilogue_x2d(const GemmArgs& a, f32x16 (&acc)[8][1], float out_scale, float* scratch,
                                             int scratch_floats, int b, int m0, int p0, int tn, int tpc)
{
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* tr = scratch + wave * (32 * TRS);
    float* red = scratch + 4 * 32 * TRS;                       // [2][4 waves][256 channels]
    float* rbs = red + 2 * 4 * DBM;
    const int pw = p0 + wave * 32;
    const int ngrp = a.rowbias ? a.P / a.rb_group : 0;
    const int g0 = a.rowbias ? p0 / a.rb_group : 0;
    const int G = a.rowbias ? (min(p0 + DBN, a.P) - 1) / a.rb_group - g0 + 1 : 0;
    const bool rb_lds = a.rowbias && DBM * G <= scratch_floats - EPI_SCRATCH_FLOATS;
    if (rb_lds) {
        for (int e = tid; e < DBM * G; e += DNT) {
            const int rl = e / G, g = e % G;
            rbs[e] = (m0 + rl < a.M) ? a.rowbias[((long long)b * a.M + m0 + rl) * ngrp + g0 + g] : 0.f;
        }
        __syncthreads();
    }
    // compute role: channel c of tile i, positions pw + 8g + 4 half + e; store role: row rr + 8k of tile i, positions pst..+3
    const int rr = lane >> 3, cc = lane & 7;
    const int pst = pw + 4 * cc;
    const bool st_full = pst + 3 < a.P;
    float* const ydst = a.Y + (long long)b * a.y_rows * a.P + (long long)(m0 + rr) * a.P + pst;
    float* const trw = tr + c * TRS + 4 * half;
    const float* const trr = tr + rr * TRS + 4 * cc;
    bool pv[4][4];                                             // position of (g, e) inside the cloud
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) pv[g][e] = pw + 8 * g + 4 * half + e < a.P;

    auto tiles = [&](auto rb_tag) {
        constexpr bool RB = decltype(rb_tag)::value;
        const bool run_one_group = RB && (a.rb_group % 4 == 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row_l = i * 32 + c, row = m0 + row_l, rowc = min(row, a.M - 1);
            const bool rok = row < a.M;
            const float bv = a.bias ? a.bias[rowc] : 0.0f;
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int pb = pw + 8 * g + 4 * half;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(acc[i][0][4 * g + e], out_scale, bv);   // out_scale = 2^n: exact
                if (RB) {
                    if (run_one_group) {
                        const int grp = min(pb, a.P - 1) / a.rb_group;
                        const float rb = rb_lds ? rbs[row_l * G + grp - g0] : a.rowbias[((long long)b * a.M + rowc) * ngrp + grp];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += rb;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int grp = min(pb + e, a.P - 1) / a.rb_group;
                            v[e] += rb_lds ? rbs[row_l * G + grp - g0] : a.rowbias[((long long)b * a.M + rowc) * ngrp + grp];
                        }
                    }
                }
                if (EPI == EPI_STATS) {
                    // (adding 0 for rows / positions outside the tensor leaves the sums of the old epilogue bit for bit)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float u = (rok && pv[g][e]) ? v[e] : 0.0f;
                        s += u;
                        q = __builtin_fmaf(u, u, q);
                    }
                }
                *reinterpret_cast<float4*>(trw + 8 * g) = make_float4(v[0], v[1], v[2], v[3]);
            }
            if (EPI != EPI_NONE) {
                s += __shfl_xor(s, 32);
                q += __shfl_xor(q, 32);
                if (half == 0) { red[wave * DBM + row_l] = s; red[4 * DBM + wave * DBM + row_l] = q; }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // see epilogue_x2d_fast
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float4 w = *reinterpret_cast<const float4*>(trr + 8 * k * TRS);
                float* dst = ydst + (long long)(i * 32 + 8 * k) * a.P;
                if (m0 + i * 32 + rr + 8 * k < a.M) {
                    if (a.y_vec && st_full) {
                        *reinterpret_cast<float4*>(dst) = w;
                    } else {
                        if (pst + 0 < a.P) dst[0] = w.x;
                        if (pst + 1 < a.P) dst[1] = w.y;
                        if (pst + 2 < a.P) dst[2] = w.z;
                        if (pst + 3 < a.P) dst[3] = w.w;
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);                 // one channel tile at a time (register pressure: 128 VGPRs)
        }
    };
    if (a.rowbias) tiles(std::true_type{}); else tiles(std::false_type{});
    if (EPI != EPI_NONE) {
        __syncthreads();
        if (m0 + tid < a.M) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { s += red[w * DBM + tid]; q += red[4 * DBM + w * DBM + tid]; }
            const long long ntn = (long long)a.nb * tpc;
            a.stats[(long long)(m0 + tid) * ntn + tn] = s;
            a.stats[ntn * a.M + (long long)(m0 + tid) * ntn + tn] = q;
        }
    }
}

// The same for a tile that lies entirely inside the tensor (every tile of the step's layers): no row or position
// predicates, 32-bit buffer addressing with the row as a scalar offset, the row bias (one value per run of four
// positions: rb_group % 4 == 0) straight from L2.  ~100 instructions per channel tile instead of the ~500 the general
// form compiles to -- its instruction stream alone took ~10 us per tile.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// RED (data-gradient launches, EPI_NONE, no row bias): the tile just computed is dX = the gradient of the producing layer's
// activated output; while it is in registers in the store role (8 rows x 128 B per instruction) the same lanes read the
// producing layer's pre-BN tile (red_y, the same coalesced pattern) and take its BatchNorm-backward sums -- what the
// stand-alone usip_bn_backward_reduce_f32 pass re-read (dX, Y) from HBM for.  Sums over a row's 32 positions: three DPP
// steps over its 8 lanes, fixed order; per tile and channel one partial (four waves added in order); per-neighbourhood
// sums (red_group = 16 or 32 positions) fall out of the same steps.
template <int EPI, bool RB, bool RED = false>
__device__ __forceinline__ void epilogue_x2d_fast(const GemmArgs& a, f32x16 (&acc)[8][1], float out_scale, float* scratch,
                                                  int b, int m0, int p0, int tn, int tpc)
{
    static_assert(!RED || (EPI == EPI_NONE && !RB), "RED: data-gradient launches only");
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* tr = scratch + wave * (32 * TRS);
    float* red = scratch + 4 * 32 * TRS;
    const int pw = p0 + wave * 32;
    const int rr = lane >> 3, cc = lane & 7;
    const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.Y + (long long)b * a.y_rows * a.P), 0, (unsigned)a.y_rows * (unsigned)a.P * 4u, 0x00020000);
    const int st_voff = ((m0 + rr) * a.P + pw + 4 * cc) * 4;
    float* const trw = tr + c * TRS + 4 * half;
    const float* const trr = tr + rr * TRS + 4 * cc;
    float bv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) bv[i] = a.bias ? a.bias[m0 + i * 32 + c] : 0.0f;
    const int ngrp = RB ? a.P / a.rb_group : 0;
    int grp[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) grp[g] = RB ? (pw + 8 * g + 4 * half) / a.rb_group : 0;
    const float* rbp = RB ? a.rowbias + ((long long)b * a.M + m0 + c) * ngrp : nullptr;
    // RED state
    float* cfr = scratch + EPI_SCRATCH_FLOATS;                  // [256 rows][scale, shift, mean, invstd]
    const __amdgpu_buffer_rsrc_t rYp = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(RED ? a.red_y + (long long)b * a.M * a.P : a.Y), 0, RED ? (unsigned)a.M * (unsigned)a.P * 4u : 4u, 0x00020000);
    const int yp_voff = RED ? ((m0 + rr) * a.P + pw + 4 * cc) * 4 : 0;
    const int rgrp = RED ? a.red_group : 0, rngrp = (RED && rgrp) ? a.P / rgrp : 0;
    float mx = 0.f;
    u32x4 ynext = {0u, 0u, 0u, 0u};                             // the producing layer's tile, one 8-row slab ahead of its use
    if (RED) {
        ynext = __builtin_amdgcn_raw_buffer_load_b128(rYp, yp_voff, 0, 0);
        for (int e = tid; e < DBM * 4; e += DNT) cfr[(e & 255) * 4 + (e >> 8)] = a.red_coef[(long long)(e >> 8) * a.M + m0 + (e & 255)];
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float rb[4] = {0.f, 0.f, 0.f, 0.f};
        if (RB) {
#pragma unroll
            for (int g = 0; g < 4; ++g) rb[g] = rbp[(long long)i * 32 * ngrp + grp[g]];
        }
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = __builtin_fmaf(acc[i][0][4 * g + e], out_scale, bv[i]);   // out_scale = 2^n: exact
                if (RB) v[e] += rb[g];
                if (EPI == EPI_STATS) { s += v[e]; q = __builtin_fmaf(v[e], v[e], q); }
            }
            *reinterpret_cast<float4*>(trw + 8 * g) = make_float4(v[0], v[1], v[2], v[3]);
        }
        if (EPI != EPI_NONE) {
            s += __shfl_xor(s, 32);
            q += __shfl_xor(q, 32);
            if (half == 0) { red[wave * DBM + i * 32 + c] = s; red[4 * DBM + wave * DBM + i * 32 + c] = q; }
        }
        // The transposition reads what OTHER lanes of this wave just wrote.  With the reads issued right behind the
        // writes (this lean form) a few lanes read the previous tile's values -- 16 wrong elements in 15 % of the
        // workgroups, r04p; the general form with hundreds of instructions in between never did.  Wait for the writes.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float4 w = *reinterpret_cast<const float4*>(trr + 8 * k * TRS);
            const u32x4 d = {__float_as_uint(w.x), __float_as_uint(w.y), __float_as_uint(w.z), __float_as_uint(w.w)};
            __builtin_amdgcn_raw_buffer_store_b128(d, rY, st_voff, (i * 32 + 8 * k) * a.P * 4, st_aux<ST_X2D>());
            if (RED) {
                const int row_l = i * 32 + rr + 8 * k;
                const float4 c4 = *reinterpret_cast<const float4*>(cfr + row_l * 4);
                const float wv[4] = {w.x, w.y, w.z, w.w};
                const u32x4 yq = ynext;                        // requested one (i, k) item ago
                if (i * 4 + k + 1 < 32)
                    ynext = __builtin_amdgcn_raw_buffer_load_b128(rYp, yp_voff, ((i * 4 + k + 1) / 4 * 32 + 8 * ((i * 4 + k + 1) % 4)) * a.P * 4, 0);
                const float yv[4] = {__uint_as_float(yq.x), __uint_as_float(yq.y), __uint_as_float(yq.z), __uint_as_float(yq.w)};
                float gd = 0.f, gy = 0.f, s2 = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float dv = (__builtin_fmaf(yv[e], c4.x, c4.y) > 0.f) ? wv[e] : 0.f;
                    gd += dv;
                    gy += yv[e];
                    s2 = __builtin_fmaf(dv, (yv[e] - c4.z) * c4.w, s2);
                    mx = fmaxf(mx, fabsf(dv));
                }
                // lanes 8 rr .. 8 rr + 7 hold the row's 32 positions: quad sums (16 positions), then the other quad
                gd += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, gd), 0xB1, 0xF, 0xF, false));
                gd += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, gd), 0x4E, 0xF, 0xF, false));
                s2 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s2), 0xB1, 0xF, 0xF, false));
                s2 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s2), 0x4E, 0xF, 0xF, false));
                if (rgrp) {
                    gy += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, gy), 0xB1, 0xF, 0xF, false));
                    gy += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, gy), 0x4E, 0xF, 0xF, false));
                }
                const float gd8 = gd + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, gd), 0x141, 0xF, 0xF, false));
                const float s28 = s2 + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s2), 0x141, 0xF, 0xF, false));
                if (rgrp == 16) {
                    if ((cc & 3) == 0) {
                        const long long rowid = (long long)b * a.M + m0 + row_l;
                        a.red_gsum[rowid * rngrp + (pw + 4 * cc) / 16] = gd;
                        a.red_gsum[((long long)a.nb * a.M + rowid) * rngrp + (pw + 4 * cc) / 16] = gy;
                    }
                } else if (rgrp == 32) {
                    const float gy8 = gy + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, gy), 0x141, 0xF, 0xF, false));
                    if (cc == 0) {
                        const long long rowid = (long long)b * a.M + m0 + row_l;
                        a.red_gsum[rowid * rngrp + pw / 32] = gd8;
                        a.red_gsum[((long long)a.nb * a.M + rowid) * rngrp + pw / 32] = gy8;
                    }
                }
                if (cc == 0) { red[wave * DBM + row_l] = gd8; red[4 * DBM + wave * DBM + row_l] = s28; }
            }
        }
        // gfx950 / ROCm 7.2: a buffer_store_dwordx4 with an SGPR soffset whose data registers the NEXT instruction
        // overwrites (here: the v_fma of the next channel tile re-using the tuple) stored the new value in the first
        // dword of lanes 12-15 of every 16 -- 16 wrong elements in 15 % of the workgroups, timing dependent (r04p-r04s;
        // hipcc inserts wait states only for the immediate-soffset form).  Eight wait states behind the last store.
        asm volatile("s_nop 7" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
    if (EPI != EPI_NONE) {
        __syncthreads();
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) { s += red[w * DBM + tid]; q += red[4 * DBM + w * DBM + tid]; }
        const long long ntn = (long long)a.nb * tpc;
        a.stats[(long long)(m0 + tid) * ntn + tn] = s;
        a.stats[ntn * a.M + (long long)(m0 + tid) * ntn + tn] = q;
    }
    if (RED) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        float* wmx = cfr;                                       // (the coefficient table is read; reuse behind the barrier)
        __syncthreads();
        if (lane == 0) wmx[wave] = mx;
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) { s += red[w * DBM + tid]; q += red[4 * DBM + w * DBM + tid]; }
        const long long ntn = (long long)a.nb * tpc;
        a.red_out[(long long)tn * a.M + m0 + tid] = s;          // [2][tiles][M]
        a.red_out[(ntn + tn) * a.M + m0 + tid] = q;
        __syncthreads();
        if (tid == 0) a.red_out[2 * ntn * a.M + (long long)tn * (a.M / DBM) + m0 / DBM] = fmaxf(fmaxf(wmx[0], wmx[1]), fmaxf(wmx[2], wmx[3]));
    }
}

// DIRECT (data-gradient launches: no statistics, no bias, every tile inside the tensor): the MFMA operands are NOT swapped
// -- weights as A, the streamed operand as B -- so a lane holds ONE position and 16 channels per accumulator tile, and
// every accumulator register is a store of two full 128-B lines (rows r and r + 4, 32 positions each) straight from the
// registers: no LDS round trip, no barrier behind the tile.  (The swapped form exists for the statistics: with a lane =
// a channel they are sums over registers.)
__device__ __forceinline__ void epilogue_x2d_direct(const GemmArgs& a, f32x16 (&acc)[8][1], float out_scale, int b, int m0, int p0)
{
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.Y + (long long)b * a.y_rows * a.P), 0, (unsigned)a.y_rows * (unsigned)a.P * 4u, 0x00020000);
    const int voff = ((m0 + 4 * half) * a.P + p0 + wave * 32 + c) * 4;
    const int rowb = a.P * 4;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[i][0][4 * g + e] * out_scale;          // 2^n: exact
#pragma unroll
            for (int e = 0; e < 4; ++e)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[e]), rY, voff, (i * 32 + 8 * g + e) * rowb, st_aux<ST_X2D>());
            // the stores read their data registers late (see epilogue_x2d_fast): nothing may overwrite v[] at once
            asm volatile("s_nop 7" ::: "memory");
        }
    }
}

template <int PRO, int EPI, bool KTAIL, int DEPTH, bool REDK = false, bool GEN = false, bool DIRECT = false>
__global__ __launch_bounds__(DNT) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_x2d_kernel(
    const GemmArgs a, const uint4* __restrict__ planes)
{
    constexpr bool POOL = (PRO == PRO_BN_BWD_POOL);
    constexpr bool TWO = (PRO == PRO_BN_BWD) || POOL;
    constexpr int NC = TWO ? 4 : 2;                            // prologue coefficients per input channel
    constexpr int KMAX = TWO ? 512 : 640;
    constexpr int KPAD = KMAX + 16;
    constexpr int RING = DSLOTS * DSTAGE;
    __shared__ __attribute__((aligned(16))) unsigned char smem[RING + NC * KPAD * 4];
    float* cf = reinterpret_cast<float*>(smem + RING);         // [k][NC], zero beyond K

    const int tid = threadIdx.x, lane = tid & 63, c = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // same logical tile order and XCD remap as the other GEMMs of the library
    const int tpc = (a.P + DBN - 1) / DBN, nmt = (a.M + DBM - 1) / DBM;
    const int total = a.nb * tpc * nmt;
    const int nk = (a.K + XBK - 1) / XBK;
    // measurement aid (tools/x2_knob_bench.py, knob x2_direct = 16 / 32 / 48): bit 0 = run only two stages of the main
    // loop, bit 1 = no epilogue -- wrong results, used to split a launch's time into prologue / loop / epilogue
    const int dbg = a.a_trans;
    const int nk_run = (dbg & 1) ? min(nk, 2) : nk;

    // operand scales (see gemm_x3p_kernel): weights carry theirs behind the image, the streamed operand's comes from a
    // rigorous bound of what the prologue can produce
    float xs, out_scale;
    {
        float* redm = reinterpret_cast<float*>(smem);
        float bnd = 0.f;
        if (PRO == PRO_AFFINE_RELU) {
            const float rn = sqrtf((float)a.nb * (float)a.P);
            for (int k = tid; k < a.K; k += DNT) {
                const float c0 = a.coef[k], c1 = a.coef[a.K + k], mu = a.coef[2 * a.K + k], is = a.coef[3 * a.K + k];
                bnd = fmaxf(bnd, fabsf(c0) / is * rn + fabsf(__builtin_fmaf(mu, c0, c1)));
            }
        } else {
            for (int i = tid; i < (a.K + 63) / 64; i += DNT) bnd = fmaxf(bnd, a.coef[4 * a.K + i]);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) bnd = fmaxf(bnd, __shfl_xor(bnd, off));
        if (lane == 0) redm[wave] = bnd;
        __syncthreads();
        bnd = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]));
        xs = pow2_scale(bnd, X2H_TOP);
        const float ws = __uint_as_float(planes[(long long)nmt * nk * (DSTAGE / 16)].x);
        out_scale = 1.0f / (xs * ws);
        for (int i = tid; i < nk * XBK * NC; i += DNT) {
            const int k = i / NC, j = i % NC;
            cf[i] = (k < a.K) ? a.coef[j * a.K + k] * xs : 0.0f;
        }
        __syncthreads();                                       // redm is read; the ring may be written from here on
    }

    // Persistent workgroups (two per CU): the scale and the coefficient table above are per launch, not per tile -- with
    // one tile per workgroup they and the first loads' latency were 30 of the 240 us of the 512 x 512 layer.  Virtual
    // block v -> logical tile as before (blocks of one XCD, v & 7, take consecutive tiles, so the two row tiles of a
    // position tile meet in one L2); gridDim.x is a multiple of 8 whenever it is smaller than `total`.
    for (int v = blockIdx.x; v < total; v += gridDim.x) {
    int L = v;
    if ((total & 7) == 0) L = (v & 7) * (total >> 3) + (v >> 3);
    const int mt = L % nmt, tn = L / nmt;
    const int b = tn / tpc, pt = tn % tpc;
    const int m0 = mt * DBM, p0 = pt * DBN;

    // Buffer descriptors as plain SGPR quads (base, stride 0, bytes, raw-buffer flags): every memory instruction of the
    // main loop is inline asm (see dma_half / load_x1), which takes them as "s" operands.
    typedef int v4i32 __attribute__((ext_vector_type(4)));
    auto make_rsrc = [](const void* base, unsigned bytes) {
        const unsigned long long p = (unsigned long long)reinterpret_cast<uintptr_t>(base);
        return v4i32{(int)__builtin_amdgcn_readfirstlane((unsigned)p),
                     (int)__builtin_amdgcn_readfirstlane((unsigned)(p >> 32) & 0xffffu),
                     (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000};
    };
    // weight image of this row tile through a buffer descriptor: a stage is 16 KiB = 4 x (256 lanes x 16 B), wave w
    // copies chunk rows [w*64, w*64+64) of each quarter: voffset = tid * 16, scalar offset = stage and quarter
    const v4i32 rAv = make_rsrc(planes + (long long)mt * nk * (DSTAGE / 16), (unsigned)nk * DSTAGE);
    const int a_voff = tid * 16;

    // streamed operand: lane = (position c of the wave's 32, k-half h)
    const unsigned pc = (unsigned)min(p0 + wave * 32 + c, a.P - 1);
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
    const int rs = a.P * 4, rsg = pgrp * 4;                    // row strides in bytes
    const int xoff = (int)(pc * 4u), goff = POOL ? (int)((pc / (unsigned)a.pool_group) * 4u) : 0;
    const int xvoff = xoff + h * 8 * rs, gvoff = goff + h * 8 * rsg;
    const int xkin = POOL ? (int)(pc % (unsigned)a.pool_group) : 0;

    // DEPTH register sets of raw operand values: the loads run DEPTH stages ahead of their conversion.  One stage of one
    // wave is 8 x 256 B; a CU (8 waves) keeps DEPTH x 16 KiB in flight, and tools/l2_cu_bw.hip shows that HBM needs
    // ~32 KiB per CU in flight before it delivers its 24 GB/s per CU (16 KiB: 18).
    struct XSet { float rx[8]; float ry[TWO ? 8 : 1]; int rarg[POOL ? 8 : 1]; };
    XSet xs0, xs1;
    auto load_x1 = [&](int kt, XSet& S, int i) {               // element i (row kt*16 + 8h + i) of stage kt
        const int kb = kt * XBK;
        int vo = xvoff, so = (kb + i) * rs, vg = gvoff, sg = (kb + i) * rsg;
        if (KTAIL) {                                           // rows beyond K: clamp per lane (finite data for zero weights)
            const int k = min(kb + h * 8 + i, a.K - 1);
            vo = xoff + k * rs; so = 0; vg = goff + k * rsg; sg = 0;
        }
        // Plain (compiler-visible) loads: hipcc places the counted waits in front of their uses.  Round 4 tried them as
        // inline asm with hand-counted waits -- faster by nothing, and unsafe: an asm load's destination is an ordinary
        // value for the register allocator, which may copy it (to satisfy a tied asm operand, a loop phi) BEFORE the
        // hand-written wait: one stage of garbage per tile in one of twelve instantiations (r04k).  The weight DMA stays
        // asm (it writes no register); the waits hipcc computes without knowing about it are only ever too strict.
        if (POOL) {
            S.rx[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rPd, vg, sg, 0));
            S.rarg[POOL ? i : 0] = (int)__builtin_amdgcn_raw_buffer_load_b32(rPa, vg, sg, 0);
        } else {
            S.rx[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rX, vo, so, 0));
        }
        if (TWO) S.ry[TWO ? i : 0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rX2, vo, so, 0));
    };
    auto load_x = [&](int kt, XSet& S) {
#pragma unroll
        for (int i = 0; i < 8; ++i) load_x1(kt, S, i);
    };
    // prologue of element i / split of pair j of the stage in S -> packed fp16 planes (the MFMA operand of the lane);
    // cq: the coefficients of the stage for this half-wave, TWO: one float4 (c0..c3) per k, else one per PAIR of k
    const float4* cfl = reinterpret_cast<const float4*>(cf) + h * (8 * NC / 4);
    float4 cq[TWO ? 8 : 4];
    float cv[8];
    auto cf_read = [&](int kt, int i) { cq[i] = cfl[kt * (XBK * NC / 4) + i]; };
    auto conv_elem = [&](const XSet& S, int i) {
        if (TWO) {
            float x = S.rx[i];
            if (POOL) x = (S.rarg[POOL ? i : 0] == xkin) ? x : 0.f;
            const float4 c4 = cq[TWO ? i : 0];
            cv[i] = pro_apply<PRO_BN_BWD>(x, S.ry[TWO ? i : 0], c4.x, c4.y, c4.z, c4.w);
        } else {
            const float4 c4 = cq[TWO ? 0 : i / 2];
            cv[i] = (i & 1) ? pro_apply<PRO_AFFINE_RELU>(S.rx[i], 0.f, c4.z, c4.w, 0.f, 0.f)
                            : pro_apply<PRO_AFFINE_RELU>(S.rx[i], 0.f, c4.x, c4.y, 0.f, 0.f);
        }
    };
    auto convert_all = [&](int kt, const XSet& S, unsigned (&ph)[4], unsigned (&pl)[4]) {
#pragma unroll
        for (int i = 0; i < (TWO ? 8 : 4); ++i) cf_read(kt, i);
#pragma unroll
        for (int i = 0; i < 8; ++i) conv_elem(S, i);
#pragma unroll
        for (int j = 0; j < 4; ++j) split_pair_mix(cv[2 * j], cv[2 * j + 1], ph[j], pl[j]);
    };

    f32x16 acc[8][1];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][0][r] = 0.0f;

    // weight fragment of channel tile t: row t*32 + c, half h; (row >> 3) & 1 does not depend on t
    const int fa0 = d_lds_off(c, h);
    // fragments of one PAIR of channel tiles (2u, 2u+1): 16 registers; two sets, used alternately
    struct Frag { f16x8 lo[2], hi[2]; };
    Frag FA, FB;
    auto read_pair = [&](int kt, int u, int half, Frag& F) {
        const unsigned char* As = smem + (kt & (DSLOTS - 1)) * DSTAGE + fa0 + u * 2048;
        if (half == 0) {
            F.lo[0] = *reinterpret_cast<const f16x8*>(As + DPL);
            F.lo[1] = *reinterpret_cast<const f16x8*>(As + DPL + 1024);
        } else {
            F.hi[0] = *reinterpret_cast<const f16x8*>(As);
            F.hi[1] = *reinterpret_cast<const f16x8*>(As + 1024);
        }
    };
    // The weight DMA is inline asm: hipcc puts `s_waitcnt vmcnt(0)` in front of the next LDS read it cannot prove disjoint
    // from a pending LDS-DMA it knows about (here: the coefficient reads) -- a drain of every load in flight, once per
    // stage.  What has landed when is this kernel's own business (the counted wait in front of the barrier).
    const unsigned ring_lds = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)smem);
    auto dma_half = [&](int kt, int hf) {
        const unsigned dst = ring_lds + (unsigned)((kt & (DSLOTS - 1)) * DSTAGE) + (unsigned)(wave * 1024);
        // m0 (the LDS base of an LDS-DMA) is saved and restored INSIDE the statement: hipcc treats "m0" in a clobber list as a
        // reserved register it may ignore (344 -Winline-asm warnings); this form clobbers nothing
#pragma unroll
        for (int j = 2 * hf; j < 2 * hf + 2; ++j) {
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "s"(dst + j * 4096), "v"(a_voff), "s"(rAv), "s"(kt * DSTAGE + j * 4096) : "memory");
        }
    };
    constexpr int NX = POOL ? 24 : (TWO ? 16 : 8);             // register loads of one stage of the streamed operand
    // Memory instructions issued AFTER the last DMA instruction of the previous stage (slot 5) when this stage's barrier
    // is reached: the previous stage's loads from slot 6 on (3/4 of them, see the slot tables in stage()), this stage's
    // 4 DMA and NX loads.  Waiting until at most that many are outstanding means that DMA has landed.
    constexpr int BARRIER_VMCNT = NX * 3 / 4 + 4 + NX;

    // One 16-k stage = 24 MFMAs in program order, each followed by the few other instructions that issue while it runs
    // (an in-order wave hides ~5 issue slots behind a 32-cycle MFMA; the matrix instructions are inline asm, which the
    // compiler would otherwise bunch: eight back to back, the conversion after them -- nothing overlapped, measured equal
    // to the LDS-staged kernel).  The 8 channel tiles are taken in PAIRS (2u, 2u+1), six MFMAs per pair: X_hi.A_lo,
    // X_lo.A_hi, X_hi.A_hi for both tiles (smallest terms first; dependent MFMAs are two apart), so that only 16
    // fragment registers are in use and 16 more hold the next pair, read from LDS a pair ahead.  sched_barrier(0)
    // after every slot pins the order.
    //   slots  0..17  pairs 0-2; fillers: fragment reads of the next pair, coefficients, weight DMA of stage kt+2 (slots 4, 5),
    //                 prologue + split of stage kt+1 into the other plane set, operand loads of stage kt+1+DEPTH (each
    //                 register is reloaded right after its value is used)
    //   barrier       every wave's DMA of stage kt+1 has landed (counted wait: only what was issued after it may be
    //                 outstanding, BARRIER_VMCNT instructions)
    //   slots 18..23  pair 3; fillers: fragment reads of pair 0 of stage kt+1
    auto stage = [&](int kt, XSet& S, const unsigned (&ch)[4], const unsigned (&cl)[4], unsigned (&nh)[4], unsigned (&nl)[4]) {
        const f16x8 xh = __builtin_bit_cast(f16x8, make_uint4(ch[0], ch[1], ch[2], ch[3]));
        const f16x8 xl = __builtin_bit_cast(f16x8, make_uint4(cl[0], cl[1], cl[2], cl[3]));
        const int k1 = min(kt + 1, nk - 1), k2 = min(kt + 2, nk - 1), kx = min(kt + 1 + DEPTH, nk - 1);
        auto mfma = [&](int sl) {
            const int u = sl / 6, i = sl % 6, t = 2 * u + (i & 1);
            Frag& F = (u & 1) ? FB : FA;
            const f16x8& x = (i / 2 == 1) ? xl : xh;
            const f16x8& f = (i / 2 == 0) ? F.lo[i & 1] : F.hi[i & 1];
            if constexpr (DIRECT) asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %1, %0" : "+a"(acc[t][0]) : "v"(x), "v"(f));
            else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[t][0]) : "v"(x), "v"(f));
        };
        auto filler = [&](int sl) {
            const int u = sl / 6, i = sl % 6;
            // fragments of the next pair: (u+1) of this stage into the other set, or pair 0 of stage kt+1 into FA
            if (i < 2) {
                if (u < 3) read_pair(kt, u + 1, i, (u & 1) ? FA : FB);
                else read_pair(kt + 1, 0, i, FA);
            }
            // (not earlier: hipcc's wait in front of the first conversion of a stage is vmcnt(0) -- it would drain them)
            if (sl == 4 || sl == 5) dma_half(k2, sl - 4);
            if (!TWO) {
                if (sl == 2) { cf_read(k1, 0); cf_read(k1, 1); cf_read(k1, 2); cf_read(k1, 3); }
                constexpr int CA[4] = {3, 5, 9, 11}, CB[4] = {4, 8, 10, 14};      // slots of pair j: prologue / split + reload
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (sl == CA[j]) { conv_elem(S, 2 * j); conv_elem(S, 2 * j + 1); }
                    if (sl == CB[j]) {
                        split_pair_mix(cv[2 * j], cv[2 * j + 1], nh[j], nl[j]);
                        load_x1(kx, S, 2 * j); load_x1(kx, S, 2 * j + 1);
                    }
                }
            } else {
                // 8 prologues of 5 instructions, 4 splits of 3; a coefficient quad is read two slots before its use
                constexpr int CR[8] = {0, 0, 1, 3, 4, 6, 8, 9}, CE[8] = {2, 3, 5, 7, 8, 10, 12, 13}, CS[4] = {4, 9, 11, 14};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (sl == CR[e]) cf_read(k1, e);
                    if (sl == CE[e]) conv_elem(S, e);
                    if (sl == CE[e] + 1) load_x1(kx, S, e);       // its registers are free again
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (sl == CS[j]) split_pair_mix(cv[2 * j], cv[2 * j + 1], nh[j], nl[j]);
            }
        };
#define USIP_X2D_SLOT(N_)                                              \
        mfma(N_);                                                      \
        filler(N_);                                                    \
        __builtin_amdgcn_sched_barrier(0);
        USIP_X2D_SLOT(0) USIP_X2D_SLOT(1) USIP_X2D_SLOT(2) USIP_X2D_SLOT(3) USIP_X2D_SLOT(4) USIP_X2D_SLOT(5)
        USIP_X2D_SLOT(6) USIP_X2D_SLOT(7) USIP_X2D_SLOT(8) USIP_X2D_SLOT(9) USIP_X2D_SLOT(10) USIP_X2D_SLOT(11)
        USIP_X2D_SLOT(12) USIP_X2D_SLOT(13) USIP_X2D_SLOT(14) USIP_X2D_SLOT(15) USIP_X2D_SLOT(16) USIP_X2D_SLOT(17)
        // this wave's DMA of stage kt+1 has landed: what was issued after it may stay in flight across the barrier
        // (__syncthreads() would drain it); lgkmcnt(0): its reads of the slot are done
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(BARRIER_VMCNT) : "memory");
        __builtin_amdgcn_s_barrier();
        USIP_X2D_SLOT(18) USIP_X2D_SLOT(19) USIP_X2D_SLOT(20) USIP_X2D_SLOT(21) USIP_X2D_SLOT(22) USIP_X2D_SLOT(23)
#undef USIP_X2D_SLOT
    };

    unsigned ah[4], al[4], bh[4], bl[4];
    dma_half(0, 0); dma_half(0, 1);
    dma_half(min(1, nk - 1), 0); dma_half(min(1, nk - 1), 1);
    load_x(0, xs0);
    load_x(min(1, nk - 1), xs1);                               // stage s lives in set s & 1 (DEPTH 2) / always xs1 (DEPTH 1);
    convert_all(0, xs0, ah, al);                               // both requested at once: one memory latency per tile, not two
    if (DEPTH == 2) load_x(min(2, nk - 1), xs0);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NX) : "memory");  // the DMA (older than every load; asm: hipcc's barrier does not
    __syncthreads();                                           // wait for it): stages 0 and 1 of the weights have landed
    read_pair(0, 0, 0, FA); read_pair(0, 0, 1, FA);
    if constexpr (DEPTH == 2) {
        // two stages per trip (nk is even: the launcher sends odd nk to DEPTH 1): register sets and plane sets swap roles
        for (int kt = 0; kt < nk_run; kt += 2) {
            stage(kt, xs1, ah, al, bh, bl);
            stage(kt + 1, xs0, bh, bl, ah, al);
        }
    } else {
        for (int kt = 0; kt < nk_run; ++kt) {
            stage(kt, xs1, ah, al, bh, bl);
#pragma unroll
            for (int j = 0; j < 4; ++j) { ah[j] = bh[j]; al[j] = bl[j]; }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the clamped repeat of the last DMA (asm: hipcc does not know)
    __syncthreads();                                           // the ring is scratch from here on
    // The MFMAs are inline asm (accumulators pinned to AGPRs: with them in the unified file hipcc shuffled and spilled
    // accumulator tuples in the data-gradient forms), so the compiler does not know that the instructions before this
    // point are matrix instructions whose results need up to 18 wait states before a v_accvgpr_read: wait by hand.
#pragma unroll
    for (int t = 0; t < 8; ++t) asm volatile("s_nop 7\n\ts_nop 7" : "+a"(acc[t][0]));
    // epilogue scratch: ring slots 2 and 3 (every DMA has landed and every fragment read is done: drains + barrier above)
    if (dbg & 2) {
        if (a.P == -12345) a.Y[tid] = acc[0][0][0] + acc[1][0][1] + acc[2][0][2] + acc[3][0][3] + acc[4][0][4] + acc[5][0][5] +
                                      acc[6][0][6] + acc[7][0][7];
    } else {
        float* scr = reinterpret_cast<float*>(smem + 2 * DSTAGE);
        // GEN (chosen by the launcher when not every tile lies inside the tensor, or the stores cannot be 16-B vectors):
        // the general epilogue; otherwise only the lean one is compiled in (both in one kernel cost registers)
        if constexpr (DIRECT) epilogue_x2d_direct(a, acc, out_scale, b, m0, p0);
        else if constexpr (GEN) epilogue_x2d<EPI>(a, acc, out_scale, scr, 2 * DSTAGE / 4, b, m0, p0, tn, tpc);
        else if constexpr (REDK) epilogue_x2d_fast<EPI_NONE, false, true>(a, acc, out_scale, scr, b, m0, p0, tn, tpc);
        else if (a.rowbias) epilogue_x2d_fast<EPI, true>(a, acc, out_scale, scr, b, m0, p0, tn, tpc);
        else epilogue_x2d_fast<EPI, false>(a, acc, out_scale, scr, b, m0, p0, tn, tpc);
    }
    if constexpr (!DIRECT) __syncthreads();                    // the scratch becomes ring again (DIRECT never used it)
    }                                                          // tiles
}

}  // namespace

namespace usip_mlp {

int launch_gemm_x2d(const GemmArgs& a_in, const uint4* pl, int pro, hipStream_t st)
{
    if (!a_in.red_out && gemm_x2e_takes(a_in, pro)) return launch_gemm_x2e(a_in, pl, st);
    if (gemm_x2f_takes(a_in, pro)) return launch_gemm_x2f(a_in, pl, pro, st);   // round 6: one wave per SIMD, 64-position wave tiles
    GemmArgs a = a_in;
    a.a_trans = usip_tuning_value(USIP_TUNE_X2_DIRECT) >> 4;   // measurement aid, see the kernel (0 in the product)
    const int tpc = (a.P + DBN - 1) / DBN, nmt = (a.M + DBM - 1) / DBM;
    const long long total = (long long)a.nb * tpc * nmt;
    if (total > 0x7fffffffLL || (long long)a.K * a.P * 4 >= (1LL << 31)) return USIP_EINVAL;
    const int epi = a.stats ? EPI_STATS : EPI_NONE;
    const bool tail = (a.K % XBK) != 0;
    // two stages of operand loads in flight (DEPTH 2) needs an even number of stages; knob x2_direct = 2: DEPTH 1
    const bool deep = !tail && ((a.K / XBK) % 2 == 0) && (usip_tuning_value(USIP_TUNE_X2_DIRECT) & 15) != 2;
    // persistent: two workgroups per CU (LDS: 70 KiB each); 8-aligned so that v & 7 is the same XCD for every tile of a block
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8)
            n = 256;
        return n;
    }();
    const long long slots = (long long)(2 * cus) / 8 * 8;
    // the lean epilogue needs every tile inside the tensor, 16-B vector stores, 32-bit offsets, a row bias per run of 4
    const bool gen = !(a.y_vec && a.M % DBM == 0 && a.P % DBN == 0 && (!a.rowbias || a.rb_group % 4 == 0) &&
                       (long long)a.y_rows * a.P * 4 < (1LL << 31));
    const bool one_tile_each = (usip_tuning_value(USIP_TUNE_X2_DIRECT) & 15) == 4;     // measurement: knob x2_direct = 4
    dim3 grid((unsigned)((total <= slots || (total & 7) || one_tile_each) ? total : slots)), block(DNT);
    // data gradients whose output needs nothing but the scale: unswapped MFMA operands, stores straight from the registers
    const bool direct = !gen && !tail && deep && epi == EPI_NONE && !a.red_out && !a.bias && !a.rowbias &&
                        (usip_tuning_value(USIP_TUNE_X2_DIRECT) & 15) != 8;              // measurement: knob x2_direct = 8
    if (direct && pro == PRO_BN_BWD) {
        USIP_LAUNCH((gemm_x2d_kernel<PRO_BN_BWD, EPI_NONE, false, 2, false, false, true>), grid, block, 0, st, a, pl);
        USIP_LAUNCH_CHECK();
        return USIP_OK;
    }
    if (direct && pro == PRO_BN_BWD_POOL) {
        USIP_LAUNCH((gemm_x2d_kernel<PRO_BN_BWD_POOL, EPI_NONE, false, 2, false, false, true>), grid, block, 0, st, a, pl);
        USIP_LAUNCH_CHECK();
        return USIP_OK;
    }
#define USIP_X2D_CASE(P_, E_)                                                                  \
    if (pro == P_ && epi == E_) {                                                              \
        if (gen && tail) USIP_LAUNCH((gemm_x2d_kernel<P_, E_, true, 1, false, true>), grid, block, 0, st, a, pl);  \
        else if (gen) USIP_LAUNCH((gemm_x2d_kernel<P_, E_, false, 1, false, true>), grid, block, 0, st, a, pl); \
        else if (tail) USIP_LAUNCH((gemm_x2d_kernel<P_, E_, true, 1>), grid, block, 0, st, a, pl);  \
        else if (deep) USIP_LAUNCH((gemm_x2d_kernel<P_, E_, false, 2>), grid, block, 0, st, a, pl); \
        else USIP_LAUNCH((gemm_x2d_kernel<P_, E_, false, 1>), grid, block, 0, st, a, pl);      \
        USIP_LAUNCH_CHECK();                                                                   \
        return USIP_OK;                                                                        \
    }
    if (a.red_out) {
        // the variant whose epilogue also takes the producing layer's BatchNorm-backward sums: its own instantiation (in
        // one kernel with the plain epilogue it cost the plain launches 10 % through register pressure)
        if (tail || gen || epi != EPI_NONE || (pro != PRO_BN_BWD && pro != PRO_BN_BWD_POOL)) return USIP_EINVAL;
        if (pro == PRO_BN_BWD) {
            if (deep) USIP_LAUNCH((gemm_x2d_kernel<PRO_BN_BWD, EPI_NONE, false, 2, true>), grid, block, 0, st, a, pl);
            else USIP_LAUNCH((gemm_x2d_kernel<PRO_BN_BWD, EPI_NONE, false, 1, true>), grid, block, 0, st, a, pl);
        } else {
            if (deep) USIP_LAUNCH((gemm_x2d_kernel<PRO_BN_BWD_POOL, EPI_NONE, false, 2, true>), grid, block, 0, st, a, pl);
            else USIP_LAUNCH((gemm_x2d_kernel<PRO_BN_BWD_POOL, EPI_NONE, false, 1, true>), grid, block, 0, st, a, pl);
        }
        USIP_LAUNCH_CHECK();
        return USIP_OK;
    }
    USIP_X2D_CASE(PRO_AFFINE_RELU, EPI_STATS)
    USIP_X2D_CASE(PRO_AFFINE_RELU, EPI_NONE)
    USIP_X2D_CASE(PRO_BN_BWD, EPI_NONE)
    USIP_X2D_CASE(PRO_BN_BWD_POOL, EPI_NONE)
#undef USIP_X2D_CASE
    return USIP_EINVAL;
}

}  // namespace usip_mlp
