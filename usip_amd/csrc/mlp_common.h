// usip_amd/csrc/mlp_common.h -- argument blocks, prologue math and the output epilogue shared by the
// shared-MLP GEMM kernels (shared_mlp.hip: fp32 MFMA; shared_mlp_bf16.hip: bf16 multiply, fp32 accumulate).
#pragma once
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace usip_mlp {

// Cache policy of the big streams (round 6).  A store or load of the shared-MLP kernels can be marked non-temporal
// (`... nt`; buffer instructions: aux = 2): the line is not kept in L2 / the Infinity Cache in preference to other data.
// Whether that helps is a property of the STEP, not of the kernel (stand-alone the kernels time the same): it was measured
// site by site, one variant build per bit, whole step, same box, alternating (profiles/r06z_nt_cache_policy_ab.txt).
// USIP_ST_NT is the bit mask of the sites that are nt; the default is the set that gained:
//   ST_X2F_DGRAD    the 8-byte stores of the data gradient of the 256/512-wide layers          -0.08 ms per step
//   ST_WGRAD_PART   partial weight-gradient tiles (read once, by the reduction behind backward) -0.01
//   LD_WGRAD_G      (dZ, Y) in the weight gradient: their last use in the step                  -0.02
//   LD_LAYER_BWD_G  (dZ, Y) in the fused backward of the 64/128-wide layers: their last use     -0.02 ... -0.06
//   LD_X2F_FWD      the streamed operand of the 256/512-wide forward GEMMs (next use: backward)  -0.04
// together 4.60 -> 4.42 ms on one box.  Measured and NOT taken (neutral within 0.01 ms unless a figure is given):
// ST_NARROW_FWD, ST_X2R, ST_X2F_FWD, ST_X2D, ST_LAYER_BWD_DX, LD_NARROW_FWD, LD_X3P_FWD, LD_WGRAD3_G; ST_TILE_EPI (+0.24:
// conv1's output is read at once by conv2), ST_TILE_EPI_PLAIN (+0.06), LD_X2F_DGRAD (+0.03: the weight gradient re-reads
// them), LD_X2R (+0.02), LD_WGRAD_X (+0.03: the previous layer's backward re-reads it), LD_LAYER_BWD_X (+0.03),
// LD_BN_REDUCE_Z / _Y (the stand-alone BatchNorm-backward reduction: +0.00 / +0.01, both +0.03), LD_WGRAD_GEN_G (+0.02); ST_X2R split by launch (conv4's / conv5's output): -0.01 / 0.00.
enum { ST_NARROW_FWD = 1, ST_X2R = 2, ST_TILE_EPI = 4, ST_X2F_FWD = 8, ST_X2F_DGRAD = 16, ST_X2D = 32, ST_LAYER_BWD_DX = 64,
       ST_WGRAD_PART = 128, LD_WGRAD_G = 256, ST_TILE_EPI_PLAIN = 512, LD_LAYER_BWD_G = 1024, LD_X2F_FWD = 2048,
       LD_X2F_DGRAD = 4096, LD_NARROW_FWD = 8192, LD_X2R = 16384, LD_WGRAD_X = 32768, LD_WGRAD3_G = 65536,
       LD_X3P_FWD = 131072, LD_LAYER_BWD_X = 262144, LD_BN_REDUCE_Z = 524288, LD_BN_REDUCE_Y = 1048576, LD_WGRAD_GEN_G = 2097152, ST_X2R_PLAIN = 4194304 };
#ifndef USIP_ST_NT
#define USIP_ST_NT (ST_X2F_DGRAD | ST_WGRAD_PART | LD_WGRAD_G | LD_LAYER_BWD_G | LD_X2F_FWD)
#endif
template <int SITE> constexpr int st_aux() { return (USIP_ST_NT & SITE) ? 2 : 0; }
typedef float st_f4v __attribute__((ext_vector_type(4)));
template <int SITE>
__device__ __forceinline__ void st_out(float* p, float v)
{
    if (USIP_ST_NT & SITE) __builtin_nontemporal_store(v, p); else *p = v;
}
template <int SITE>
__device__ __forceinline__ float4 ld_in4(const float* p)
{
    if (USIP_ST_NT & SITE) {
        const st_f4v q = __builtin_nontemporal_load(reinterpret_cast<const st_f4v*>(p));
        return make_float4(q.x, q.y, q.z, q.w);
    }
    return *reinterpret_cast<const float4*>(p);
}
template <int SITE>
__device__ __forceinline__ void st_out4(float* p, float x, float y, float z, float w)
{
    if (USIP_ST_NT & SITE) { st_f4v q = {x, y, z, w}; __builtin_nontemporal_store(q, reinterpret_cast<st_f4v*>(p)); }
    else *reinterpret_cast<float4*>(p) = make_float4(x, y, z, w);
}

enum { PRO_NONE = 0, PRO_AFFINE_RELU = 1, PRO_BN_BWD = 2, PRO_BN_BWD_POOL = 3 };
// PRO_BN_BWD_POOL: the layer's output went ONLY into a max over K neighbours, so its incoming gradient is
// dZ[c][m][k] = (k == arg[c][m]) ? dpooled[c][m] : 0.  It is synthesised from the two small [C][M] arrays
// instead of being written as a dense tensor by the pooling backward and read back three times.

struct GemmArgs {
    const float* At; int lda;          // [K][M], row stride lda
    const float* X;                    // [nb][K][P]
    const float* X2;                   // [nb][K][P]  (PRO_BN_BWD: the layer's pre-BN output Y)
    const float* coef;                 // [4][K] prologue coefficients per input channel
    const float* bias;                 // [M] or null
    float* Y;                          // [nb][M][P]
    float* stats;                      // [2][M][ntn] or null
    int M, K, P, nb;
    const float* rowbias; int rb_group; // Y += rowbias[b][m][p / rb_group]  ([nb][M][P/rb_group]) or null
    const float* pool_dp; const int* pool_arg; int pool_group;   // PRO_BN_BWD_POOL: [nb][K][P/group] each
    int a_trans;                        // 1: the matrix operand is stored [M][K] (row stride lda), read transposed
    int y_rows;                         // rows per cloud of the tensor Y points into (>= M): Y[b] = Y + b*y_rows*P
    int y_vec;                          // 1: P % 4 == 0 and Y 16-B aligned: runs of 4 positions are stored as one float4
    // gemm_x2d.hip, data-gradient launches only (all null / 0 elsewhere): the output dX is the gradient of the lazily
    // activated output of the layer whose pre-BN tensor is red_y [nb][M][P] with coefficients red_coef [4][M] (scale,
    // shift, mean, invstd); the epilogue leaves that layer's BatchNorm-backward partial sums in red_out =
    // [2][tiles][M] (sum d, sum d xhat with d = dX where relu is on) + [tiles * row tiles] maxima of |d|, tiles = nb *
    // ceil(P / 128); red_gsum (optional) [2][nb * M][P / red_group]: per-neighbourhood sums of d and of y
    const float* red_y; const float* red_coef; float* red_out; float* red_gsum; int red_group;
};

// prologue on one element of the streamed operand, channel coefficients c0..c3
template <int PRO>
__device__ __forceinline__ float pro_apply(float x, float x2, float c0, float c1, float c2, float c3)
{
    if (PRO == PRO_AFFINE_RELU) return fmaxf(__builtin_fmaf(x, c0, c1), 0.0f);
    if (PRO == PRO_BN_BWD) {
        // x = dZ, x2 = Y (pre-BN).  a1 = gamma*invstd, a0 = beta - mean*a1 reproduce the forward's
        // z = relu(fma(y, a1, a0)) decision exactly; dY = a1*dYhat + q1*y + q0 (see bn_bwd_finalize).
        const float dyh = (__builtin_fmaf(x2, c0, c1) > 0.0f) ? x : 0.0f;
        return __builtin_fmaf(c0, dyh, __builtin_fmaf(c2, x2, c3));
    }
    return x;
}

// EPI: 0 none | 1 BatchNorm statistics of Y: per-tile (sum, sum^2) partials
enum { EPI_NONE = 0, EPI_STATS = 1 };

// Epilogue shared by the fp32 and the bf16-multiply kernels: + bias (+ row bias), store, BatchNorm partial
// statistics.  The kernels issue their MFMAs with the operands SWAPPED (streamed operand as A, matrix operand as
// B), so an accumulator tile is D'[position][channel]: lane l holds ONE channel (l & 31) and, per 32 x 32 tile,
// four runs of four consecutive positions (r = 4g + e  ->  position 8g + 4(l >> 5) + e).  Hence
//   * a run is one 16-B store (16 stores per thread instead of 64 scalar ones: the store tail of the short-K
//     layers was issue-bound -- 17 % of a 128x64 forward GEMM);
//   * bias and the channel's statistics are lane-local: plain adds plus ONE cross-half exchange per channel
//     instead of a 32-lane reduction per accumulator register.
// `scratch` is LDS the main loop no longer needs (>= 2*WN*BM floats; what is left holds the tile's row bias when
// it fits).
// NJ = 32-position tiles per wave (2: a wave covers 64 positions; 1: gemm_x2d.hip, a wave covers 32 positions of
// every channel of the tile).
template <int WM, int WN, int EPI, int TM = 2, int NJ = 2>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& a, f32x16 (&acc)[TM][NJ], float* scratch,
                                              int scratch_floats, int b, int m0, int p0, int tn, int tpc)
{
    constexpr int BM = WM * 32 * TM, BN = WN * 32 * NJ;
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    float* Yb = a.Y + (long long)b * a.y_rows * a.P;
    float* red = scratch;                                    // [2][WN][BM]
    const int ngrp = a.rowbias ? a.P / a.rb_group : 0;
    // Row bias of this tile -> LDS once ([BM][G], G = neighbourhoods the tile's positions span) instead of one
    // global load per output element.
    float* rbs = red + 2 * WN * BM;
    const int g0 = a.rowbias ? p0 / a.rb_group : 0;
    const int G = a.rowbias ? (min(p0 + BN, a.P) - 1) / a.rb_group - g0 + 1 : 0;
    const bool rb_lds = a.rowbias && BM * G <= scratch_floats - 2 * WN * BM;
    if (rb_lds) {
        for (int e = tid; e < BM * G; e += 256) {
            const int rl = e / G, g = e % G;
            rbs[e] = (m0 + rl < a.M) ? a.rowbias[((long long)b * a.M + m0 + rl) * ngrp + g0 + g] : 0.f;
        }
        __syncthreads();
    }
    const bool run_one_group = a.rowbias && (a.rb_group % 4 == 0);   // a run of 4 positions never straddles a group
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row_l = (wm * TM + i) * 32 + c;            // the lane's channel in this tile row
        const int row = m0 + row_l;
        const int rowc = min(row, a.M - 1);
        const bool rok = row < a.M;
        const float bv = a.bias ? a.bias[rowc] : 0.0f;
        float* yrow = Yb + (long long)rowc * a.P;
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int pb = p0 + wn * (32 * NJ) + j * 32 + 8 * g + 4 * half;   // first position of the run
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e] + bv;
                if (a.rowbias) {
                    if (run_one_group) {
                        const int grp = min(pb, a.P - 1) / a.rb_group;
                        const float rb = rb_lds ? rbs[row_l * G + grp - g0]
                                                : a.rowbias[((long long)b * a.M + rowc) * ngrp + grp];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += rb;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int grp = min(pb + e, a.P - 1) / a.rb_group;
                            v[e] += rb_lds ? rbs[row_l * G + grp - g0]
                                           : a.rowbias[((long long)b * a.M + rowc) * ngrp + grp];
                        }
                    }
                }
                if (rok) {
                    if (a.y_vec && pb + 3 < a.P) {
                        st_out4<(EPI == EPI_STATS) ? ST_TILE_EPI : ST_TILE_EPI_PLAIN>(yrow + pb, v[0], v[1], v[2], v[3]);
                        if (EPI == EPI_STATS) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) { s += v[e]; q = __builtin_fmaf(v[e], v[e], q); }
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (pb + e < a.P) {
                                yrow[pb + e] = v[e];
                                if (EPI == EPI_STATS) { s += v[e]; q = __builtin_fmaf(v[e], v[e], q); }
                            }
                    }
                }
            }
        }
        if (EPI != EPI_NONE) {
            s += __shfl_xor(s, 32); q += __shfl_xor(q, 32);  // the other half-wave holds the channel's other positions
            if (half == 0) {
                red[wn * BM + row_l] = s;
                red[WN * BM + wn * BM + row_l] = q;
            }
        }
    }
    if (EPI != EPI_NONE) {
        __syncthreads();
        if (tid < BM && m0 + tid < a.M) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int w = 0; w < WN; ++w) { s += red[w * BM + tid]; q += red[WN * BM + w * BM + tid]; }
            const long long ntn = (long long)a.nb * tpc;
            // layout [2][M][tiles]: the finalisation kernels then read a channel's partials contiguously
            a.stats[(long long)(m0 + tid) * ntn + tn] = s;
            a.stats[ntn * a.M + (long long)(m0 + tid) * ntn + tn] = q;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// weight gradient: dW[m=co][n=ci] = sum over positions of pro(G)[co][p] * X[ci][p]
struct WgradArgs {
    const float* G;  const float* G2; const float* coef;   // [nb][M][P] (+ Y and [4][M] for PRO_BN_BWD)
    const float* X;                                         // [nb][N][P]
    const float* xcoef;                                     // [2][N] or null: X := relu(X*xcoef[0][n] + xcoef[1][n])
    const float* pool_dp; const int* pool_arg; int pool_group;   // PRO_BN_BWD_POOL: [nb][M][P/group] each
    float* part;                                            // [slices][M][N]
    int M, N, P, nb, seglen, segs;                          // segs position segments per cloud
};

// One stage of a [rows][32 positions] operand tile of the weight gradient: thread -> (row, 4
// consecutive positions), 8 lanes cover one 128-B row segment.  Raw loads only; the BatchNorm-backward
// prologue runs when the registers are written to LDS (after the MFMAs the loads overlap with).
template <int N4, bool TWO, bool VEC, int Q4 = 8>
__device__ __forceinline__ void wgrad_load_rows(const float* __restrict__ base, const float* __restrict__ base2,
                                                int rows, int P, int r0, int p, int pend, int tid,
                                                float4 (&dst)[N4], float4 (&dst2)[TWO ? N4 : 1])
{
#pragma unroll
    for (int i = 0; i < N4; ++i) {
        const int f = tid + i * 256, row = f / Q4, kq = (f % Q4) * 4;         // Q4 float4 per row and stage
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f), w = v;
        if (VEC) {
            // P % 4 == 0 and segments start at multiples of 32: a float4 is entirely inside or outside.
            // Branch-free: clamped (valid) address, zeroed afterwards.
            const long long off = (long long)min(r0 + row, rows - 1) * P + min(p + kq, P - 4);
            if (TWO) {                                                  // (dZ, Y): masked at LDS-store time
                v = ld_in4<LD_WGRAD_GEN_G>(base + off);
                w = ld_in4<LD_WGRAD_GEN_G>(base2 + off);
            } else {
                v = *reinterpret_cast<const float4*>(base + off);
            }
        } else if (r0 + row < rows) {
            const long long off = (long long)(r0 + row) * P + p + kq;
            if (p + kq + 0 < pend) { v.x = base[off + 0]; if (TWO) w.x = base2[off + 0]; }
            if (p + kq + 1 < pend) { v.y = base[off + 1]; if (TWO) w.y = base2[off + 1]; }
            if (p + kq + 2 < pend) { v.z = base[off + 2]; if (TWO) w.z = base2[off + 2]; }
            if (p + kq + 3 < pend) { v.w = base[off + 3]; if (TWO) w.w = base2[off + 3]; }
        }
        dst[i] = v;
        if (TWO) dst2[i] = w;
    }
}

// Accumulator tiles of one wave -> the slice's partial [M][N] tile in the workspace.
template <int TM, int TN>
__device__ __forceinline__ void wgrad_store_partial(const WgradArgs& a, f32x16 (&acc)[TM][TN], int slice,
                                                    int m0, int n0, int wm, int wn, int lane)
{
    float* out = a.part + (long long)slice * a.M * a.N;
    const int half = lane >> 5, c = lane & 31;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int col = n0 + (wn * TN + j) * 32 + c;
                if (row < a.M && col < a.N) st_out<ST_WGRAD_PART>(out + (long long)row * a.N + col, acc[i][j][r]);
            }
}

// bf16-multiply variants (shared_mlp_bf16.hip); same argument blocks, same outputs up to operand rounding
int launch_gemm_bf16(const GemmArgs& a, int pro, hipStream_t st);
int launch_wgrad_bf16(const WgradArgs& a, int pro, bool xpro, bool vec, int small, unsigned blocks, hipStream_t st);
int launch_wgrad_reduce(const float* part, float* dW, long long elems, int slices, int N, int ldw, int coloff,
                        hipStream_t st);
// fp32-accurate products out of three bf16 planes per operand (128 x 128 tiles only; see shared_mlp_bf16.hip)
int launch_gemm_x3(const GemmArgs& a, int pro, hipStream_t st);
int launch_wgrad_x3(const WgradArgs& a, int pro, bool xpro, bool vec, unsigned blocks, hipStream_t st);
// 256 x 256 tiles, own slicing (shared_mlp_x3.hip); needs the vector path (P % 4 == 0, 16-B aligned operands)
void wgrad_x3_plan(int M, int N, int P, int nb, int* seglen, int* segs, int* tiles);
int launch_wgrad_x3_256(const WgradArgs& a, int pro, bool xpro, unsigned blocks, hipStream_t st);
// f32x2 forward / data gradient with the streamed operand global -> registers -> MFMA (gemm_x2d.hip); `pl` = the
// usip_mlp_split2h_f32 image with 256-row tiles; tiles of 256 channels x 128 positions
int launch_gemm_x2d(const GemmArgs& a, const uint4* pl, int pro, hipStream_t st);
// gemm_x2e.hip: the forward launches of that family with both operands by LDS-DMA (one 8-wave workgroup per CU)
bool gemm_x2e_takes(const GemmArgs& a, int pro);
int launch_gemm_x2e(const GemmArgs& a, const uint4* pl, hipStream_t st);
// gemm_x2f.hip (round 6): the launches of that family with whole 256 x 256 tiles as ONE wave per SIMD, 64 positions per wave
bool gemm_x2f_takes(const GemmArgs& a, int pro);
int launch_gemm_x2f(const GemmArgs& a, const uint4* pl, int pro, hipStream_t st);
// the same from two fp16 planes per operand (pro 2 / 3 with the [5][M] coef4, xcoef = [4][N] batch statistics)
int launch_wgrad_x2h_256(const WgradArgs& a, int pro, unsigned blocks, hipStream_t st);

}  // namespace usip_mlp
