// usip_amd/csrc/layer_bwd_x2.hip -- backward of a shared-MLP layer with <= 128 inputs and <= 128 outputs as ONE kernel,
// f32x2 arithmetic (two fp16 planes per operand, three plane products: shared_mlp_x3.hip): conv2, conv3 and the pooled
// conv5 of RPN_Detector_Ball (models/networks.py:705-712; the layers' backward is autograd's in the reference:
// models/layers.py:208-216, :293-303).  (The feature half of conv4, 64 -> 128, is the same template with one LDS buffer; the
// dispatcher launches it since round 3.)
//
// These layers are HBM-bound and their (dZ, Y) pair used to be read two or three times per step (BatchNorm-backward
// reduction, data-gradient GEMM, weight-gradient GEMM).  narrow_bwd.hip fused the two products for 64-input layers
// with exact-fp32 MFMAs (v_mfma_f32_32x32x2_f32) and ended up matrix-pipe bound at half the HBM rate (3.5-3.9 TB/s,
// MFMA pipe 50-58 % busy).  Here the same fusion runs on the 16-bit matrix cores at 1/5 of the matrix time, with the
// lessons of gemm_x2r_kernel: weight fragments of the data-gradient resident in registers, D[channel][position]
// accumulators so that a store instruction writes two full 128-B lines, per-lane running sums instead of per-tile
// reductions.  A workgroup walks every G-th BP-position tile (round-robin: the chip streams consecutive tiles):
//     dY        = BatchNorm'(ReLU'(dZ)) from (dZ, Y, coef4)    (POOL: dZ = (k == arg) ? dpooled : 0, never stored)
//     dX[ci][p] = sum_co W[co][ci] dY[co][p]                    written tile by tile
//     dW[co][ci]+= sum_p dY[co][p] act(X)[ci][p]               accumulated in registers over the workgroup's tiles
//     RED: s1[ci] += dX [relu on], s2[ci] += dX [relu on] xhat, max |dX [relu on]|   (the producing layer's BatchNorm
//          backward sums and the bound its own f32x2 backward needs; one partial per workgroup).  The tile of dX also
//          goes to LDS and is summed by the X loader's threads (one channel, 8 positions each: two registers of state
//          and per-thread coefficients) -- per-lane sums in the accumulator layout cost 32 registers and spilled.
// The two products contract over different indices, so dY sits in LDS in both orientations: [position][co] chunks of 8
// channels (B operand of dX; written as 16-B chunks) and [co][position] chunks of 8 positions (A operand of dW; written
// as 2-B elements, lanes = consecutive positions); act(X) only as [ci][position] -- its loader thread owns 8 consecutive
// positions of one channel and writes whole 16-B chunks.  Every image row is padded by 16 B (144- / 272- / 80-B strides):
// conflict-free ds_read_b128 with every address = one base register + an immediate (XOR-swizzled rows cost a
// register per address: the first build spilled).  Operand scales are exact powers of two from rigorous bounds (shared_mlp_x3.hip): dY from
// row 4 of coef4, act(X) from the producer's batch statistics, W from the trailer of its split image.
#include <type_traits>

#include "mlp_common.h"
#include "split_common.h"

using namespace usip_mlp;

namespace {

struct LayerBwdArgs {
    const float* dZ; const float* Y; const float* coef4;       // [nb][COUT][P] x2 (POOL: dZ unused), [5][COUT]
    const float* pool_dp; const int* pool_arg; int pool_group; // POOL: [nb][COUT][P / pool_group] each
    const float* X; int x_rows; const float* xcoef;            // [nb][x_rows][P] (rows [0, CIN) used), [4][CIN]
    const uint4* planes;                                       // usip_mlp_split2h_f32 image of W as the dgrad operand (CIN rows, K = COUT)
    float* dX; int dx_rows;                                    // [nb][dx_rows][P], rows [0, CIN) written
    float* part;                                               // [workgroups][COUT][CIN]
    float* red;                                                // RED: [2][workgroups][CIN] sums, then [workgroups] maxima
    float* gsum;                                               // RED + POOL, may be null: [2][nb * CIN][P / pool_group] per-neighbourhood sums
    int P, nb;
    // WS (round 6): the PRODUCING layer's weight gradient on the way.  That layer's input S [nb][ws_rows <= 8][P] needs no
    // gradient (the detector's first layer), so all its backward still has to do is dW' = dY' . S^T with
    // dY' = a1' dYhat' + q1' (x - mu') + (q1' mu' + q0') -- linear in three sums this pass can take while it holds
    // dYhat' = dX [relu on] and x in LDS anyway:  wsum [workgroups][CIN][16] = sum_p dYhat' S_j | sum_p (x - mu') S_j
    // (j < 8), wsum3 [workgroups][8] = sum_p S_j.  usip_mlp_wsum_finalize_f32 combines them once that layer's coef4 is known.
    const float* wsrc; int ws_rows; float* wsum; float* wsum3;
};

// byte offset of (row, position) in a [rows][BP positions] fp16 image: rows of BP * 2 + 16 bytes
template <int BP>
__device__ __forceinline__ int rows_off(int row, int pos) { return row * (BP * 2 + 16) + pos * 2; }

// DB: two LDS buffers and ONE barrier per tile -- the next tile is written while this one is multiplied (gemm_x2r_kernel's
// pipeline); without it a tile is written, a barrier, multiplied, a barrier (the forms that also keep fp32 images of X
// and dX for the producing layer's sums have no room for a second buffer).
// PG (pooled form, pool_group a multiple of the tile): the tile lies inside ONE neighbourhood, so its (dpooled, arg) pair
// per channel -- 2 COUT values -- is fetched by 2 COUT threads one tile ahead and handed over through LDS, instead of 2 GC
// broadcast loads per thread and tile (those tiny requests were a third of the kernel's L2 requests; the hand-over
// changed nothing in time but frees 15 registers).
// SPLIT: the first half of the waves does the data gradient (and holds the weight fragments), the second half the weight
// gradient (and holds its accumulators) -- the two register-hungry roles no longer add up in every wave.
template <int CIN, int COUT, bool POOL, bool RED, int NW, int BP, bool DB, bool PG, bool SPLIT, bool WS>
__device__ __forceinline__ void layer_bwd_x2_body(const LayerBwdArgs& a)
{
    static_assert(!WS || (RED && !DB && !POOL && BP == 32 && CIN * BP / 8 == 64 * NW), "WS: the single-buffer RED form, 8 threads per source piece row");
    constexpr int NT = 64 * NW;
    constexpr int GC = COUT * BP / NT;                         // output channels per (dZ, Y) loader thread: 8 or 16
    constexpr bool EARLY = (GC == 8) && !(RED && DB);          // next tile's loads re-issued inside write_tile (else: register pressure)
    constexpr int NCI = CIN / 32, NCO = COUT / 32, NPT = BP / 32;
    constexpr int NDX = NCI * NPT;                             // 32 x 32 tiles of dX per position tile: one per wave
    constexpr int NDW = NCO * NCI / NW;                        // 32 x 32 tiles of dW per wave (same co tile)
    constexpr int KS = COUT / 16, PS = BP / 16;                // k-steps of dX (over co), of dW (over positions)
    constexpr int XPC = BP / 8, XRP = NT / XPC, NPX = CIN / XRP;   // X loader: 8-position pieces per row, rows per pass, passes
    constexpr int RB1 = COUT * 2 + 16, PL1 = BP * RB1;         // G1 [plane][position][COUT (+ 16 B)]
    constexpr int RS2 = BP * 2 + 16;                           // G2 [plane][co][BP (+ 16 B)], X2 [plane][ci][BP (+ 16 B)]
    constexpr int PL2 = COUT * RS2, PLX = CIN * RS2;
    constexpr int BUF = 2 * PL1 + 2 * PL2 + 2 * PLX;           // one buffer: both planes of the three images
    static_assert(NDX <= NW && (NCO * NCI) % NW == 0 && (GC == 8 || GC == 16) && CIN % XRP == 0 && (!RED || NPX == 1) &&
                  true, "tile roles");
    constexpr int XRS = BP + 4;                                // floats per row of the fp32 images (16 B of padding, as above)
    __shared__ __attribute__((aligned(16))) unsigned char smem[(DB ? 2 : 1) * BUF + (RED ? 2 * CIN * XRS * 4 : 16)];
    __shared__ float cfG[4][COUT];
    __shared__ float cfX[2][CIN];
    __shared__ float redm[2][NW];
    __shared__ unsigned poolv[PG ? 2 : 1][2][PG ? COUT : 1];    // PG: [tile parity][dpooled | arg][channel]
    __shared__ __attribute__((aligned(16))) float GS[WS ? 2 : 1][WS ? 8 : 1][WS ? BP : 4];   // WS: [tile parity][source row][position]
    static_assert(!PG || (POOL && 2 * COUT <= NT), "PG");
    // RED: the tile of dX for the sums' pass -- one buffer behind an fp32 copy of the X tile (XR), or, with two LDS buffers
    // per tile (DB), two dX buffers and the raw X tile kept in registers instead
    float* XR = reinterpret_cast<float*>(smem + (DB ? 2 : 1) * BUF);
    float* DX = XR + ((RED && !DB) ? CIN * XRS : 0);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, kh = lane >> 5;
    // Tiles are dealt round-robin: workgroup w takes tiles w, w + G, w + 2 G, ... of the nb * P / BP tiles, so that at any
    // moment the G workgroups stream G consecutive tiles -- the same DRAM pages of every channel row (r03: with one
    // contiguous 2048-position segment per workgroup every 128-B access of the chip hit a different page of a different
    // row: 3.5-4.0 TB/s where gemm_x2r_kernel, which interleaves, reaches 4.6-4.8).
    // (With per-neighbourhood sums the unit dealt is a RUN of pool_group / BP tiles: a neighbourhood stays in one workgroup.)
    const int tpc = a.P / BP, G = gridDim.x;
    const int tpg = (RED && POOL && a.gsum) ? a.pool_group / BP : 1;
    const int total = a.nb * tpc / tpg;                        // runs
    const int ntile = blockIdx.x < total ? ((total - blockIdx.x + G - 1) / G) * tpg : 0;
    auto cloud_of = [&](int t, int& p0) {
        const int T = (blockIdx.x + (t / tpg) * G) * tpg + t % tpg;
        const int bb = T / tpc;
        p0 = (T - bb * tpc) * BP;
        return bb;
    };

    // operand scales
    float bg = 0.f, bx = 0.f;
    for (int i = tid; i < (COUT + 63) / 64; i += NT) bg = fmaxf(bg, a.coef4[4 * COUT + i]);
    {
        const float rn = sqrtf((float)a.nb * (float)a.P);
        for (int k = tid; k < CIN; k += NT) {
            const float c0 = a.xcoef[k], c1 = a.xcoef[CIN + k], mu = a.xcoef[2 * CIN + k], is = a.xcoef[3 * CIN + k];
            bx = fmaxf(bx, fabsf(c0) / is * rn + fabsf(__builtin_fmaf(mu, c0, c1)));
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { bg = fmaxf(bg, __shfl_xor(bg, off)); bx = fmaxf(bx, __shfl_xor(bx, off)); }
    if (lane == 0) { redm[0][wave] = bg; redm[1][wave] = bx; }
    __syncthreads();
    bg = redm[0][0]; bx = redm[1][0];
#pragma unroll
    for (int w = 1; w < NW; ++w) { bg = fmaxf(bg, redm[0][w]); bx = fmaxf(bx, redm[1][w]); }
    const float sG = pow2_scale(bg, X2H_TOP), sX = pow2_scale(bx, X2H_TOP);
    const float ws = __uint_as_float(a.planes[(long long)KS * 512].x);     // behind the image (one 128-row M tile)
    const float dx_scale = 1.0f / (sG * ws), dw_scale = 1.0f / (sG * sX);
    for (int i = tid; i < 4 * COUT; i += NT) cfG[i / COUT][i % COUT] = a.coef4[i] * sG;
    for (int i = tid; i < 2 * CIN; i += NT) cfX[i / CIN][i % CIN] = a.xcoef[i] * sX;

    // roles.  dX: wave w < NDX owns input tile w % NCI at position tile w / NCI, its weight fragments stay in registers.
    // dW: a wave owns NDWR consecutive tiles T (co tile T / NCI, ci tile T % NCI); see `run` below.
    const int dx_ci = wave % NCI, dx_pt = (wave / NCI) % NPT;
    float s1 = 0.f, s2 = 0.f, mx = 0.f;                        // RED: the thread's channel (its X row), its 8 positions of every tile
    float rxk[(RED && DB) ? 2 : 1][8];                         // RED + DB: the raw X of the tiles in the two LDS buffers
    float gd = 0.f, gy = 0.f;                                  // per-neighbourhood sums of the current run (a.gsum)
    typedef float ws_f32x2 __attribute__((ext_vector_type(2)));
    ws_f32x2 wSa[WS ? 8 : 1], wSb[WS ? 8 : 1];                 // WS: S1_j, S2_j of this thread's channel, each as (even, odd) positions: packed FMAs on
                                                               // natural register pairs (no half is selected or swapped: tests/test_kernel_isa.py)
    float wS3 = 0.f;
    float4 rgs = make_float4(0.f, 0.f, 0.f, 0.f);              // WS: threads [0, 64): 4 positions of one source row of the next tile
#pragma unroll
    for (int j = 0; j < (WS ? 8 : 1); ++j) { wSa[j] = ws_f32x2{0.f, 0.f}; wSb[j] = ws_f32x2{0.f, 0.f}; }

    // loaders.  (dZ, Y): thread -> position gp of the tile, the GC output channels [GC gg, GC gg + GC); buffer loads with
    // the row as a scalar offset.  X: thread -> 8 consecutive positions (piece xq) of row xr0 (+ XRP per pass).
    const int gp = tid % BP, gg = tid / BP;
    const int pgrp = POOL ? a.P / a.pool_group : 0;
    const unsigned cloud_bytes = (unsigned)COUT * (unsigned)a.P * 4u, pool_bytes = (unsigned)COUT * (unsigned)pgrp * 4u;
    auto rsrc_Z = [&](int bb) { return __builtin_amdgcn_make_buffer_rsrc((void*)((POOL ? a.Y : a.dZ) + (long long)bb * COUT * a.P), 0, cloud_bytes, 0x00020000); };
    auto rsrc_Y = [&](int bb) { return __builtin_amdgcn_make_buffer_rsrc((void*)(a.Y + (long long)bb * COUT * a.P), 0, cloud_bytes, 0x00020000); };
    auto rsrc_Pd = [&](int bb) { return __builtin_amdgcn_make_buffer_rsrc((void*)(POOL ? a.pool_dp + (long long)bb * COUT * pgrp : a.Y), 0, POOL ? pool_bytes : 4u, 0x00020000); };
    auto rsrc_Pa = [&](int bb) { return __builtin_amdgcn_make_buffer_rsrc((void*)(POOL ? (const float*)(a.pool_arg + (long long)bb * COUT * pgrp) : a.Y), 0, POOL ? pool_bytes : 4u, 0x00020000); };
    const int xq = tid % XPC, xr0 = tid / XPC;
    float rz[GC], ry[GC], rx[NPX][8];
    int ra[POOL ? GC : 1], rkin = 0;
    auto load_gy = [&](int t) {
        int p0;
        const int bb = cloud_of(t, p0);
        const int p = p0 + gp;
        const __amdgpu_buffer_rsrc_t rZ = rsrc_Z(bb), rY = rsrc_Y(bb), rPd = rsrc_Pd(bb), rPa = rsrc_Pa(bb);
        const int voff = (int)(((unsigned)p + (unsigned)(GC * gg) * (unsigned)a.P) * 4u);
        if (PG) {
            // (the pair comes through poolv)
        } else if (POOL) {
            const int g = p / a.pool_group;
            rkin = p - g * a.pool_group;
            const int poff = (int)(((unsigned)g + (unsigned)(GC * gg) * (unsigned)pgrp) * 4u);
#pragma unroll
            for (int i = 0; i < GC; ++i) {
                rz[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rPd, poff, i * pgrp * 4, 0));
                ra[POOL ? i : 0] = (int)__builtin_amdgcn_raw_buffer_load_b32(rPa, poff, i * pgrp * 4, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < GC; ++i)
                rz[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rZ, voff, i * a.P * 4, st_aux<LD_LAYER_BWD_G>()));
        }
#pragma unroll
        for (int i = 0; i < GC; ++i)
            ry[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rY, voff, i * a.P * 4, st_aux<LD_LAYER_BWD_G>()));
    };
    auto load_x = [&](int t) {
        int p0;
        const int bb = cloud_of(t, p0);
        const float* xbase = a.X + ((long long)bb * a.x_rows + xr0) * a.P + p0 + xq * 8;
#pragma unroll
        for (int q = 0; q < NPX; ++q) {
            const float* src = xbase + (long long)q * XRP * a.P;
            const float4 u = ld_in4<LD_LAYER_BWD_X>(src), v = ld_in4<LD_LAYER_BWD_X>(src + 4);
            rx[q][0] = u.x; rx[q][1] = u.y; rx[q][2] = u.z; rx[q][3] = u.w;
            rx[q][4] = v.x; rx[q][5] = v.y; rx[q][6] = v.z; rx[q][7] = v.w;
        }
        if (WS && tid < 64) {                                  // row tid / 8, positions 4 (tid % 8) .. + 3 (rows beyond ws_rows: zeros)
            const int j = tid >> 3;
            rgs = (j < a.ws_rows) ? *reinterpret_cast<const float4*>(a.wsrc + ((long long)bb * a.ws_rows + j) * a.P + p0 + 4 * (tid & 7))
                                  : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    // One tile from the prefetched registers into LDS.  The loads of the NEXT tile (the last tile: a harmless repeat) are re-issued as soon as
    // a register group has been consumed -- in front of the splits and the LDS writes, not behind them: with the loads
    // issued after this phase a tile cost a full memory latency on top of it (4.3 us per 32-position tile).
    unsigned pzn = 0;                                          // PG: this thread's value (dpooled or arg of one channel) of the tile after next
    const int pch = tid % COUT, pwhich = tid / COUT;           // PG: threads [0, COUT) fetch dpooled, [COUT, 2 COUT) arg
    auto load_pool = [&](int t) {
        if (PG && tid < 2 * COUT) {
            int p0;
            const int bb = cloud_of(t, p0);
            const int g = p0 / a.pool_group;
            const int off = (int)(((unsigned)pch * (unsigned)pgrp + (unsigned)g) * 4u);
            pzn = pwhich ? __builtin_amdgcn_raw_buffer_load_b32(rsrc_Pa(bb), off, 0, 0) : __builtin_amdgcn_raw_buffer_load_b32(rsrc_Pd(bb), off, 0, 0);
        }
    };
    auto write_tile = [&](int buf, int nxt, int tcur) {
        unsigned char* G1 = smem + buf * BUF;
        unsigned char* G2 = G1 + 2 * PL1;
        unsigned char* X2 = G2 + 2 * PL2;
        if (WS && tid < 64) *reinterpret_cast<float4*>(&GS[WS ? tcur & 1 : 0][WS ? tid >> 3 : 0][WS ? 4 * (tid & 7) : 0]) = rgs;
        // act(X) first: its temporaries are dead before the 2 GC values of dY come to life
        if (RED && DB) {
#pragma unroll
            for (int i = 0; i < 8; ++i) rxk[(RED && DB) ? (tcur & 1) : 0][i] = rx[0][i];
        }
#pragma unroll
        for (int q = 0; q < NPX; ++q) {
            const int row = xr0 + q * XRP;
            const float sc = cfX[0][row], sh = cfX[1][row];
            if (RED && !DB) {
                float4* dst = reinterpret_cast<float4*>(XR + row * XRS + xq * 8);
                dst[0] = make_float4(rx[q][0], rx[q][1], rx[q][2], rx[q][3]);
                dst[1] = make_float4(rx[q][4], rx[q][5], rx[q][6], rx[q][7]);
            }
            unsigned q0[4], q1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                split_pair_h(fmaxf(__builtin_fmaf(rx[q][2 * j], sc, sh), 0.f), fmaxf(__builtin_fmaf(rx[q][2 * j + 1], sc, sh), 0.f),
                             q0[j], q1[j]);
            const int off = rows_off<BP>(row, xq * 8);
            *reinterpret_cast<uint4*>(X2 + off) = make_uint4(q0[0], q0[1], q0[2], q0[3]);
            *reinterpret_cast<uint4*>(X2 + PLX + off) = make_uint4(q1[0], q1[1], q1[2], q1[3]);
        }
        if (EARLY) load_x(nxt);
        float v[GC];
        int p0cur;
        (void)cloud_of(tcur, p0cur);
        const int kin = PG ? (p0cur + gp) % a.pool_group : rkin;
#pragma unroll
        for (int i = 0; i < GC; ++i) {
            const int co = GC * gg + i;
            float dz = rz[i];
            if (PG) dz = ((int)poolv[PG ? tcur & 1 : 0][1][PG ? co : 0] == kin) ? __uint_as_float(poolv[PG ? tcur & 1 : 0][0][PG ? co : 0]) : 0.f;
            else if (POOL) dz = (ra[POOL ? i : 0] == kin) ? dz : 0.f;
            v[i] = pro_apply<PRO_BN_BWD>(dz, ry[i], cfG[0][co], cfG[1][co], cfG[2][co], cfG[3][co]);
        }
        if (PG) {                                              // hand the next tile's pair over, fetch the one after
            if (tid < 2 * COUT) poolv[PG ? (tcur + 1) & 1 : 0][PG ? pwhich : 0][PG ? pch : 0] = pzn;
            load_pool(min(tcur + 2, ntile - 1));
        }
        if (EARLY) load_gy(nxt);
#pragma unroll
        for (int h = 0; h < GC / 8; ++h) {
            unsigned p0[4], p1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) split_pair_h(v[8 * h + 2 * j], v[8 * h + 2 * j + 1], p0[j], p1[j]);
            const int off = gp * RB1 + ((GC / 8) * gg + h) * 16;
            *reinterpret_cast<uint4*>(G1 + off) = make_uint4(p0[0], p0[1], p0[2], p0[3]);
            *reinterpret_cast<uint4*>(G1 + PL1 + off) = make_uint4(p1[0], p1[1], p1[2], p1[3]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int co = GC * gg + 8 * h + 2 * j;
                const int o0 = rows_off<BP>(co, gp), o1 = rows_off<BP>(co + 1, gp);
                *reinterpret_cast<unsigned short*>(G2 + o0) = (unsigned short)(p0[j] & 0xffffu);
                *reinterpret_cast<unsigned short*>(G2 + o1) = (unsigned short)(p0[j] >> 16);
                *reinterpret_cast<unsigned short*>(G2 + PL2 + o0) = (unsigned short)(p1[j] & 0xffffu);
                *reinterpret_cast<unsigned short*>(G2 + PL2 + o1) = (unsigned short)(p1[j] >> 16);
            }
        }
        if (!EARLY) { load_gy(nxt); load_x(nxt); }
    };

    // RED: the thread's channel is its X row xr0; what it reads of XR it alone overwrites (in write_tile), and DX is
    // rewritten only behind the next tile's first barrier: no barrier of its own
    const float rsc = RED ? a.xcoef[xr0] : 0.f, rsh = RED ? a.xcoef[CIN + xr0] : 0.f;
    const float rmu = RED ? a.xcoef[2 * CIN + xr0] : 0.f, ris = RED ? a.xcoef[3 * CIN + xr0] : 0.f;
    // tp = the tile the pass is for (DB: selects the dX buffer and the register copy of X; with a.gsum: the tile's place in
    // its run of tpg tiles = one neighbourhood, whose sums are written behind the run's last tile)
    auto red_pass = [&](int tp) {
        const float* dbase = DX + ((RED && DB) ? (tp & 1) * CIN * XRS : 0) + xr0 * XRS + xq * 8;
        const float* xbase_l = XR + xr0 * XRS + xq * 8;
#pragma unroll
        for (int h = 0; h < 2; ++h) {                          // four positions at a time (registers)
            const float4 d4 = reinterpret_cast<const float4*>(dbase)[h];
            float4 x4;
            if (RED && DB) {
                const int k = (RED && DB) ? (tp & 1) : 0;
                x4 = make_float4(rxk[k][4 * h], rxk[k][4 * h + 1], rxk[k][4 * h + 2], rxk[k][4 * h + 3]);
            } else {
                x4 = reinterpret_cast<const float4*>(xbase_l)[h];
            }
            const float xv[4] = {x4.x, x4.y, x4.z, x4.w}, dv[4] = {d4.x, d4.y, d4.z, d4.w};
            float dm[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float d = (__builtin_fmaf(xv[i], rsc, rsh) > 0.f) ? dv[i] : 0.f;
                dm[i] = d;
                s1 += d;
                s2 = __builtin_fmaf(d, (xv[i] - rmu) * ris, s2);
                mx = fmaxf(mx, fabsf(d));
                if (POOL) { gd += d; gy += xv[i]; }
            }
            if (WS && tp >= 0) {                               // (tp < 0: the zero tile in front of the first one)
                const ws_f32x2 d01 = {dm[0], dm[1]}, d23 = {dm[2], dm[3]};
                const ws_f32x2 x01 = {xv[0] - rmu, xv[1] - rmu}, x23 = {xv[2] - rmu, xv[3] - rmu};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float4 g4 = *reinterpret_cast<const float4*>(&GS[WS ? tp & 1 : 0][WS ? j : 0][WS ? xq * 8 + 4 * h : 0]);
                    const float gv[4] = {g4.x, g4.y, g4.z, g4.w};
                    const ws_f32x2 g01 = {g4.x, g4.y}, g23 = {g4.z, g4.w};
                    wSa[WS ? j : 0] = __builtin_elementwise_fma(d01, g01, wSa[WS ? j : 0]);        // v_pk_fma_f32
                    wSa[WS ? j : 0] = __builtin_elementwise_fma(d23, g23, wSa[WS ? j : 0]);
                    wSb[WS ? j : 0] = __builtin_elementwise_fma(x01, g01, wSb[WS ? j : 0]);
                    wSb[WS ? j : 0] = __builtin_elementwise_fma(x23, g23, wSb[WS ? j : 0]);
                    if (xr0 == j) wS3 += (gv[0] + gv[1]) + (gv[2] + gv[3]);
                }
            }
        }
        if (POOL && a.gsum && tp >= 0 && (tp % tpg) == tpg - 1) {
            // the XPC threads of a channel are neighbouring lanes: the neighbourhood's sums, written by the first of them
            float u = gd, v = gy;
#pragma unroll
            for (int off = 1; off < XPC; off <<= 1) { u += __shfl_xor(u, off); v += __shfl_xor(v, off); }
            if (xq == 0) {
                int p0;
                const int bb = cloud_of(tp, p0);
                const int ngrp = a.P / a.pool_group;
                const long long rowid = (long long)bb * CIN + xr0;
                a.gsum[rowid * ngrp + p0 / a.pool_group] = u;
                a.gsum[((long long)a.nb * CIN + rowid) * ngrp + p0 / a.pool_group] = v;
            }
            gd = 0.f; gy = 0.f;
        }
    };

    if (ntile > 0) { load_gy(0); load_x(0); }
    if (PG && ntile > 0) {
        load_pool(0);
        if (tid < 2 * COUT) poolv[0][PG ? pwhich : 0][PG ? pch : 0] = pzn;
        load_pool(min(1, ntile - 1));
    }
    if (RED && !DB) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { XR[xr0 * XRS + xq * 8 + i] = 0.f; DX[xr0 * XRS + xq * 8 + i] = 0.f; }
    }
    __syncthreads();                                           // coefficients are in LDS
    auto run = [&](auto dxr, auto ndwr, const int dw_first) {
    constexpr bool DXR = decltype(dxr)::value;                 // this wave may own a dX tile (then: if wave < NDX)
    constexpr int NDWR = decltype(ndwr)::value;                // dW tiles of this wave: dw_first .. dw_first + NDWR - 1
    static_assert(NDWR <= NCI || NDWR % NCI == 0, "a wave's dW tiles share a co tile, or cover whole co rows");
    const bool does_dx = DXR && ((NDX == NW) || wave < NDX);
    bf16x8 fa[DXR ? KS : 1][2];
    if (DXR) {
        const int row = dx_ci * 32 + c;
        const int chunk = row * 2 + ((kh ^ (row >> 3)) & 1);
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) fa[DXR ? s : 0][pl] = __builtin_bit_cast(bf16x8, a.planes[(long long)(s * 2 + pl) * 256 + chunk]);
    }
    f32x16 acc_dw[NDWR > 0 ? NDWR : 1];
#pragma unroll
    for (int u = 0; u < (NDWR > 0 ? NDWR : 1); ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_dw[u][r] = 0.f;
    // the two products of one tile from LDS buffer `buf`; returns the tile of dX in acc
    auto multiply = [&](int buf, f32x16& acc) {
        const unsigned char* G1 = smem + buf * BUF;
        const unsigned char* G2 = G1 + 2 * PL1;
        const unsigned char* X2 = G2 + 2 * PL2;
        if (DXR && does_dx) {
            const int pos = dx_pt * 32 + c;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int off = pos * RB1 + (2 * s + kh) * 16;
                const f16x8 b0 = *reinterpret_cast<const f16x8*>(G1 + off);
                const f16x8 b1 = *reinterpret_cast<const f16x8*>(G1 + PL1 + off);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[DXR ? s : 0][0]), b1, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[DXR ? s : 0][1]), b0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[DXR ? s : 0][0]), b0, acc, 0, 0, 0);
                if (KS == 8 && (s & 1)) __builtin_amdgcn_sched_barrier(0);     // keeps hipcc from hoisting all 16 fragment reads (spills)
            }
        }
        // dW[co][ci] += sum_p dY[co][p] act(X)[ci][p]
#pragma unroll
        for (int s = 0; s < PS; ++s) {
            f16x8 a0, a1;
#pragma unroll
            for (int u = 0; u < NDWR; ++u) {
                const int T = dw_first + u;
                if ((NDWR <= NCI) ? (u == 0) : (u % NCI == 0)) {           // a new co tile: its dY^T fragments
                    const int offA = rows_off<BP>((T / NCI) * 32 + c, (2 * s + kh) * 8);
                    a0 = *reinterpret_cast<const f16x8*>(G2 + offA);
                    a1 = *reinterpret_cast<const f16x8*>(G2 + PL2 + offA);
                }
                const int offB = rows_off<BP>((T % NCI) * 32 + c, (2 * s + kh) * 8);
                const f16x8 b0 = *reinterpret_cast<const f16x8*>(X2 + offB);
                const f16x8 b1 = *reinterpret_cast<const f16x8*>(X2 + PLX + offB);
                acc_dw[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc_dw[u], 0, 0, 0);
                acc_dw[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc_dw[u], 0, 0, 0);
                acc_dw[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc_dw[u], 0, 0, 0);
            }
        }
    };
    // lane = position pos of the tile; register r = input channel 32 dx_ci + 8 (r >> 2) + 4 kh + (r & 3): a store
    // instruction writes 32 consecutive positions of two rows
    auto store_dx = [&](int t, const f32x16& acc) {
        if (!(DXR && does_dx)) return;
        const int pos = dx_pt * 32 + c;
        int p0;
        const int bb = cloud_of(t, p0);
        float* orow = a.dX + (long long)bb * a.dx_rows * a.P + p0 + pos;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = dx_ci * 32 + 8 * (r >> 2) + 4 * kh + (r & 3);
            const float v = acc[r] * dx_scale;
            st_out<ST_LAYER_BWD_DX>(orow + (long long)ci * a.P, v);
            if (RED) DX[((RED && DB) ? (t & 1) * CIN * XRS : 0) + ci * XRS + pos] = v;
        }
    };
    if (DB) {
        if (ntile > 0) write_tile(0, min(1, ntile - 1), 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        int cur = 0;
        for (int t = 0; t < ntile; ++t) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            if (RED && t > 0) red_pass(t - 1);                 // (the dX buffer and the register copy of X of the previous tile)
            multiply(cur, acc);
            if (t + 1 < ntile) write_tile(cur ^ 1, min(t + 2, ntile - 1), t + 1);   // the next tile, while the MFMAs drain
            store_dx(t, acc);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                      // (raw: the loads stay in flight across it)
            cur ^= 1;
        }
    } else {
        for (int t = 0; t < ntile; ++t) {
            if (RED) red_pass(t - 1);                          // the previous tile (the first time: the zeros written above)
            write_tile(0, min(t + 1, ntile - 1), t);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                      // (raw: the loads stay in flight across it)
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            multiply(0, acc);
            store_dx(t, acc);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                      // everyone is done reading this tile's LDS
        }
    }
    {
        float* out = a.part + (long long)blockIdx.x * COUT * CIN;
#pragma unroll
        for (int u = 0; u < NDWR; ++u) {
            const int T = dw_first + u;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (T / NCI) * 32 + 8 * (r >> 2) + 4 * kh + (r & 3);
                st_out<ST_WGRAD_PART>(out + row * CIN + (T % NCI) * 32 + c, acc_dw[u][r] * dw_scale);
            }
        }
    }
    };   // run
    if constexpr (SPLIT) {
        constexpr int HALF = NW / 2, NDWS = NCO * NCI / HALF;
        static_assert(NDX == HALF, "SPLIT: one dX tile per wave of the first half");
        if (wave < HALF) run(std::true_type{}, std::integral_constant<int, 0>{}, 0);
        else run(std::false_type{}, std::integral_constant<int, NDWS>{}, (wave - HALF) * NDWS);
    } else {
        run(std::true_type{}, std::integral_constant<int, NDW>{}, wave * NDW);
    }
    if (RED) {
        if (ntile > 0) red_pass(ntile - 1);                    // the last tile (behind the loop's closing barrier)
        // the XPC threads of a channel are neighbouring lanes; one partial per channel and workgroup
#pragma unroll
        for (int off = 1; off < XPC; off <<= 1) {
            s1 += __shfl_xor(s1, off);
            s2 += __shfl_xor(s2, off);
            mx = fmaxf(mx, __shfl_xor(mx, off));
        }
        const long long nblk = G;
        if (xq == 0) {
            a.red[(long long)blockIdx.x * CIN + xr0] = s1;
            a.red[(nblk + blockIdx.x) * CIN + xr0] = s2;
        }
        if (WS) {
            float wS1[8], wS2[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                wS1[j] = wSa[WS ? j : 0].x + wSa[WS ? j : 0].y; wS2[j] = wSb[WS ? j : 0].x + wSb[WS ? j : 0].y;
#pragma unroll
                for (int off = 1; off < XPC; off <<= 1) {
                    wS1[j] += __shfl_xor(wS1[j], off);
                    wS2[j] += __shfl_xor(wS2[j], off);
                }
            }
#pragma unroll
            for (int off = 1; off < XPC; off <<= 1) wS3 += __shfl_xor(wS3, off);
            if (xq == 0) {
                float* o = a.wsum + ((long long)blockIdx.x * CIN + xr0) * 16;
#pragma unroll
                for (int j = 0; j < 8; ++j) { o[j] = wS1[j]; o[8 + j] = wS2[j]; }
                if (xr0 < 8) a.wsum3[(long long)blockIdx.x * 8 + xr0] = wS3;
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        __syncthreads();
        float* rs = reinterpret_cast<float*>(smem);
        if (lane == 0) rs[wave] = mx;
        __syncthreads();
        if (tid == 0) {
            float m = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) m = fmaxf(m, rs[w]);
            a.red[2 * nblk * CIN + blockIdx.x] = m;
        }
    }
}

template <int CIN, int COUT, bool POOL, bool RED, int NW, int BP, bool DB, bool PG = false, bool SPLIT = false>
__global__ __launch_bounds__(64 * NW, 2) void layer_bwd_x2_kernel(const LayerBwdArgs a)
{
    layer_bwd_x2_body<CIN, COUT, POOL, RED, NW, BP, DB, PG, SPLIT, false>(a);
}

// ... and with the producing layer's weight-gradient sums taken on the way (LayerBwdArgs::wsrc)
template <int CIN, int COUT, int NW, int BP>
__global__ __launch_bounds__(64 * NW, 2) void layer_bwd_x2ws_kernel(const LayerBwdArgs a)
{
    layer_bwd_x2_body<CIN, COUT, false, true, NW, BP, false, false, false, true>(a);
}

// WS: dW'[c][j] = a1' S1 + q1' S2 + (q1' mu' + q0') S3 from the per-workgroup sums (see LayerBwdArgs).  One workgroup per
// channel; thread -> (source row j = tid % 8, slice tid / 8 of the workgroups); fp64, fixed order.
__global__ __launch_bounds__(256) void wsum_finalize_kernel(
    const float* __restrict__ wsum, const float* __restrict__ wsum3, int blocks, int C, const float* __restrict__ coef4,
    const float* __restrict__ mean, int ws_rows, float* __restrict__ dW, int lddw)
{
    __shared__ double red[3][32][8];
    const int c = blockIdx.x, j = threadIdx.x & 7, sl = threadIdx.x >> 3;
    const int per = (blocks + 31) / 32, b0 = sl * per, b1 = min(blocks, b0 + per);
    double s1 = 0, s2 = 0, s3 = 0;
    int b = b0;
    for (; b + 7 < b1; b += 8) {                               // 24 loads in flight (one at a time: a chain of L2 round trips, 10 us)
        float u[8], v[8], w[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float* o = wsum + ((long long)(b + i) * C + c) * 16;
            u[i] = o[j]; v[i] = o[8 + j]; w[i] = wsum3[(long long)(b + i) * 8 + j];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) { s1 += (double)u[i]; s2 += (double)v[i]; s3 += (double)w[i]; }
    }
    for (; b < b1; ++b) {
        const float* o = wsum + ((long long)b * C + c) * 16;
        s1 += (double)o[j]; s2 += (double)o[8 + j]; s3 += (double)wsum3[(long long)b * 8 + j];
    }
    red[0][sl][j] = s1; red[1][sl][j] = s2; red[2][sl][j] = s3;
    __syncthreads();
    if (threadIdx.x < 8 && j < ws_rows) {
        double t1 = 0, t2 = 0, t3 = 0;
#pragma unroll
        for (int g = 0; g < 32; ++g) { t1 += red[0][g][j]; t2 += red[1][g][j]; t3 += red[2][g][j]; }
        const float a1 = coef4[c], q1 = coef4[2 * C + c], q0 = coef4[3 * C + c], mu = mean[c];
        dW[(long long)c * lddw + j] = (float)((double)a1 * t1 + (double)q1 * t2 + (double)__builtin_fmaf(q1, mu, q0) * t3);
    }
}

// Workgroups: as many as the chip holds AT ONCE (two 4-wave workgroups per CU, one 8-wave workgroup for the 128 x 128
// form), so that the launch is a single round; every workgroup takes every G-th 32-position tile.
int layer_bwd_blocks(int Cin, int Cout, int P, int nb)
{
    const long long slots = (Cin == 128 && Cout == 128) ? 256 : 512;
    const long long total = (long long)nb * (P / 32);
    return (int)(total < slots ? total : slots);
}

}  // namespace

// 1 when usip_mlp_layer_backward_x2h_f32 takes the shape: (Cin, Cout) = (64, 64), or (128, 128) in the pooled form (dZ
// synthesised from dpooled / arg), or (64, 128) (the feature half of conv4); positions a multiple of 64.  (The plain
// (128, 128) form has no caller.)  The (64, 128) form was withdrawn for a day: built WITH hipcc's SLP vectoriser it
// produced 1-4 slightly wrong tiles of 1024 per launch whenever two of its workgroups shared a CU -- a packed
// `v_pk_fma_f32` with swapped halves (op_sel) evaluating c2 * y + c3 returned c3 in its low half for lanes 48-63.  The
// library is now compiled with -fno-slp-vectorize (usip_amd/build.py); tools/layer_bwd_race.py and
// test_fused_layer_backward_is_bit_stable_over_repeated_launches keep watch.
extern "C" int usip_mlp_layer_backward_x2h_supported(int Cin, int Cout, int P, int pooled)
{
    if (P <= 0 || P % 64 != 0 || (long long)Cout * P >= (1LL << 30) || (long long)Cin * P >= (1LL << 30)) return 0;
    if (pooled) return (Cin == 128 && Cout == 128) ? 1 : 0;
    return (Cin == 64 && (Cout == 64 || Cout == 128)) ? 1 : 0;
}

extern "C" int usip_mlp_layer_backward_x2h_blocks(int Cin, int Cout, int P, int nb)
{
    return layer_bwd_blocks(Cin, Cout, P, nb);
}

extern "C" long long usip_mlp_layer_backward_x2h_workspace(int Cin, int Cout, int P, int nb)
{
    return (long long)usip_mlp_layer_backward_x2h_blocks(Cin, Cout, P, nb) * Cout * Cin;
}

// dX[b][ci][p] = sum_co W[co][ci] dY[b][co][p]  and  dW[co * lddw + ci] = sum_{b,p} dY[b][co][p] act(X)[b][ci][p]  with
// f32x2 products; dY = BatchNorm'(ReLU'(dZ)) rebuilt from (dZ, Y, coef4) -- coef4 = the [5][Cout] array
// usip_bn_backward_reduce_f32 / usip_bn_backward_finalize_f32 write with want_bound -- or, when pool_dp / pool_arg are
// given (dZ = NULL), from dZ[b][co][p] = (p % pool_group == pool_arg[b][co][p / pool_group]) ? pool_dp[b][co][p /
// pool_group] : 0; act(X) = relu(X xcoef[0] + xcoef[1]), xcoef = the PRODUCING layer's [4][Cin] (scale, shift, mean,
// invstd) of a training-mode BatchNorm over exactly these nb * P samples.  group_sums (pooled form with red_partial, may be
// NULL; pool_group a multiple of 32): [2][nb * Cin][P / pool_group], per neighbourhood of the PRODUCING layer sum_k dX [relu
// on] and sum_k X -- what usip_bn_backward_reduce_f32 leaves in `gsum` for a pooled-concat layer.  planes: usip_mlp_split2h_f32 image of W as
// the data-gradient operand (At = W [Cout][ldw] K-major, M = Cin, K = Cout).  X / dX point at the first of the Cin rows
// inside [nb][x_rows][P] / [nb][dx_rows][P]; all pointers 16-B aligned.  workspace:
// usip_mlp_layer_backward_x2h_workspace floats.  red_partial (may be NULL): receives [2][blocks][Cin] partial sums of the
// producing layer's BatchNorm backward against dX, then [blocks] maxima of |dX [relu on]|
// (usip_bn_backward_finalize_f32 with a maxima pointer turns them into that layer's [5][Cin] coef4).
static int layer_backward_x2h(const float* dZ, const float* Y, const float* coef4, const float* pool_dp,
                              const int32_t* pool_arg, int pool_group, const float* X, int x_rows,
                              const float* xcoef, const void* planes, float* dX, int dx_rows,
                              float* workspace, float* dW, int lddw, float* red_partial,
                              float* group_sums, int Cin, int Cout, int P, int nb, const float* wsrc, int ws_rows,
                              float* wsum, float* wsum3, void* stream)
{
    const bool pooled = pool_dp != nullptr;
    if (!usip_mlp_layer_backward_x2h_supported(Cin, Cout, P, pooled ? 1 : 0) || nb < 1 || x_rows < Cin || dx_rows < Cin ||
        lddw < Cin)
        return USIP_EINVAL;
    if ((!pooled && !dZ) || !Y || !coef4 || !X || !xcoef || !planes || !dX || !workspace || !dW) return USIP_EINVAL;
    if (group_sums && (!pooled || !red_partial || pool_group % 32 != 0)) return USIP_EINVAL;
    if (pooled && (!pool_arg || pool_group < 1 || P % pool_group != 0)) return USIP_EINVAL;
    if ((reinterpret_cast<uintptr_t>(dZ) | reinterpret_cast<uintptr_t>(Y) | reinterpret_cast<uintptr_t>(X) |
         reinterpret_cast<uintptr_t>(dX) | reinterpret_cast<uintptr_t>(planes)) & 15u)
        return USIP_EINVAL;
    const int blocks = layer_bwd_blocks(Cin, Cout, P, nb);
    if (wsrc && (!wsum || !wsum3 || !red_partial || pooled || Cin != 64 || Cout != 64 || ws_rows < 1 || ws_rows > 8 ||
                 (reinterpret_cast<uintptr_t>(wsrc) & 15u)))
        return USIP_EINVAL;
    LayerBwdArgs a{dZ, Y, coef4, pool_dp, pool_arg, pool_group, X, x_rows, xcoef, reinterpret_cast<const uint4*>(planes),
                   dX, dx_rows, workspace, red_partial, group_sums, P, nb, wsrc, ws_rows, wsum, wsum3};
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)blocks);
    const bool red = red_partial != nullptr;
    if (Cin == 64 && Cout == 128) {
        // the feature half of conv4: two waves for dX (with the 64 registers of weight fragments), two for dW (with its 64)
        if (pooled) return USIP_EINVAL;
        if (red) USIP_LAUNCH((layer_bwd_x2_kernel<64, 128, false, true, 4, 32, false, false, true>), grid, dim3(256), 0, st, a);
        else USIP_LAUNCH((layer_bwd_x2_kernel<64, 128, false, false, 4, 32, false, false, true>), grid, dim3(256), 0, st, a);
    } else if (Cin == 64) {
        if (red && wsrc) USIP_LAUNCH((layer_bwd_x2ws_kernel<64, 64, 4, 32>), grid, dim3(256), 0, st, a);
        else if (red) USIP_LAUNCH((layer_bwd_x2_kernel<64, 64, false, true, 4, 32, false>), grid, dim3(256), 0, st, a);
        else USIP_LAUNCH((layer_bwd_x2_kernel<64, 64, false, false, 4, 32, false>), grid, dim3(256), 0, st, a);
    } else {
        // (the single-buffer form with 64-position tiles measured the same, 224-230 us at 16 x 32768 positions, and
        // sits on the edge of spilling: 252-256 VGPRs)
        if (red && pool_group % 32 != 0) return USIP_EINVAL;
        if (red)
            USIP_LAUNCH((layer_bwd_x2_kernel<128, 128, true, true, 8, 32, false, true, true>), grid, dim3(512), 0, st, a);
        else if (pool_group % 32 == 0)
            USIP_LAUNCH((layer_bwd_x2_kernel<128, 128, true, false, 8, 32, true, true, true>), grid, dim3(512), 0, st, a);
        else
            USIP_LAUNCH((layer_bwd_x2_kernel<128, 128, true, false, 8, 32, true, false>), grid, dim3(512), 0, st, a);
    }
    USIP_LAUNCH_CHECK();
    return usip_mlp::launch_wgrad_reduce(workspace, dW, (long long)Cout * Cin, blocks, Cin, lddw, 0, st);
}

extern "C" int usip_mlp_layer_backward_x2h_f32(const float* dZ, const float* Y, const float* coef4, const float* pool_dp,
                                               const int32_t* pool_arg, int pool_group, const float* X, int x_rows,
                                               const float* xcoef, const void* planes, float* dX, int dx_rows,
                                               float* workspace, float* dW, int lddw, float* red_partial,
                                               float* group_sums, int Cin, int Cout, int P, int nb, void* stream)
{
    return layer_backward_x2h(dZ, Y, coef4, pool_dp, pool_arg, pool_group, X, x_rows, xcoef, planes, dX, dx_rows, workspace, dW,
                              lddw, red_partial, group_sums, Cin, Cout, P, nb, nullptr, 0, nullptr, nullptr, stream);
}

// The (64, 64) form with red_partial that ALSO takes the sums from which the PRODUCING layer's weight gradient follows,
// for a producing layer whose input S [nb][ws_rows <= 8][P] needs no gradient (conv1 of RPN_Detector_Ball, the first
// PointNet layer of RPN_Detector: models/networks.py:705, layers.py:524-544): wsum [blocks][64][16], wsum3 [blocks][8]
// (blocks = usip_mlp_layer_backward_x2h_blocks) -- see usip_mlp_wsum_finalize_f32.  That layer then needs no pass of its
// own over its (dZ, Y).
extern "C" int usip_mlp_layer_backward_x2h_ws_f32(const float* dZ, const float* Y, const float* coef4, const float* X,
                                                  int x_rows, const float* xcoef, const void* planes, float* dX,
                                                  int dx_rows, float* workspace, float* dW, int lddw, float* red_partial,
                                                  const float* wsrc, int ws_rows, float* wsum, float* wsum3,
                                                  int Cin, int Cout, int P, int nb, void* stream)
{
    if (!wsrc) return USIP_EINVAL;
    return layer_backward_x2h(dZ, Y, coef4, nullptr, nullptr, 0, X, x_rows, xcoef, planes, dX, dx_rows, workspace, dW, lddw,
                              red_partial, nullptr, Cin, Cout, P, nb, wsrc, ws_rows, wsum, wsum3, stream);
}

// dW[c * lddw + j] (j < ws_rows) = coef4[0][c] S1[c][j] + coef4[2][c] S2[c][j] + (coef4[2][c] mean[c] + coef4[3][c]) S3[j]
// with S* = the sums of usip_mlp_layer_backward_x2h_ws_f32 over its `blocks` workgroups (fp64, fixed order): the weight
// gradient dY . S^T of a training-mode BatchNorm layer with C = 64 channels whose dY = coef4[0] dYhat + coef4[2] y +
// coef4[3] (coef4 = what usip_bn_backward_finalize_*_f32 made of the same call's red_partial; mean = its batch mean).
extern "C" int usip_mlp_wsum_finalize_f32(const float* wsum, const float* wsum3, int blocks, int C, const float* coef4,
                                          const float* mean, int ws_rows, float* dW, int lddw, void* stream)
{
    if (!wsum || !wsum3 || !coef4 || !mean || !dW || blocks < 1 || C < 1 || ws_rows < 1 || ws_rows > 8 || lddw < ws_rows)
        return USIP_EINVAL;
    USIP_LAUNCH(wsum_finalize_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, wsum, wsum3, blocks, C, coef4, mean,
                ws_rows, dW, lddw);
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}
