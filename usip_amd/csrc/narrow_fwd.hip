// usip_amd/csrc/narrow_fwd.hip -- forward of a NARROW shared-MLP layer (64 inputs, 64 or 128 outputs: conv2, conv3 and
// the feature half of conv4 of RPN_Detector_Ball, the PointNet layers of RPN_Detector; models/networks.py:705-709,
// models/layers.py:524-544) as a STREAMING kernel.
//
// These layers are HBM-bound (16 flop/B at 64 x 64) and the generic tile kernel leaves 40 % of the achievable rate on
// the table (3.4 TB/s where an element-wise pass over the same bytes reaches 5.7; tools/copy_probe.py): r02 counters
// show its waves waiting on memory 65 % of the time with the matrix pipe 36 % busy -- a workgroup has ONE 16-KiB
// stage in flight, its output leaves as 16-B pieces of 64 different rows per instruction (4.5 M write requests of
// ~30 B for 134 MB), and the load, multiply and store phases of a tile do not overlap.  Here:
//   * persistent workgroups of four waves, two or three per CU, walk 64-position tiles; the whole K = 64 slab of
//     tile t+2 is copied memory -> LDS by LDS-DMA (global_load_lds_dwordx4, no staging registers) while tile t is
//     multiplied; the workgroups of a CU drift out of phase, so one multiplies while another stores;
//   * the weight fragments live in registers for the workgroup's lifetime (a wave owns 32 or 64 output channels);
//   * the accumulator is D[channel][position] (weights as the MFMA A operand): a lane holds ONE position and 16
//     channels per 32 x 32 tile, so every store instruction writes two full 128-B lines -- no partial sectors;
//   * BatchNorm statistics are per-lane running sums over all the workgroup's tiles, reduced across lanes once at
//     the end: one partial per channel and WORKGROUP (512-768) instead of per tile (2048-4096).
// (First version, measured and replaced: ONE workgroup of eight waves per CU with 256-position tiles.  Its loads and
// stores alone ran at 5.8 TB/s and its multiply alone at 70 % of the matrix pipe, but the eight waves move in lock
// step through load-wait / multiply / store, so the phases added up instead of overlapping: 75-85 us, no better than
// the generic kernel's 79.)
// fp32 in, fp32 accumulate (v_mfma_f32_32x32x2_f32): the arithmetic class of the generic fp32 kernel.
#include "mlp_common.h"

using namespace usip_mlp;

namespace {

constexpr int NF_K = 64;             // input channels (the whole contraction in one slab)
constexpr int NF_T = 256;            // threads: 4 waves = 2 position tiles x 2 channel groups
constexpr int NF_BN = 64;            // positions per tile

struct NarrowFwdArgs {
    const float* At; int lda;        // K-major weights: W[m][k] = At[k * lda + m]
    const float* X;                  // [nb][64][P]
    const float* coef;               // [>=2][64]: X := relu(X * coef[0][k] + coef[1][k]) (PRO) or unused
    const float* bias;               // [M] or null
    const float* rowbias; int rb_group;   // Y += rowbias[b][m][p / rb_group] or null
    float* Y; int y_rows;            // Y[b] = Y + b * y_rows * P, rows [0, M) written
    float* stats;                    // [2][M][G] or null, G = gridDim.x
    int P, nb;
};

template <int N>
__device__ __forceinline__ void wait_vm()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// OTW = output tiles (32 channels) per wave: 1 for M = 64, 2 for M = 128.  Wave w: positions [32 (w & 1), +32) of the
// tile, channels [32 OTW (w >> 1), +32 OTW).
template <int OTW, bool PRO, bool STATS, bool RB>
__global__ __launch_bounds__(NF_T, OTW == 1 ? 3 : 2) void narrow_fwd_kernel(const NarrowFwdArgs a)
{
    constexpr int M = OTW * 64, BN = NF_BN;
    constexpr int NDMA = NF_K * BN * 4 / (NF_T * 16);        // DMA instructions per thread and tile: 4
    constexpr int NST = 16 * OTW;                             // store instructions per thread and tile
    __shared__ __attribute__((aligned(16))) float Xs[2][NF_K][BN];
    __shared__ float2 cf[NF_K];
    __shared__ float red[2][2][M];                            // final statistics: [sum | sum^2][position tile][channel]
    __shared__ float bsh[M];                                  // bias
    __shared__ float rbs[RB ? 4 : 1][64];                     // row bias of the wave's channels in the current tile

    const int tid = threadIdx.x, lane = tid & 63, c = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pcol = wave & 1;
    const int o0 = (wave >> 1) * OTW;                         // first of the wave's output tiles
    const int tpc = (a.P + BN - 1) / BN, total = a.nb * tpc;
    const int G = gridDim.x;

    if (PRO && tid < NF_K) cf[tid] = make_float2(a.coef[tid], a.coef[NF_K + tid]);
    if (tid < M) bsh[tid] = a.bias ? a.bias[tid] : 0.f;
    // weight fragments: the A operand of output tile t at k-step i is W[(o0 + t) * 32 + c][half * 32 + i]
    float wr[OTW][32];
#pragma unroll
    for (int t = 0; t < OTW; ++t)
#pragma unroll
        for (int i = 0; i < 32; ++i) wr[t][i] = a.At[(long long)(half * 32 + i) * a.lda + (o0 + t) * 32 + c];
    // channel of accumulator register q of tile t: (o0 + t) * 32 + 8 * (q / 4) + 4 * half + q % 4

    float s1[STATS ? OTW : 1][16], s2[STATS ? OTW : 1][16];
    if (STATS) {
#pragma unroll
        for (int t = 0; t < OTW; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) { s1[t][q] = 0.f; s2[t][q] = 0.f; }
    }

    // memory -> LDS copy of one tile: a wave instruction moves 1 KiB = four k-rows of 64 positions
    auto dma_tile = [&](int tile, int buf) {
        const int b = tile / tpc, p0 = (tile - b * tpc) * BN;
        const float* xb = a.X + (long long)b * NF_K * a.P;
#pragma unroll
        for (int j = 0; j < NDMA; ++j) {
            const int r4 = (j * 4 + wave) * 4;                // first of the instruction's four rows
            const float* src = xb + (long long)(r4 + (lane >> 4)) * a.P + min(p0 + (lane & 15) * 4, a.P - 4);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)&Xs[buf][r4][0], 16, 0, st_aux<LD_NARROW_FWD>());
        }
    };
    auto wait_newer = [&](bool ragged, bool stores, bool dma) {   // at most the named newer operations stay in flight
        if (ragged) wait_vm<0>();                             // (a ragged tile's masked stores may have been skipped)
        else if (stores && dma) wait_vm<NDMA + NST>();
        else if (stores) wait_vm<NST>();
        else if (dma) wait_vm<NDMA>();
        else wait_vm<0>();
    };

    // tiles of a workgroup: g, g + G, g + 2G, ... (at any moment the chip works on one contiguous stretch of positions)
    const int first = blockIdx.x;
    if (first < total) dma_tile(first, 0);
    if (first + G < total) dma_tile(first + G, 1);
    __syncthreads();                                          // cf, bsh visible (the DMA is waited for below)

    int it = 0;
    bool rag1 = false;                                        // tile it-1 ragged (partly outside the cloud)
    for (int tile = first; tile < total; tile += G, ++it) {
        const int buf = it & 1;
        const int b = tile / tpc, p0 = (tile - b * tpc) * BN;
        // DMA(t) done.  Issued after it (vmcnt retires in issue order): DMA(t+1) and the stores of tile t-1.
        wait_newer(rag1, it >= 1, tile + G < total);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");

        const int p = p0 + pcol * 32 + c;                     // the lane's position
        const bool pok = p < a.P;
        float rbv = 0.f;
        if (RB) {
            // the wave's 32 positions lie in ONE group (rb_group % 32 == 0): lane l fetches the value of channel
            // o0*32 + l; it goes through LDS to the lanes that hold that channel after the multiply, where its
            // latency hides
            const int ngrp = a.P / a.rb_group, grp = min(p0 + pcol * 32, a.P - 1) / a.rb_group;
            rbv = a.rowbias[((long long)b * M + o0 * 32 + (lane & (32 * OTW - 1))) * ngrp + grp];
        }
        f32x16 acc[OTW];
#pragma unroll
        for (int t = 0; t < OTW; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[t][q] = bsh[(o0 + t) * 32 + 8 * (q >> 2) + 4 * half + (q & 3)];
        const float* xcol = &Xs[buf][half * 32][pcol * 32 + c];
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            float x = xcol[i * BN];                           // B operand: X[k = half*32 + i][position]
            if (PRO) {
                const float2 k2 = cf[half * 32 + i];
                x = fmaxf(__builtin_fmaf(x, k2.x, k2.y), 0.0f);
            }
#pragma unroll
            for (int t = 0; t < OTW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[t][i], x, acc[t], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                         // every wave is done with Xs[buf]
        const bool issue = tile + 2 * G < total;
        if (issue) dma_tile(tile + 2 * G, buf);

        if (RB) {                                             // newer than the row-bias load: that DMA
            if (issue) wait_vm<NDMA>(); else wait_vm<0>();
            rbs[wave][lane] = rbv;                            // wave-private: no barrier
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int t = 0; t < OTW; ++t)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[t][q] += rbs[wave][t * 32 + 8 * (q >> 2) + 4 * half + (q & 3)];
        }
        // 32-bit element offset: a uniform row base + one VGPR offset per store (the host checks the size of Y)
        const unsigned yo = (unsigned)(((long long)b * a.y_rows + o0 * 32 + 4 * half) * a.P + min(p, a.P - 1));
        rag1 = p0 + BN > a.P;
        if (!rag1) {                                          // whole tile inside (uniform): no per-store predicate
#pragma unroll
            for (int t = 0; t < OTW; ++t)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const float v = acc[t][q];
                    float* rowbase = a.Y + (size_t)(t * 32 + 8 * (q >> 2) + (q & 3)) * a.P;       // wave-uniform
                    st_out<ST_NARROW_FWD>(rowbase + yo, v);
                    if (STATS) { s1[t][q] += v; s2[t][q] = __builtin_fmaf(v, v, s2[t][q]); }
                }
        } else {
#pragma unroll
            for (int t = 0; t < OTW; ++t)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const float v = acc[t][q];
                    float* rowbase = a.Y + (size_t)(t * 32 + 8 * (q >> 2) + (q & 3)) * a.P;
                    if (pok) {
                        st_out<ST_NARROW_FWD>(rowbase + yo, v);
                        if (STATS) { s1[t][q] += v; s2[t][q] = __builtin_fmaf(v, v, s2[t][q]); }
                    }
                }
        }
    }

    if (STATS) {
        // per channel: the 32 positions of a half-wave -> lane c == 0 (two 16-lane DPP row sums and one exchange),
        // then the two waves that share the channels
#pragma unroll
        for (int t = 0; t < OTW; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                float u = usip_row16_sum(s1[t][q]), v = usip_row16_sum(s2[t][q]);
                u += __shfl_xor(u, 16);
                v += __shfl_xor(v, 16);
                if (c == 0) {
                    const int ch = (o0 + t) * 32 + 8 * (q >> 2) + 4 * half + (q & 3);
                    red[0][pcol][ch] = u;
                    red[1][pcol][ch] = v;
                }
            }
        __syncthreads();
        if (tid < M) {
            a.stats[(long long)tid * G + blockIdx.x] = red[0][0][tid] + red[0][1][tid];
            a.stats[(long long)G * M + (long long)tid * G + blockIdx.x] = red[1][0][tid] + red[1][1][tid];
        }
    }
}

}  // namespace

// Workgroups (= statistics partials per channel) the kernel uses for this shape, 0 when the shape is not its own.
extern "C" int usip_mlp_narrow_forward_blocks(int M, int K, int P, int nb)
{
    if (K != NF_K || (M != 64 && M != 128) || P < 4 || P % 4 != 0 || nb < 1) return 0;
    const long long total = (long long)nb * ((P + NF_BN - 1) / NF_BN);
    int G = (M == 64) ? 768 : 512;                           // three / two workgroups per CU
    if (usip_tuning_value(USIP_TUNE_R5_FORMS) & 64) G = (M == 64) ? 1024 : 768;   // measurement: one more per CU (round 6)
    if (total < 4 * G) return 0;                             // too few tiles to keep the persistent workgroups busy
    return G;
}

// Y = W . act(X) + bias (+ rowbias), optional BatchNorm partials; the contract of usip_mlp_gemm_f32 for K = 64,
// M in {64, 128}, P % 4 == 0, 16-B aligned X.  stats: [2][M][usip_mlp_narrow_forward_blocks(...)].
extern "C" int usip_mlp_narrow_forward_f32(const float* At, int lda, const float* X, const float* coef, int pro,
                                           const float* bias, const float* rowbias, int rb_group, float* Y,
                                           int y_rows, float* stats, int M, int K, int P, int nb, void* stream)
{
    const int G = usip_mlp_narrow_forward_blocks(M, K, P, nb);
    if (G == 0 || !At || !X || !Y || lda < M || (pro != 0 && pro != 1) || (pro == 1 && !coef)) return USIP_EINVAL;
    if ((reinterpret_cast<uintptr_t>(X) & 15u) || (rowbias && (rb_group < 32 || rb_group % 32 != 0 || P % rb_group != 0)))
        return USIP_EINVAL;
    if (y_rows == 0) y_rows = M;
    if (y_rows < M || (long long)nb * y_rows * P >= (1LL << 32)) return USIP_EINVAL;    // 32-bit element offsets into Y
    NarrowFwdArgs a{At, lda, X, coef, bias, rowbias, rb_group, Y, y_rows, stats, P, nb};
    hipStream_t st = (hipStream_t)stream;
#define USIP_NF(OTW_, PRO_, ST_)                                                                      \
    do {                                                                                              \
        if (rowbias) USIP_LAUNCH((narrow_fwd_kernel<OTW_, PRO_, ST_, true>), dim3(G), dim3(NF_T), 0, st, a);  \
        else USIP_LAUNCH((narrow_fwd_kernel<OTW_, PRO_, ST_, false>), dim3(G), dim3(NF_T), 0, st, a);         \
    } while (0)
    const bool p1 = pro == 1, s = stats != nullptr;
    if (M == 64) {
        if (p1) { if (s) USIP_NF(1, true, true); else USIP_NF(1, true, false); }
        else    { if (s) USIP_NF(1, false, true); else USIP_NF(1, false, false); }
    } else {
        if (p1) { if (s) USIP_NF(2, true, true); else USIP_NF(2, true, false); }
        else    { if (s) USIP_NF(2, false, true); else USIP_NF(2, false, false); }
    }
#undef USIP_NF
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}
