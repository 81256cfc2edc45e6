// usip_amd/csrc/gemm_x2f.hip -- round 6: the f32x2 forward / data-gradient GEMM of the 256..640-wide shared-MLP layers
// (models/layers.py:208-216, :293-303, :401-440) with ONE wave per SIMD and a 64-position wave tile ("fat" wave).
//
// Why.  gemm_x2d.hip runs two 4-wave workgroups per CU; a wave owns 32 positions x 256 channels (8 accumulator tiles = 128
// AGPRs -- all two co-resident waves can have).  Per 24 MFMAs such a wave issues 16 weight-fragment reads, 8 operand loads
// and 2 LDS-DMA pieces, and two waves on a SIMD mostly fill each other's gaps instead of running side by side
// (DESIGN.md 5 "Round 4").  tools/probes/mfma_lone_clock.hip (profiles/r06a_lone_wave_probe.txt) measured the alternative on the
// skeleton of a stage: ONE wave per SIMD with all 512 registers -- 16 accumulator tiles = 64 positions x 256 channels, 48 MFMAs
// per stage for the same 16 fragment reads, 8 loads (dwordx2) and 4 DMA pieces -- sustains the rate of bare MFMAs (1840 TFLOP/s
// of fp16 products; today's stage skeleton 1600, today's kernel 1150 inside its loop).  The chip holds ~2 GHz for the lone
// wave (0.9-1.2 with two), so the ~37 cycles it needs per MFMA with its fillers cost nothing in time.
//
// What.  Tile 256 channels x 256 positions, one 4-wave workgroup per CU, persistent.  Wave w owns positions [64 w, 64 w + 64)
// of the tile as TWO 32-position blocks: block 0 = the EVEN positions, block 1 = the ODD ones -- so lane (c = l & 31,
// h = l >> 5) loads X[k = 16 kt + 8 h + i][pw + 2 c .. 2 c + 1] with ONE buffer_load_dwordx2 per k (a half-wave reads 256
// contiguous bytes of a row) and the two values are the lane's MFMA operand elements of the two blocks: no transposition, half
// the vector-memory instructions per element of gemm_x2d.hip.  Prologue, two-plane fp16 split, scales, plane products
// (X_hi.A_lo, X_lo.A_hi, X_hi.A_hi, smallest first) and their order per accumulator are gemm_x2d.hip's: OUTPUTS ARE BIT-IDENTICAL
// to it (tests/test_f32x2_mode_gpu.py); the statistics differ by fp32 summation order inside a tile.
//   * a weight fragment (lo / hi plane of one 32-channel tile, 8 registers) feeds SIX MFMAs (both blocks): 16 ds_read_b128 per
//     48 MFMAs, two fragment sets (16 registers) read one channel tile ahead;
//   * the weight image of a stage (16 KB) arrives by LDS-DMA into a four-slot ring THREE stages ahead (issued behind the stage's
//     barrier, when every wave is done with the slot it overwrites); one barrier per stage with a counted vmcnt;
//   * the 48 MFMAs of a stage and their fillers are placed by hand in 48 slots (sched_barrier(0) between slots); the 256
//     vector registers hold two stages of raw operand values (loads two stages ahead, as in gemm_x2d.hip), both plane sets,
//     the coefficient quads and the epilogue's staging -- nothing spills;
//   * epilogues: forward (bias / row bias / BatchNorm statistics): the accumulators go through a wave-private LDS transposition
//     and every buffer_store_dwordx4 writes 4 rows x 256 B; data gradients: straight from the registers, a store instruction
//     (dwordx2: the even and the odd position of a lane) writes 2 rows x 256 B.  The epilogue's LDS is its OWN area (one
//     workgroup per CU has 160 KB), so the next tile's first weight stages and operand loads are requested BEFORE the epilogue
//     and have landed when it ends (gemm_x2d.hip: 5-8 k cycles of set-up per tile exposed).
// Launch conditions (gemm_x2f_takes): M % 256 == 0, P % 256 == 0, K % 64 == 0, 16-B aligned whole-tile output, row bias per run of
// >= 8 positions, no REDK form; everything else stays with gemm_x2d.hip.  Knob x2_direct = 12 switches this kernel off (A/B);
// the product runs every launch that fits on it.
#include "mlp_common.h"
#include "split_common.h"
#include <type_traits>

using namespace usip_mlp;

namespace {

constexpr int FBM = 256, FBN = 256, FNT = 256, FSLOTS = 4;
constexpr int FPL = FBM * 32;                                  // bytes of one plane of one 16-k stage of the weights
constexpr int FSTAGE = 2 * FPL;                                // hi + lo: 16 KB
constexpr int FTRS = 68;                                       // floats per transposition row (64 + 4: conflict-free b128 writes)
constexpr int FSCR_FLOATS = 4 * 2 * 32 * FTRS + 2 * 4 * FBM;   // transposition areas (two per wave) + statistics exchange

// byte offset of (row, 16-B half) inside a [rows][16 fp16] plane (the image usip_mlp_split2h_f32 writes)
__device__ __forceinline__ int f_lds_off(int row, int half) { return row * 32 + ((half ^ (row >> 3)) & 1) * 16; }

// two fp32 -> packed fp16 high parts and packed fp16 low parts (x = hi + lo up to 2^-22 |x|), 3 VALU instructions
__device__ __forceinline__ void f_split_pair(float x, float y, unsigned& hi, unsigned& lo)
{
    const f32x2 v = {x, y};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));          // v_cvt_pk_f16_f32, RNE
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(lo) : "v"(x), "v"(hi));
    asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(y), "v"(hi));
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// One accumulator element, AGPR -> VGPR, where the epilogue wants it.  Written out because hipcc, left to itself, copies
// accumulator tuples to vector registers WHOLESALE at the top of the epilogue (130+ v_accvgpr_read in a row, then spills of the
// state that lives across the epilogue -- and a pending scratch reload at the K loop's header turns every trip's first
// counted wait into vmcnt(0), profiles/r06i_*).  The element is a sub-register of the tuple the MFMA statements pinned to AGPRs.
__device__ __forceinline__ float f_aread(const f32x16& t, int r)
{
    float x;
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(t[r]));
    return x;
}

// Forward epilogue.  After the swapped MFMAs lane (c, half) holds channel c of each 32-channel tile i and, per block, the
// positions 8 g + 4 half + e of the block (r = 4 g + e); block 0 / 1 = even / odd positions of the wave's 64, so the lane's
// values (block 0, r), (block 1, r) for e = 0..3 are EIGHT consecutive positions 16 g + 8 half + 0..7.  They go to the wave's
// transposition area (row = channel, 64 positions + 4 pad), and come back in the store role: lane = (rr = l >> 4, cc = l & 15)
// reads 16 B of row rr + 4 k -- a buffer_store_dwordx4 writes 4 rows x 256 B.
// (Measured and NOT kept, profiles/r06t_deferred_pieces_ab.txt: the last two channel tiles left in the wave's LDS buffers and
// stored from inside the NEXT tile's K loop, one 1-KB piece per stage -- the epilogue drops 17 -> 14.5 k cycles, the loop gains
// 1.5 k (40 cycles per stage that carries a store) and the workgroup's last tile has to flush 16 pieces behind its epilogue:
// 512 x 512 202 against 210 us, 256 x 256 72 against 67.)
// The tile's BatchNorm partials leave LDS one tile LATE: four scattered 4-B stores per thread at the end of an epilogue are
// the youngest memory operations when the next tile's first counted wait comes (1-2 k cycles of exposed store latency per
// tile, profiles/r06j_x2f_tile_trace_512.txt: first trip 3.4 k cycles after nothing, 5.1 k after an epilogue).  The sums of
// tile t stay in `red` through tile t + 1's K loop and are stored at the top of ITS epilogue (or behind the last tile).
__device__ __forceinline__ void stats_flush_x2f(const GemmArgs& a, const float* scratch, int m0, int tn128, int tpc128)
{
    const float* red = scratch + 4 * 2 * 32 * FTRS;
    const int tid = threadIdx.x;
    // the caller allocated one statistics slot per 128 positions (usip_mlp_gemm_tiles): waves 0-1 / 2-3 fill the two slots
    const long long ntn = (long long)a.nb * tpc128;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        const float s = red[(2 * hf) * FBM + tid] + red[(2 * hf + 1) * FBM + tid];
        const float q = red[4 * FBM + (2 * hf) * FBM + tid] + red[4 * FBM + (2 * hf + 1) * FBM + tid];
        a.stats[(long long)(m0 + tid) * ntn + tn128 + hf] = s;
        a.stats[ntn * a.M + (long long)(m0 + tid) * ntn + tn128 + hf] = q;
    }
}

template <int EPI, bool RB>
__device__ __forceinline__ void epilogue_x2f(const GemmArgs& a, f32x16 (&acc)[8][2], float out_scale, float* scratch,
                                             int b, int m0, int p0, int prev_m0, int prev_tn128, int tpc128)
{
    if (EPI != EPI_NONE && prev_tn128 >= 0) {
        // the previous tile's sums (every wave wrote its part a whole K loop ago); nobody may overwrite `red` before all have read
        stats_flush_x2f(a, scratch, prev_m0, prev_tn128, tpc128);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* tr = scratch + wave * (2 * 32 * FTRS);              // two transposition buffers per wave
    float* red = scratch + 4 * 2 * 32 * FTRS;                  // [2][4 waves][256 channels]
    const int pw = p0 + wave * 64;
    const int rr = lane >> 4, cc = lane & 15;
    const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.Y + (long long)b * a.y_rows * a.P), 0, (unsigned)a.y_rows * (unsigned)a.P * 4u, 0x00020000);
    const int st_voff = ((m0 + rr) * a.P + pw + 4 * cc) * 4;
    float* const trw = tr + c * FTRS + 8 * half;
    const float* const trr = tr + rr * FTRS + 4 * cc;
    const int ngrp = RB ? a.P / a.rb_group : 0;
    int grp[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) grp[g] = RB ? (pw + 16 * g + 8 * half) / a.rb_group : 0;   // rb_group % 8 == 0: one group per run
    const float* rbp = RB ? a.rowbias + ((long long)b * a.M + m0 + c) * ngrp : nullptr;
    const float* biasp = a.bias ? a.bias + m0 + c : nullptr;
    const f32x2 os2 = {out_scale, out_scale};
    // channel tile i: scale + bias (+ row bias), statistics, 8 x 16-B writes into transposition buffer i & 1.  Pairs (even
    // position, odd position) on the packed fp32 VALU -- each half rounded exactly like its scalar twin; the two halves are two
    // running sums, added at the end.
    auto produce = [&](int i) {
#if defined(USIP_X2F_EXP) && USIP_X2F_EXP == 2                 // measurement build: transposed reads + stores only
        return;
#endif
        const float bv = biasp ? biasp[i * 32] : 0.0f;
        float rb[4] = {0.f, 0.f, 0.f, 0.f};
        if (RB) {
#pragma unroll
            for (int g = 0; g < 4; ++g) rb[g] = rbp[(long long)i * 32 * ngrp + grp[g]];
        }
        // four independent running sums (one per run of positions): a lone wave has nobody to hide a dependent chain behind
        f32x2 s2g[4], q2g[4];
        float* w = trw + (i & 1) * (32 * FTRS);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x2 u[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#if defined(USIP_X2F_EXP) && USIP_X2F_EXP == 3                 // measurement build: scalar fp32 instead of the packed forms
#pragma unroll
                for (int bk = 0; bk < 2; ++bk) {
                    float t = __builtin_fmaf(f_aread(acc[i][bk], 4 * g + e), out_scale, bv);
                    if (RB) t += rb[g];
                    u[e][bk] = t;
                }
                if (EPI == EPI_STATS) {
                    s2g[g] = (e == 0) ? u[e] : s2g[g] + u[e];
                    q2g[g] = (e == 0) ? u[e] * u[e] : __builtin_elementwise_fma(u[e], u[e], q2g[g]);
                }
                continue;
#endif
                const f32x2 av = {f_aread(acc[i][0], 4 * g + e), f_aread(acc[i][1], 4 * g + e)};
                u[e] = __builtin_elementwise_fma(av, os2, (f32x2){bv, bv});               // out_scale = 2^n: exact
                if (RB) u[e] = u[e] + (f32x2){rb[g], rb[g]};
                if (EPI == EPI_STATS) {
                    s2g[g] = (e == 0) ? u[e] : s2g[g] + u[e];
                    q2g[g] = (e == 0) ? u[e] * u[e] : __builtin_elementwise_fma(u[e], u[e], q2g[g]);
                }
            }
            *reinterpret_cast<float4*>(w + 16 * g) = make_float4(u[0][0], u[0][1], u[1][0], u[1][1]);
            *reinterpret_cast<float4*>(w + 16 * g + 4) = make_float4(u[2][0], u[2][1], u[3][0], u[3][1]);
        }
        if (EPI != EPI_NONE) {
            const f32x2 s2 = (s2g[0] + s2g[1]) + (s2g[2] + s2g[3]), q2 = (q2g[0] + q2g[1]) + (q2g[2] + q2g[3]);
            float s = s2[0] + s2[1], q = q2[0] + q2[1];
            s += __shfl_xor(s, 32);
            q += __shfl_xor(q, 32);
            if (half == 0) { red[wave * FBM + i * 32 + c] = s; red[4 * FBM + wave * FBM + i * 32 + c] = q; }
        }
    };
    // Two-deep: while the 8 transposed reads of tile i are in flight the wave computes and writes tile i + 1 (other buffer),
    // then stores tile i -- the LDS round trip of the first version (write, wait, read, wait, store: ~2.1 k cycles per channel
    // tile, profiles/r06c_x2f_tile_trace_512.txt) is covered by the next tile's arithmetic.  A wave's DS operations execute in
    // order, so a buffer's reads (tile i) precede its next writes (tile i + 2) without a wait.
    produce(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        // the transposition reads what OTHER lanes of this wave wrote: wait for the writes (gemm_x2d.hip, r04p)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        float4 w[8];
        const float* r = trr + (i & 1) * (32 * FTRS);
#pragma unroll
        for (int k = 0; k < 8; ++k) w[k] = *reinterpret_cast<const float4*>(r + 4 * k * FTRS);
        __builtin_amdgcn_sched_barrier(0);
        if (i < 7) produce(i + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const u32x4 d = {__float_as_uint(w[k].x), __float_as_uint(w[k].y), __float_as_uint(w[k].z), __float_as_uint(w[k].w)};
#if defined(USIP_X2F_EXP) && USIP_X2F_EXP == 1                 // measurement build: the epilogue without its stores
            asm volatile("" ::"v"(d));
#else
            __builtin_amdgcn_raw_buffer_store_b128(d, rY, st_voff, (i * 32 + 4 * k) * a.P * 4, st_aux<ST_X2F_FWD>());
#endif
        }
        // buffer_store_dwordx4 with an SGPR soffset reads its data registers late (gemm_x2d.hip): eight wait states
        // behind the last store before anything may overwrite them
        asm volatile("s_nop 7" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Data-gradient epilogue (no statistics, no bias): operands NOT swapped, lane = position pair (2 c, 2 c + 1) of the wave's 64,
// accumulator register r of tile i = channel row 32 i + 8 g + 4 half + e: a dwordx2 store writes rows r and r + 4, 256 B each.
__device__ __forceinline__ void epilogue_x2f_direct(const GemmArgs& a, f32x16 (&acc)[8][2], float out_scale, int b, int m0, int p0)
{
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.Y + (long long)b * a.y_rows * a.P), 0, (unsigned)a.y_rows * (unsigned)a.P * 4u, 0x00020000);
    const int voff = ((m0 + 4 * half) * a.P + p0 + wave * 64 + 2 * c) * 4;
    const int rowb = a.P * 4;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            u32x2 v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                v[e] = u32x2{__float_as_uint(f_aread(acc[i][0], 4 * g + e) * out_scale),
                             __float_as_uint(f_aread(acc[i][1], 4 * g + e) * out_scale)};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                __builtin_amdgcn_raw_buffer_store_b64(v[e], rY, voff, (i * 32 + 8 * g + e) * rowb, st_aux<ST_X2F_DGRAD>());
            asm volatile("s_nop 7" ::: "memory");              // the stores read their data registers late (see above)
            __builtin_amdgcn_sched_barrier(0);                 // one group of rows at a time (the scheduler otherwise reads dozens of
        }                                                      // accumulators ahead and spills the state that lives across the epilogue)
    }
}

// slot sl of a stage -> index among the slots that carry no fragment read (sl % 6 >= 2), or -1
__host__ __device__ constexpr int f_ns(int sl) { return (sl % 6) >= 2 ? (sl / 6) * 4 + (sl % 6) - 2 : -1; }

// Measurement build (-DUSIP_X2F_TRACE, tools/x2f_trace.py; never the product): s_memtime stamps parked in the lanes of two
// registers with v_writelane (the loop keeps its schedule), written out by workgroups 0..7 when they end.
#ifdef USIP_X2F_TRACE
}
__device__ unsigned g_x2f_trace[8 * 4 * 128];
extern "C" int usip_x2f_trace_read(void* dst, void* stream)
{
    return (int)hipMemcpyFromSymbolAsync(dst, HIP_SYMBOL(g_x2f_trace), sizeof(unsigned) * 8 * 4 * 128, 0, hipMemcpyDeviceToDevice,
                                         (hipStream_t)stream);
}
namespace {
#define X2F_TP(REG_, IDX_)                                                                                      \
    {                                                                                                           \
        unsigned long long tt_;                                                                                 \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(tt_)::"memory");                            \
        unsigned keep_;                                                                                         \
        asm volatile("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tv_writelane_b32 %0, %2, m0\n\ts_mov_b32 m0, %1" \
                     : "+v"(REG_), "=&s"(keep_) : "s"((unsigned)tt_), "s"((int)(IDX_)));                        \
    }
#else
#define X2F_TP(REG_, IDX_)
#endif

template <int PRO, int EPI, bool DIRECT>
__global__ __launch_bounds__(FNT) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_x2f_kernel(
    const GemmArgs a, const uint4* __restrict__ planes)
{
    constexpr bool POOL = (PRO == PRO_BN_BWD_POOL);
    constexpr bool TWO = (PRO == PRO_BN_BWD) || POOL;
    constexpr int NC = TWO ? 4 : 2;                            // prologue coefficients per input channel
    constexpr int KMAX = TWO ? 512 : 640;
    constexpr int KPAD = KMAX + 16;
    constexpr int RING = FSLOTS * FSTAGE;
    // ONE array (a second __shared__ object makes hipcc drain vmcnt before LDS reads): ring | epilogue scratch | coefficients
    constexpr int SCR = DIRECT ? 64 : FSCR_FLOATS * 4;         // (the data-gradient epilogue stores straight from the registers)
    __shared__ __attribute__((aligned(16))) unsigned char smem[RING + SCR + NC * KPAD * 4];
    float* scr = reinterpret_cast<float*>(smem + RING);
    float* cf = reinterpret_cast<float*>(smem + RING + SCR);   // [k][NC]

    const int tid = threadIdx.x, lane = tid & 63, c = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int tpc = a.P / FBN, nmt = a.M / FBM;
    const int total = a.nb * tpc * nmt;
    const int nk = a.K / XBK;                                  // even (K % 32 == 0)

    // operand scales (see gemm_x2d_kernel / gemm_x3p_kernel): weights carry theirs behind the image, the streamed operand's
    // comes from a rigorous bound of what the prologue can produce
    float xs, out_scale;
    {
        float* redm = scr;
        float bnd = 0.f;
        if (PRO == PRO_AFFINE_RELU) {
            const float rn = sqrtf((float)a.nb * (float)a.P);
            for (int k = tid; k < a.K; k += FNT) {
                const float c0 = a.coef[k], c1 = a.coef[a.K + k], mu = a.coef[2 * a.K + k], is = a.coef[3 * a.K + k];
                bnd = fmaxf(bnd, fabsf(c0) / is * rn + fabsf(__builtin_fmaf(mu, c0, c1)));
            }
        } else {
            for (int i = tid; i < (a.K + 63) / 64; i += FNT) bnd = fmaxf(bnd, a.coef[4 * a.K + i]);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) bnd = fmaxf(bnd, __shfl_xor(bnd, off));
        if (lane == 0) redm[wave] = bnd;
        __syncthreads();
        bnd = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]));
        xs = pow2_scale(bnd, X2H_TOP);
        const float ws = __uint_as_float(planes[(long long)nmt * nk * (FSTAGE / 16)].x);
        // (wave-uniform: keep it in a scalar register -- the forward kernel sits at the 256-VGPR edge)
        out_scale = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(1.0f / (xs * ws))));
        for (int i = tid; i < nk * XBK * NC; i += FNT) {
            const int k = i / NC, j = i % NC;
            cf[i] = a.coef[j * a.K + k] * xs;
        }
        __syncthreads();
    }

    // Buffer descriptors are LAUNCH constants (the whole operand tensor, the whole plane image): what changes from tile to tile
    // is one scalar byte offset per operand, so the software pipeline below can run ACROSS tile boundaries -- the last three
    // stages of a tile request (and its last stage converts) the first stages of the workgroup's NEXT tile.
    typedef int v4i32 __attribute__((ext_vector_type(4)));
    auto make_rsrc = [](const void* base, unsigned bytes) {
        const unsigned long long p = (unsigned long long)reinterpret_cast<uintptr_t>(base);
        return v4i32{(int)__builtin_amdgcn_readfirstlane((unsigned)p),
                     (int)__builtin_amdgcn_readfirstlane((unsigned)(p >> 32) & 0xffffu),
                     (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000};
    };
    const v4i32 rAv = make_rsrc(planes, (unsigned)nmt * (unsigned)nk * FSTAGE);
    const int a_voff = wave * 4096 + lane * 16;                // wave w copies bytes [4096 w, 4096 w + 4096) of a stage's 16 KB
    const unsigned ring_lds = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)smem);
    const int pgrp = POOL ? a.P / a.pool_group : 0;
    const unsigned x_bytes = (unsigned)a.nb * (unsigned)a.K * (unsigned)a.P * 4u;
    const unsigned pool_bytes = (unsigned)a.nb * (unsigned)a.K * (unsigned)pgrp * 4u;
    const int rs = a.P * 4, rsg = pgrp * 4;                    // row strides in bytes
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)(POOL ? a.X2 : a.X), 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rX2 = __builtin_amdgcn_make_buffer_rsrc((void*)(TWO ? a.X2 : a.X), 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rPd = __builtin_amdgcn_make_buffer_rsrc((void*)(POOL ? a.pool_dp : a.X), 0,
                                                                         POOL ? pool_bytes : 4u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rPa = __builtin_amdgcn_make_buffer_rsrc((void*)(POOL ? (const float*)a.pool_arg : a.X), 0,
                                                                         POOL ? pool_bytes : 4u, 0x00020000);
    // lane parts of the operand addresses: positions (2 c, 2 c + 1) of the wave's 64, k-half h
    // (one address register per row of a stage: the row's offset in the VECTOR address leaves one scalar offset per stage --
    // the lone wave's loop is bound by its own instruction issue, ~5 cycles per instruction, profiles/r06g_*)
    const int pl = wave * 64 + 2 * c;
    int xv[8], gv[POOL ? 8 : 1];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        xv[i] = pl * 4 + (h * 8 + i) * rs;
        if (POOL) gv[POOL ? i : 0] = (pl / a.pool_group) * 4 + (h * 8 + i) * rsg;
    }
    const int xkin = POOL ? pl % a.pool_group : 0;             // of the even position (256 % group == 0: the same for every tile)

    // raw operand values of one stage: row i = k 16 kt + 8 h + i, .x / .y = the even / odd position of the lane
    struct XSet { f32x2 rx[POOL ? 1 : 8]; float rd[POOL ? 8 : 1]; int rarg[POOL ? 8 : 1]; f32x2 ry[TWO ? 8 : 1]; };
    struct Frag { f16x8 lo, hi; };
    constexpr int NX = POOL ? 24 : (TWO ? 16 : 8);             // register loads of one stage of the streamed operand
    // Vector-memory instructions issued AFTER this wave's DMA pieces of stage kt+1 (slots 44..47 of stage kt-2) when stage
    // kt's barrier (in front of slot 42) is reached: stage kt-1's NX loads and 4 pieces (stage kt+2), stage kt's NX loads
    // (and, across a tile boundary, the epilogue's stores: more in flight only makes the wait stricter).
    constexpr int BARRIER_VMCNT = 2 * NX + 4;
    static_assert(BARRIER_VMCNT <= 63, "vmcnt is six bits");

    const int fa0 = f_lds_off(c, h);                           // weight fragment of channel tile t: row t*32 + c, half h
    const float4* cfl = reinterpret_cast<const float4*>(cf) + h * (8 * NC / 4);

    struct Tile { int b, m0, p0, tn128; int xo, go, wo; };    // xo / go / wo: scalar byte offsets of the tile's operand / pool / planes
    auto tile_of = [&](int v) {
        int L = v;
        if ((total & 7) == 0) L = (v & 7) * (total >> 3) + (v >> 3);
        const int mt = L % nmt, tn = L / nmt;
        const int b = tn / tpc, pt = tn % tpc;
        const int p0 = pt * FBN;
        return Tile{b, mt * FBM, p0, (b * tpc + pt) * 2, (b * a.K * a.P + p0) * 4,
                    POOL ? (b * a.K * pgrp + p0 / a.pool_group) * 4 : 0, mt * nk * FSTAGE};
    };

    XSet xs0, xs1;
    // Two of this wave's four 1-KB pieces of stage kt (pair 0 / 1): ONE m0 set-up for both -- the second piece differs by the
    // instruction's immediate offset, which applies to the memory AND the LDS address.  m0 is saved and restored INSIDE the
    // statement.
    auto dma_pair = [&](int wo, int kt, int pair) {
        const unsigned dst = ring_lds + (unsigned)((kt & (FSLOTS - 1)) * FSTAGE) + (unsigned)(wave * 4096) + (unsigned)(pair * 2048);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                     "buffer_load_dwordx4 %2, %3, %4 offen lds\n\t"
                     "buffer_load_dwordx4 %2, %3, %4 offen offset:1024 lds\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(dst), "v"(a_voff), "s"(rAv), "s"(wo + kt * FSTAGE + pair * 2048) : "memory");
    };
    auto load_x1 = [&](int so, int sg, XSet& S, int i) {       // row i of the stage at scalar offsets (so, sg)
        if (POOL) {
            S.rd[POOL ? i : 0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rPd, gv[POOL ? i : 0], sg, 0));
            S.rarg[POOL ? i : 0] = (int)__builtin_amdgcn_raw_buffer_load_b32(rPa, gv[POOL ? i : 0], sg, 0);
        } else {
            S.rx[POOL ? 0 : i] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rX, xv[i], so, st_aux<(PRO >= PRO_BN_BWD) ? LD_X2F_DGRAD : LD_X2F_FWD>()));
        }
        if (TWO) S.ry[TWO ? i : 0] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rX2, xv[i], so, st_aux<LD_X2F_DGRAD>()));
    };

    // prologue of row i (both positions) of the stage in S; cq: coefficients of the stage for this half-wave, TWO: one float4
    // (c0..c3) per k, else one per PAIR of k
    float4 cq[TWO ? 8 : 4];
    float cv[2][8];
    auto cf_read = [&](int kt, int i) { cq[i] = cfl[kt * (XBK * NC / 4) + i]; };
    auto conv_blk = [&](const XSet& S, int i, int bk) {
        if (TWO) {
            const float4 c4 = cq[TWO ? i : 0];
            float x;
            if (POOL) x = (S.rarg[POOL ? i : 0] == xkin + bk) ? S.rd[POOL ? i : 0] : 0.f;
            else x = S.rx[POOL ? 0 : i][bk];
            cv[bk][i] = pro_apply<PRO_BN_BWD>(x, S.ry[TWO ? i : 0][bk], c4.x, c4.y, c4.z, c4.w);
        } else {
            const float4 c4 = cq[TWO ? 0 : i / 2];
            cv[bk][i] = (i & 1) ? pro_apply<PRO_AFFINE_RELU>(S.rx[POOL ? 0 : i][bk], 0.f, c4.z, c4.w, 0.f, 0.f)
                                : pro_apply<PRO_AFFINE_RELU>(S.rx[POOL ? 0 : i][bk], 0.f, c4.x, c4.y, 0.f, 0.f);
        }
    };
    auto convert_all = [&](int kt, const XSet& S, unsigned (&ph)[2][4], unsigned (&pl_)[2][4]) {
#pragma unroll
        for (int i = 0; i < (TWO ? 8 : 4); ++i) cf_read(kt, i);
#pragma unroll
        for (int i = 0; i < 8; ++i) { conv_blk(S, i, 0); conv_blk(S, i, 1); }
#pragma unroll
        for (int bk = 0; bk < 2; ++bk)
#pragma unroll
            for (int j = 0; j < 4; ++j) f_split_pair(cv[bk][2 * j], cv[bk][2 * j + 1], ph[bk][j], pl_[bk][j]);
    };

    f32x16 acc[8][2];
    Frag FA, FB;
    auto read_frag = [&](int kt, int t, int which, Frag& F) {
        const unsigned char* As = smem + (kt & (FSLOTS - 1)) * FSTAGE + fa0 + t * 1024;
        if (which == 0) F.lo = *reinterpret_cast<const f16x8*>(As + FPL);
        else F.hi = *reinterpret_cast<const f16x8*>(As);
    };

    // One 16-k stage = 48 MFMAs in program order, each followed by the few other instructions that issue in its shadow.
    // Channel tile t = slots 6t .. 6t+5: A_lo.X_hi (block 0, 1), A_hi.X_lo (0, 1), A_hi.X_hi (0, 1) -- smallest terms first,
    // dependent MFMAs two apart.  Fillers: slots 6t, 6t+1 read the fragments of tile t+1 (tile 0 of stage kt+1 behind the
    // barrier); the other 32 slots ("ns" 0..31, 28 in front of the barrier) carry the coefficients, prologue and split of stage
    // kt+1 into the other plane set, the reload of each operand row (stage kt+3) right behind its last use, and -- behind the
    // barrier -- this wave's 4 DMA pieces of stage kt+3 (ring slot of stage kt-1: every wave is done with it).  "Stage kt+1 /
    // kt+3" wrap into the workgroup's next tile (xo3 / go3 / wo3 = that stage's tile, k3 its index there; nk % 4 == 0, so ring
    // slots and register sets continue across the boundary).  FIRST (stage 0 of a tile): the first product of every accumulator
    // takes C = 0 -- no 256 v_accvgpr_write per tile.
    auto stage = [&](auto first_tag, int kt, int k1, int k3, int xo3, int go3, int wo3, XSet& S, const unsigned (&chh)[2][4],
                     const unsigned (&cll)[2][4], unsigned (&nh)[2][4], unsigned (&nl)[2][4]) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const f16x8 xh0 = __builtin_bit_cast(f16x8, make_uint4(chh[0][0], chh[0][1], chh[0][2], chh[0][3]));
        const f16x8 xh1 = __builtin_bit_cast(f16x8, make_uint4(chh[1][0], chh[1][1], chh[1][2], chh[1][3]));
        const f16x8 xl0 = __builtin_bit_cast(f16x8, make_uint4(cll[0][0], cll[0][1], cll[0][2], cll[0][3]));
        const f16x8 xl1 = __builtin_bit_cast(f16x8, make_uint4(cll[1][0], cll[1][1], cll[1][2], cll[1][3]));
        const int so3 = xo3 + k3 * (XBK * rs), sg3 = go3 + k3 * (XBK * rsg);
        auto slot = [&](auto sl_tag) {
            constexpr int sl = decltype(sl_tag)::value;
            constexpr int t = sl / 6, i6 = sl % 6, bk = i6 & 1, prod = i6 / 2;
            if (sl == 42) {
                // this wave's DMA of stage kt+1 has landed: what was issued after it may stay in flight across the barrier;
                // lgkmcnt(0): its reads of the ring are done
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(BARRIER_VMCNT) : "memory");
                __builtin_amdgcn_s_barrier();
            }
            {
                Frag& F = (t & 1) ? FB : FA;
                const f16x8& x = (prod == 1) ? (bk ? xl1 : xl0) : (bk ? xh1 : xh0);
                const f16x8& f = (prod == 0) ? F.lo : F.hi;
                if constexpr (FIRST && prod == 0) {
                    if constexpr (DIRECT) asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %1, 0" : "=a"(acc[t][bk]) : "v"(x), "v"(f));
                    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=a"(acc[t][bk]) : "v"(x), "v"(f));
                } else {
                    if constexpr (DIRECT) asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %1, %0" : "+a"(acc[t][bk]) : "v"(x), "v"(f));
                    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[t][bk]) : "v"(x), "v"(f));
                }
            }
            if (i6 < 2) {
                if (t < 7) read_frag(kt, t + 1, i6, (t & 1) ? FA : FB);
                else read_frag(kt + 1, 0, i6, FA);
            }
            constexpr int ns = f_ns(sl);
            if (ns == 28) dma_pair(wo3, k3, 0);
            if (ns == 30) dma_pair(wo3, k3, 1);
            if (!TWO) {
                // ns 0: the four coefficient quads; per pair j of rows five slots from ns 2 + 5j: prologue of row 2j, of row
                // 2j+1, split of block 0 + reload of row 2j, split of block 1 + reload of row 2j+1, (one free)
                if (ns == 0) { cf_read(k1, 0); cf_read(k1, 1); cf_read(k1, 2); cf_read(k1, 3); }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int base = 2 + 5 * j;
                    if (ns == base) { conv_blk(S, 2 * j, 0); conv_blk(S, 2 * j, 1); }
                    if (ns == base + 1) { conv_blk(S, 2 * j + 1, 0); conv_blk(S, 2 * j + 1, 1); }
                    if (ns == base + 2) {
                        f_split_pair(cv[0][2 * j], cv[0][2 * j + 1], nh[0][j], nl[0][j]);
                        load_x1(so3, sg3, S, 2 * j);
                    }
                    if (ns == base + 3) {
                        f_split_pair(cv[1][2 * j], cv[1][2 * j + 1], nh[1][j], nl[1][j]);
                        load_x1(so3, sg3, S, 2 * j + 1);
                    }
                }
            } else {
                // ns 0, 1: coefficient quads of rows 0-3; rows 4-7 two..six slots ahead of their use; per pair j of rows six
                // slots from ns 2 + 6j: prologue (row 2j, block 0), (2j, 1), (2j+1, 0), (2j+1, 1), split of block 0 + reload of
                // row 2j, split of block 1 + reload of row 2j+1
                if (ns == 0) { cf_read(k1, 0); cf_read(k1, 1); }
                if (ns == 1) { cf_read(k1, 2); cf_read(k1, 3); }
                if (ns == 8) cf_read(k1, 4);
                if (ns == 9) cf_read(k1, 5);
                if (ns == 14) cf_read(k1, 6);
                if (ns == 15) cf_read(k1, 7);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int base = 2 + 6 * j;
                    if (ns == base) conv_blk(S, 2 * j, 0);
                    if (ns == base + 1) conv_blk(S, 2 * j, 1);
                    if (ns == base + 2) conv_blk(S, 2 * j + 1, 0);
                    if (ns == base + 3) conv_blk(S, 2 * j + 1, 1);
                    if (ns == base + 4) {
                        f_split_pair(cv[0][2 * j], cv[0][2 * j + 1], nh[0][j], nl[0][j]);
                        load_x1(so3, sg3, S, 2 * j);
                    }
                    if (ns == base + 5) {
                        f_split_pair(cv[1][2 * j], cv[1][2 * j + 1], nh[1][j], nl[1][j]);
                        load_x1(so3, sg3, S, 2 * j + 1);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
#define USIP_X2F_SLOT(N_) slot(std::integral_constant<int, N_>{});
#define USIP_X2F_SLOT6(N_) USIP_X2F_SLOT(N_) USIP_X2F_SLOT(N_ + 1) USIP_X2F_SLOT(N_ + 2) USIP_X2F_SLOT(N_ + 3) USIP_X2F_SLOT(N_ + 4) USIP_X2F_SLOT(N_ + 5)
        USIP_X2F_SLOT6(0) USIP_X2F_SLOT6(6) USIP_X2F_SLOT6(12) USIP_X2F_SLOT6(18)
        USIP_X2F_SLOT6(24) USIP_X2F_SLOT6(30) USIP_X2F_SLOT6(36) USIP_X2F_SLOT6(42)
#undef USIP_X2F_SLOT6
#undef USIP_X2F_SLOT
    };

    // (Measured and NOT kept: half of the workgroups starting 8-32 k cycles late, so that one half's epilogue stores run under
    // the other half's K loop -- the delay is simply added, 65.6 -> 67 / 71 / 77 us for the 256 x 256 forward
    // (profiles/r06k_stagger_ab.txt): a tile's 256 KB of stores take ~17 k cycles because of the CU's own write path, ~16 B/clk,
    // not because the whole chip stores at once.)
    [[maybe_unused]] unsigned trp = 0, trs = 0;                // trace: tile phases (8 per tile), stage ends of the second tile
    [[maybe_unused]] int tix = 0;
    int v = blockIdx.x;
    if (v >= total) return;
    unsigned ah[2][4], al[2][4], bh[2][4], bl[2][4];
    Tile T = tile_of(v);
    {
        // the workgroup's first tile: stages 0..2 of the weights, stages 0 and 1 of the operand, stage 0 converted, stage 2
        // requested -- the state every later tile finds when its predecessor's last stage ends
#pragma unroll
        for (int st = 0; st < 3; ++st) { dma_pair(T.wo, st, 0); dma_pair(T.wo, st, 1); }
#pragma unroll
        for (int i = 0; i < 8; ++i) load_x1(T.xo, T.go, xs0, i);
#pragma unroll
        for (int i = 0; i < 8; ++i) load_x1(T.xo + XBK * rs, T.go + XBK * rsg, xs1, i);
        convert_all(0, xs0, ah, al);                           // (waits for the loads of stage 0)
#pragma unroll
        for (int i = 0; i < 8; ++i) load_x1(T.xo + 2 * XBK * rs, T.go + 2 * XBK * rsg, xs0, i);   // stage s lives in set s & 1
        // the 12 DMA pieces are older than every load still in flight (2 NX): stages 0..2 of the weights have landed
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NX) : "memory");
        __builtin_amdgcn_s_barrier();
        read_frag(0, 0, 0, FA); read_frag(0, 0, 1, FA);
    }
    int prev_m0 = 0, prev_tn128 = -1;                          // the tile whose BatchNorm partials are still in LDS
    for (;;) {
        const int vn = v + (int)gridDim.x;
        const bool more = vn < total;
        const Tile Tn = more ? tile_of(vn) : T;                // (the last tile requests its own first stages again: harmless)
        X2F_TP(trp, tix * 8 + 0)                               // 0: tile start
        // two stages per trip: register sets and plane sets swap roles; the first trip is peeled (C = 0 in stage 0)
        stage(std::true_type{}, 0, 1, 3, T.xo, T.go, T.wo, xs1, ah, al, bh, bl);
        stage(std::false_type{}, 1, 2, 4 < nk ? 4 : 0, 4 < nk ? T.xo : Tn.xo, 4 < nk ? T.go : Tn.go, 4 < nk ? T.wo : Tn.wo,
              xs0, bh, bl, ah, al);
        X2F_TP(trp, tix * 8 + 1)                               // 1: first trip (stages 0, 1) done
        X2F_TP(trp, tix * 8 + 2)
        for (int kt = 2; kt < nk; kt += 2) {
            // stage kt + 3 / kt + 4 of this tile, or stage (.. - nk) of the next one
            const bool w3 = kt + 3 >= nk, w4 = kt + 4 >= nk;
            stage(std::false_type{}, kt, kt + 1, w3 ? kt + 3 - nk : kt + 3, w3 ? Tn.xo : T.xo, w3 ? Tn.go : T.go, w3 ? Tn.wo : T.wo,
                  xs1, ah, al, bh, bl);
#ifdef USIP_X2F_TRACE
            if (tix == 1 && kt < 64) X2F_TP(trs, kt)
#endif
            stage(std::false_type{}, kt + 1, kt + 2 >= nk ? 0 : kt + 2, w4 ? kt + 4 - nk : kt + 4, w4 ? Tn.xo : T.xo, w4 ? Tn.go : T.go,
                  w4 ? Tn.wo : T.wo, xs0, bh, bl, ah, al);
#ifdef USIP_X2F_TRACE
            if (tix == 1 && kt < 63) X2F_TP(trs, kt + 1)
#endif
        }
        X2F_TP(trp, tix * 8 + 3)                               // 3: loop done
        X2F_TP(trp, tix * 8 + 4)
        // MFMA results need up to 18 wait states before a v_accvgpr_read and the compiler does not see asm MFMAs: 24 here, in
        // program order behind the last MFMA statement (volatile asm keeps its order) and in front of everything the compiler
        // schedules below the fence.  (NOT one statement per accumulator with a tied "+a" operand, as gemm_x2d.hip has it: with
        // all 256 AGPRs live the register allocator answered that with ~300 v_accvgpr_mov / read / write per tile -- the first
        // version's 5.4 k "request" cycles, profiles/r06c_x2f_tile_trace_512.txt.)
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        X2F_TP(trp, tix * 8 + 5)                               // 5: accumulators readable
        if constexpr (DIRECT) epilogue_x2f_direct(a, acc, out_scale, T.b, T.m0, T.p0);
        else if (a.rowbias) epilogue_x2f<EPI, true>(a, acc, out_scale, scr, T.b, T.m0, T.p0, prev_m0, prev_tn128, tpc * 2);
        else epilogue_x2f<EPI, false>(a, acc, out_scale, scr, T.b, T.m0, T.p0, prev_m0, prev_tn128, tpc * 2);
        prev_m0 = T.m0;
        prev_tn128 = T.tn128;
        X2F_TP(trp, tix * 8 + 6)                               // 6: epilogue issued
#ifdef USIP_X2F_TRACE
        ++tix;
#endif
        if (!more) break;
        v = vn;
        T = Tn;
    }
    if constexpr (!DIRECT && EPI != EPI_NONE) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the last tile's sums: every wave's part is in LDS
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        stats_flush_x2f(a, scr, prev_m0, prev_tn128, tpc * 2);
    }
    // the last tile's repeats of its first stages: no LDS-DMA may be in flight when the workgroup's LDS is handed on
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef USIP_X2F_TRACE
    if (blockIdx.x < 8) {
        g_x2f_trace[(blockIdx.x * 4 + wave) * 128 + lane] = trp;
        g_x2f_trace[(blockIdx.x * 4 + wave) * 128 + 64 + lane] = trs;
    }
#endif
}

}  // namespace

namespace usip_mlp {

// Does gemm_x2f_kernel take this launch?  (The default for every shape it fits; knob x2_direct = 12: never.)
bool gemm_x2f_takes(const GemmArgs& a, int pro)
{
    // Knob x2_direct: 0 (the product) = every launch that fits; 12 and every measurement knob of the family: gemm_x2d.hip only;
    // 13 = the forward launches only, 14 = all but pro 2, 15 = all but pro 3 (same-box A/B of the data-gradient forms).
    const int knob = usip_tuning_value(USIP_TUNE_X2_DIRECT);
    if (knob != 0 && (knob < 13 || knob > 15)) return false;
    if ((knob == 13 && pro != PRO_AFFINE_RELU) || (knob == 14 && pro == PRO_BN_BWD) || (knob == 15 && pro == PRO_BN_BWD_POOL)) return false;
    if (pro != PRO_AFFINE_RELU && pro != PRO_BN_BWD && pro != PRO_BN_BWD_POOL) return false;
    if (a.red_out) return false;
    if (a.M % FBM != 0 || a.P % FBN != 0 || a.K % (4 * XBK) != 0) return false;   // (four ring slots: stages continue across tiles)
    if (a.K > (pro == PRO_AFFINE_RELU ? 640 : 512)) return false;
    if (!a.y_vec || (a.rowbias && a.rb_group % 8 != 0)) return false;
    if ((long long)a.y_rows * a.P * 4 >= (1LL << 31) || (long long)a.nb * a.K * a.P * 4 >= (1LL << 31)) return false;   // 32-bit offsets
    if ((reinterpret_cast<uintptr_t>(a.X) & 7u) != 0 || (reinterpret_cast<uintptr_t>(a.X2) & 7u) != 0) return false;
    if (pro == PRO_BN_BWD_POOL && (a.pool_group % 2 != 0 || FBN % a.pool_group != 0)) return false;
    if (pro != PRO_AFFINE_RELU && (a.stats || a.bias || a.rowbias)) return false;   // data gradients: the direct epilogue only
    return true;
}

int launch_gemm_x2f(const GemmArgs& a_in, const uint4* pl, int pro, hipStream_t st)
{
    const GemmArgs& a = a_in;
    const int tpc = a.P / FBN, nmt = a.M / FBM;
    const long long total = (long long)a.nb * tpc * nmt;
    if (total > 0x7fffffffLL) return USIP_EINVAL;
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8)
            n = 256;
        return n;
    }();
    const long long slots = (long long)cus / 8 * 8;            // one workgroup per CU (LDS: ~120 KB, 512 registers per wave)
    dim3 grid((unsigned)((total <= slots || (total & 7)) ? total : slots)), block(FNT);
    if (pro == PRO_AFFINE_RELU) {
        if (a.stats) USIP_LAUNCH((gemm_x2f_kernel<PRO_AFFINE_RELU, EPI_STATS, false>), grid, block, 0, st, a, pl);
        else USIP_LAUNCH((gemm_x2f_kernel<PRO_AFFINE_RELU, EPI_NONE, false>), grid, block, 0, st, a, pl);
    } else if (pro == PRO_BN_BWD) {
        USIP_LAUNCH((gemm_x2f_kernel<PRO_BN_BWD, EPI_NONE, true>), grid, block, 0, st, a, pl);
    } else {
        USIP_LAUNCH((gemm_x2f_kernel<PRO_BN_BWD_POOL, EPI_NONE, true>), grid, block, 0, st, a, pl);
    }
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}

}  // namespace usip_mlp
