// usip_amd/csrc/gemm_x2e.hip -- the f32x2 FORWARD GEMM of the 256..512-wide shared-MLP layers (models/layers.py:208-216,
// :293-303, :401-440; BN + ReLU prologue, optional statistics / row bias) with BOTH operands arriving by LDS-DMA.
// Round 4, the design the cycle-level ablation of gemm_x2d.hip pointed to (profiles/r04_mfma_sustained_clock.txt, part 7:
// a stage of that kernel is 1320 cycles of the wave's own instruction issue + 240 of weight-DMA issue + 650-900 of
// operand-load latency; its 8 buffer_load_dword per wave and stage can run only two stages ahead -- 128 vector registers).
//
// One 8-wave workgroup per CU, tile 256 channels x 256 positions; wave w owns positions [32 w, 32 w + 32) of ALL 256
// channels (8 accumulator tiles in AGPRs), as in gemm_x2d.hip.  What changed:
//   * the streamed operand's RAW fp32 tile of a stage (16 k-rows x 256 positions = 16 KB) is copied memory -> LDS by
//     LDS-DMA, one 1-KB piece = one full k-row of the tile (8 whole cache lines), two pieces per wave and stage, FOUR
//     stages ahead into a four-slot ring: no registers in the prefetch, nothing for the compiler to wait on;
//   * a lane reads its 8 k-values of the next stage from that ring (8 ds_read_b32, lanes = consecutive positions:
//     conflict-free) right behind the stage's barrier, applies the prologue and the two-plane split in registers (the
//     result IS the MFMA operand, as before);
//   * the weight image of a stage (16 KB) is shared by 8 waves instead of 4: two DMA pieces per wave and stage
//     instead of four.  Per CU and stage the address unit sees 32 LDS-DMA pieces instead of 32 + 64 loads.
//   * no vector memory instruction in the loop returns data to a register, so the only counter is vmcnt over the DMA
//     pieces: at a stage's barrier at most the 8 pieces of this and the previous stage may be in flight.
// Arithmetic (prologue, scales, plane products, their order, the epilogue's bias / row bias / statistics order within a
// 256-position tile) is gemm_x2d.hip's; the statistics of a 256-position tile go to the FIRST of the two 128-position
// slots the caller allocated for it, zero to the second (usip_mlp_gemm_tiles is unchanged).
// NOT the default (see gemm_x2e_takes below for the numbers).  Launch conditions: knob x2_direct = 10, pro = BN + ReLU, M % 256 == 0, K % 32 == 0, P % 256 == 0, 16-B
// aligned whole-tile output, row bias per run of >= 4 positions.
#include "mlp_common.h"
#include "split_common.h"
#include <type_traits>

using namespace usip_mlp;

namespace {

constexpr int EBM = 256, EBN = 256, ENT = 512, ENW = 8, ESLOTS = 4, EBK = 16;
constexpr int EPL = EBM * 32;                                  // bytes of one plane of one 16-k stage of the weights
constexpr int EWSTAGE = 2 * EPL;                               // hi + lo: 16 KB
constexpr int EXSTAGE = EBK * EBN * 4;                         // raw fp32 tile of the streamed operand: 16 KB
constexpr int EKPAD = 640;

// byte offset of (row, 16-B half) inside a [rows][16 fp16] plane (the image usip_mlp_split2h_f32 writes)
__device__ __forceinline__ int e_lds_off(int row, int half) { return row * 32 + ((half ^ (row >> 3)) & 1) * 16; }

// two fp32 -> packed fp16 high parts and packed fp16 low parts (x = hi + lo up to 2^-22 |x|), 3 VALU instructions
__device__ __forceinline__ void e_split_pair(float x, float y, unsigned& hi, unsigned& lo)
{
    const f32x2 v = {x, y};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));          // v_cvt_pk_f16_f32, RNE
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(lo) : "v"(x), "v"(hi));
    asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(y), "v"(hi));
}

// Epilogue: gemm_x2d.hip's lean form for 8 waves (every channel tile through a wave-private 32 x 32 transposition in LDS,
// then stores of 8 rows x 128 B; bias, row bias, statistics in the same order).
constexpr int ETRS = 36;                                       // floats per transposition row (32 + 4: conflict-free b128 writes)
typedef unsigned e_u32x4 __attribute__((ext_vector_type(4)));

template <int EPI, bool RB>
__device__ __forceinline__ void epilogue_x2e(const GemmArgs& a, f32x16 (&acc)[8][1], float out_scale, float* scratch,
                                             int b, int m0, int p0, int tn, int tpc)
{
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* tr = scratch + wave * (32 * ETRS);
    float* red = scratch + ENW * 32 * ETRS;                    // [2][8 waves][256 channels]
    const int pw = p0 + wave * 32;
    const int rr = lane >> 3, cc = lane & 7;
    const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.Y + (long long)b * a.y_rows * a.P), 0, (unsigned)a.y_rows * (unsigned)a.P * 4u, 0x00020000);
    const int st_voff = ((m0 + rr) * a.P + pw + 4 * cc) * 4;
    float* const trw = tr + c * ETRS + 4 * half;
    const float* const trr = tr + rr * ETRS + 4 * cc;
    float bv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) bv[i] = a.bias ? a.bias[m0 + i * 32 + c] : 0.0f;
    const int ngrp = RB ? a.P / a.rb_group : 0;
    int grp[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) grp[g] = RB ? (pw + 8 * g + 4 * half) / a.rb_group : 0;
    const float* rbp = RB ? a.rowbias + ((long long)b * a.M + m0 + c) * ngrp : nullptr;
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
            if (half == 0) { red[wave * EBM + i * 32 + c] = s; red[ENW * EBM + wave * EBM + i * 32 + c] = q; }
        }
        // the transposition reads what OTHER lanes of this wave just wrote: wait for the writes (gemm_x2d.hip, r04p)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float4 w = *reinterpret_cast<const float4*>(trr + 8 * k * ETRS);
            const e_u32x4 d = {__float_as_uint(w.x), __float_as_uint(w.y), __float_as_uint(w.z), __float_as_uint(w.w)};
            __builtin_amdgcn_raw_buffer_store_b128(d, rY, st_voff, (i * 32 + 8 * k) * a.P * 4, st_aux<ST_X2D>());
        }
        // gfx950 / ROCm 7.2: the stores read their data registers late (gemm_x2d.hip): eight wait states behind the last one
        asm volatile("s_nop 7" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
    if (EPI != EPI_NONE) {
        __syncthreads();
        // the caller's statistics array has one slot per 128 positions: this tile's sums go to the first of its two
        const long long ntn = (long long)a.nb * tpc * 2;
        if (tid < EBM) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int w = 0; w < ENW; ++w) { s += red[w * EBM + tid]; q += red[ENW * EBM + w * EBM + tid]; }
            a.stats[(long long)(m0 + tid) * ntn + 2 * tn] = s;
            a.stats[ntn * a.M + (long long)(m0 + tid) * ntn + 2 * tn] = q;
        } else {
            a.stats[(long long)(m0 + tid - EBM) * ntn + 2 * tn + 1] = 0.f;
            a.stats[ntn * a.M + (long long)(m0 + tid - EBM) * ntn + 2 * tn + 1] = 0.f;
        }
    }
}

template <int EPI>
__global__ __launch_bounds__(ENT) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_x2e_kernel(
    const GemmArgs a, const uint4* __restrict__ planes)
{
    constexpr int NC = 2;
    constexpr int WRING = ESLOTS * EWSTAGE, XRING = ESLOTS * EXSTAGE;
    __shared__ __attribute__((aligned(16))) unsigned char smem[WRING + XRING + NC * EKPAD * 4];
    float* cf = reinterpret_cast<float*>(smem + WRING + XRING);       // [k][c0, c1], zero beyond K

    const int tid = threadIdx.x, lane = tid & 63, c = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tpc = a.P / EBN, nmt = a.M / EBM;
    const int total = a.nb * tpc * nmt;
    const int nk = a.K / EBK;

    // operand scales (gemm_x2d.hip): weights carry theirs behind the image, the streamed operand's comes from the bound
    // |gamma| sqrt(n) + |beta| of its BatchNorm
    float xs, out_scale;
    {
        float* redm = reinterpret_cast<float*>(smem);
        float bnd = 0.f;
        const float rn = sqrtf((float)a.nb * (float)a.P);
        for (int k = tid; k < a.K; k += ENT) {
            const float c0 = a.coef[k], c1 = a.coef[a.K + k], mu = a.coef[2 * a.K + k], is = a.coef[3 * a.K + k];
            bnd = fmaxf(bnd, fabsf(c0) / is * rn + fabsf(__builtin_fmaf(mu, c0, c1)));
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) bnd = fmaxf(bnd, __shfl_xor(bnd, off));
        if (lane == 0) redm[wave] = bnd;
        __syncthreads();
        bnd = redm[0];
#pragma unroll
        for (int w = 1; w < ENW; ++w) bnd = fmaxf(bnd, redm[w]);
        xs = pow2_scale(bnd, X2H_TOP);
        const float ws = __uint_as_float(planes[(long long)nmt * nk * (EWSTAGE / 16)].x);
        out_scale = 1.0f / (xs * ws);
        for (int i = tid; i < nk * EBK * NC; i += ENT) {
            const int k = i / NC, j = i % NC;
            cf[i] = a.coef[j * a.K + k] * xs;
        }
        __syncthreads();                                       // redm is read; the rings may be written from here on
    }

    const unsigned wring_lds = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)smem);
    const unsigned xring_lds = wring_lds + WRING;
    typedef int v4i32 __attribute__((ext_vector_type(4)));
    auto make_rsrc = [](const void* base, unsigned bytes) {
        const unsigned long long p = (unsigned long long)reinterpret_cast<uintptr_t>(base);
        return v4i32{(int)__builtin_amdgcn_readfirstlane((unsigned)p),
                     (int)__builtin_amdgcn_readfirstlane((unsigned)(p >> 32) & 0xffffu),
                     (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000};
    };

    // persistent: one workgroup per CU walks its tiles (scale and coefficient table above are per launch)
    for (int v = blockIdx.x; v < total; v += gridDim.x) {
    int L = v;
    if ((total & 7) == 0) L = (v & 7) * (total >> 3) + (v >> 3);
    const int mt = L % nmt, tn = L / nmt;
    const int b = tn / tpc, pt = tn % tpc;
    const int m0 = mt * EBM, p0 = pt * EBN;

    // weight image of this row tile: a stage is 16 KiB = 2 x (512 lanes x 16 B); wave w copies bytes [1024 w, +1024) of each half
    const v4i32 rAv = make_rsrc(planes + (long long)mt * nk * (EWSTAGE / 16), (unsigned)nk * EWSTAGE);
    const int a_voff = tid * 16;
    // streamed operand: piece j of wave w = k-row 2 w + j of the stage, 256 positions = 1 KiB, lane = 4 positions
    const v4i32 rXv = make_rsrc(a.X + (long long)b * a.K * a.P, (unsigned)a.K * (unsigned)a.P * 4u);
    const int x_voff = p0 * 4 + lane * 16;
    const int rs = a.P * 4;
    auto dma_w = [&](int kt, int j) {
        const unsigned dst = wring_lds + (unsigned)((kt & (ESLOTS - 1)) * EWSTAGE) + (unsigned)(wave * 1024 + j * 8192);
        unsigned keep;                                         // m0 saved and restored inside the statement (see gemm_x2d.hip)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(dst), "v"(a_voff), "s"(rAv), "s"(kt * EWSTAGE + j * 8192) : "memory");
    };
    auto dma_x = [&](int kt, int j) {
        const int r = wave * 2 + j;
        const unsigned dst = xring_lds + (unsigned)((kt & (ESLOTS - 1)) * EXSTAGE) + (unsigned)(r * 1024);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(dst), "v"(x_voff), "s"(rXv), "s"((kt * EBK + r) * rs) : "memory");
    };

    // raw values of the NEXT stage (read from the ring behind the barrier that guarantees they landed), the lane's
    // 8 k: rows 8 h + i, position 32 wave + c
    float rx[8];
    const unsigned char* const xrd = smem + WRING + (8 * h) * 1024 + (wave * 32 + c) * 4;
    auto read_x1 = [&](int kt, int i) {
        rx[i] = *reinterpret_cast<const float*>(xrd + (kt & (ESLOTS - 1)) * EXSTAGE + i * 1024);
    };
    const float4* cfl = reinterpret_cast<const float4*>(cf) + h * (8 * NC / 4);
    float4 cq[4];
    float cv[8];
    auto cf_read = [&](int kt, int i) { cq[i] = cfl[kt * (EBK * NC / 4) + i]; };
    auto conv_elem = [&](int i) {
        const float4 c4 = cq[i / 2];
        cv[i] = (i & 1) ? pro_apply<PRO_AFFINE_RELU>(rx[i], 0.f, c4.z, c4.w, 0.f, 0.f)
                        : pro_apply<PRO_AFFINE_RELU>(rx[i], 0.f, c4.x, c4.y, 0.f, 0.f);
    };

    f32x16 acc[8][1];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][0][r] = 0.0f;

    const int fa0 = e_lds_off(c, h);
    struct Frag { f16x8 lo[2], hi[2]; };
    Frag FA, FB;
    auto read_pair = [&](int kt, int u, int half, Frag& F) {
        const unsigned char* As = smem + (kt & (ESLOTS - 1)) * EWSTAGE + fa0 + u * 2048;
        if (half == 0) {
            F.lo[0] = *reinterpret_cast<const f16x8*>(As + EPL);
            F.lo[1] = *reinterpret_cast<const f16x8*>(As + EPL + 1024);
        } else {
            F.hi[0] = *reinterpret_cast<const f16x8*>(As);
            F.hi[1] = *reinterpret_cast<const f16x8*>(As + 1024);
        }
    };

    // One 16-k stage = 24 MFMAs in program order with the other instructions placed behind them (gemm_x2d.hip's slots):
    //   slots  0..17  pairs 0-2; fillers: fragment reads of the next pair, coefficients of stage kt+1 (slot 2), prologue +
    //                 split of stage kt+1 (its raw values are in rx since the end of the previous stage), the stage's four
    //                 DMA pieces: W(kt+3) at slots 4, 5, X(kt+4) at slots 6, 7
    //   barrier       at most 8 pieces (this stage's and the previous one's) in flight: W(kt+1) and X(kt+2) have landed
    //   slots 18..23  pair 3; fillers: fragment reads of pair 0 of stage kt+1, raw values of stage kt+2 into rx
    auto stage = [&](int kt, const unsigned (&ch)[4], const unsigned (&cl)[4], unsigned (&nh)[4], unsigned (&nl)[4]) {
        const f16x8 xh = __builtin_bit_cast(f16x8, make_uint4(ch[0], ch[1], ch[2], ch[3]));
        const f16x8 xl = __builtin_bit_cast(f16x8, make_uint4(cl[0], cl[1], cl[2], cl[3]));
        const int k1 = min(kt + 1, nk - 1), k2 = min(kt + 2, nk - 1), k3 = min(kt + 3, nk - 1), k4 = min(kt + 4, nk - 1);
        auto mfma = [&](int sl) {
            const int u = sl / 6, i = sl % 6, t = 2 * u + (i & 1);
            Frag& F = (u & 1) ? FB : FA;
            const f16x8& x = (i / 2 == 1) ? xl : xh;
            const f16x8& f = (i / 2 == 0) ? F.lo[i & 1] : F.hi[i & 1];
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[t][0]) : "v"(x), "v"(f));
        };
        auto filler = [&](int sl) {
            const int u = sl / 6, i = sl % 6;
            if (i < 2) {
                if (u < 3) read_pair(kt, u + 1, i, (u & 1) ? FA : FB);
                else read_pair(kt + 1, 0, i, FA);
            }
            if (sl == 4 || sl == 5) dma_w(k3, sl - 4);
            if (sl == 6 || sl == 7) dma_x(k4, sl - 6);
            if (sl == 2) { cf_read(k1, 0); cf_read(k1, 1); cf_read(k1, 2); cf_read(k1, 3); }
            constexpr int CA[4] = {3, 5, 9, 11}, CB[4] = {4, 8, 10, 14};          // slots of pair j: prologue / split
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (sl == CA[j]) { conv_elem(2 * j); conv_elem(2 * j + 1); }
                if (sl == CB[j]) e_split_pair(cv[2 * j], cv[2 * j + 1], nh[j], nl[j]);
            }
            if (sl >= 18 && sl < 22) { read_x1(k2, 2 * (sl - 18)); read_x1(k2, 2 * (sl - 18) + 1); }
        };
#define USIP_X2E_SLOT(N_)                                              \
        mfma(N_);                                                      \
        filler(N_);                                                    \
        __builtin_amdgcn_sched_barrier(0);
        USIP_X2E_SLOT(0) USIP_X2E_SLOT(1) USIP_X2E_SLOT(2) USIP_X2E_SLOT(3) USIP_X2E_SLOT(4) USIP_X2E_SLOT(5)
        USIP_X2E_SLOT(6) USIP_X2E_SLOT(7) USIP_X2E_SLOT(8) USIP_X2E_SLOT(9) USIP_X2E_SLOT(10) USIP_X2E_SLOT(11)
        USIP_X2E_SLOT(12) USIP_X2E_SLOT(13) USIP_X2E_SLOT(14) USIP_X2E_SLOT(15) USIP_X2E_SLOT(16) USIP_X2E_SLOT(17)
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        USIP_X2E_SLOT(18) USIP_X2E_SLOT(19) USIP_X2E_SLOT(20) USIP_X2E_SLOT(21) USIP_X2E_SLOT(22) USIP_X2E_SLOT(23)
#undef USIP_X2E_SLOT
    };

    // prologue of the tile: X(0), W(0), X(1), W(1), X(2), W(2), X(3) in that order (14 pieces per wave; clamped beyond nk)
    unsigned ah[4], al[4], bh[4], bl[4];
    dma_x(0, 0); dma_x(0, 1);
    dma_w(0, 0); dma_w(0, 1);
    dma_x(min(1, nk - 1), 0); dma_x(min(1, nk - 1), 1);
    dma_w(min(1, nk - 1), 0); dma_w(min(1, nk - 1), 1);
    dma_x(min(2, nk - 1), 0); dma_x(min(2, nk - 1), 1);
    dma_w(min(2, nk - 1), 0); dma_w(min(2, nk - 1), 1);
    dma_x(min(3, nk - 1), 0); dma_x(min(3, nk - 1), 1);
    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");          // X(0), W(0) of this wave have landed ...
    __builtin_amdgcn_s_barrier();                              // ... and everybody's
#pragma unroll
    for (int i = 0; i < 8; ++i) read_x1(0, i);
#pragma unroll
    for (int i = 0; i < 4; ++i) cf_read(0, i);
#pragma unroll
    for (int i = 0; i < 8; ++i) conv_elem(i);
#pragma unroll
    for (int j = 0; j < 4; ++j) e_split_pair(cv[2 * j], cv[2 * j + 1], ah[j], al[j]);
    asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory"); // X(1), W(1)
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 8; ++i) read_x1(min(1, nk - 1), i);
    read_pair(0, 0, 0, FA); read_pair(0, 0, 1, FA);
    for (int kt = 0; kt < nk; kt += 2) {                       // nk is even (the launcher sends odd nk to gemm_x2d.hip)
        stage(kt, ah, al, bh, bl);
        stage(kt + 1, bh, bl, ah, al);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the clamped repeats of the last stages
    __syncthreads();                                           // the rings are scratch from here on
#pragma unroll
    for (int t = 0; t < 8; ++t) asm volatile("s_nop 7\n\ts_nop 7" : "+a"(acc[t][0]));
    float* scr = reinterpret_cast<float*>(smem);               // transposition areas + statistics exchange: 53 KB of the weight ring
    if (a.rowbias) epilogue_x2e<EPI, true>(a, acc, out_scale, scr, b, m0, p0, tn, tpc);
    else epilogue_x2e<EPI, false>(a, acc, out_scale, scr, b, m0, p0, tn, tpc);
    __syncthreads();                                           // the scratch becomes ring again
    }                                                          // tiles
}

}  // namespace

namespace usip_mlp {

// Does gemm_x2e_kernel take this forward launch?  OPT-IN (knob x2_direct = 10): measured at the end of round 4 it is
// 2-7 % faster than gemm_x2d.hip stand-alone (512 x 512: 216 vs 220 us, 512 x 256: 143-148 vs 149, 256 x 256: 74-75 vs 80;
// results bit-identical) and 0.6 % SLOWER inside the step (4.877 / 4.923 / 4.885 vs 4.854 / 4.882 / 4.852 ms, same box,
// alternating) -- as round 3's 256 x 256 tiles were.  Its stage takes 1930 cycles: the four older waves reach the barrier
// after ~1200 and wait ~640 for the four younger ones that share their SIMDs (tools/x2d_trace.py on a probe build).
bool gemm_x2e_takes(const GemmArgs& a, int pro)
{
    if (pro != PRO_AFFINE_RELU || (usip_tuning_value(USIP_TUNE_X2_DIRECT) & 15) != 10) return false;
    if (a.M % EBM != 0 || a.P % EBN != 0 || a.K % (2 * EBK) != 0 || a.K > EKPAD || a.K < 4 * EBK) return false;
    if (!a.y_vec || (a.rowbias && a.rb_group % 4 != 0)) return false;
    if ((long long)a.y_rows * a.P * 4 >= (1LL << 31) || (long long)a.K * a.P * 4 >= (1LL << 31)) return false;
    if ((reinterpret_cast<uintptr_t>(a.X) & 15u) != 0) return false;
    return true;
}

int launch_gemm_x2e(const GemmArgs& a, const uint4* pl, hipStream_t st)
{
    const int tpc = a.P / EBN, nmt = a.M / EBM;
    const long long total = (long long)a.nb * tpc * nmt;
    if (total > 0x7fffffffLL) return USIP_EINVAL;
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8)
            n = 256;
        return n;
    }();
    const long long slots = (long long)cus / 8 * 8;            // one workgroup per CU
    dim3 grid((unsigned)((total <= slots || (total & 7)) ? total : slots)), block(ENT);
    if (a.stats) USIP_LAUNCH((gemm_x2e_kernel<EPI_STATS>), grid, block, 0, st, a, pl);
    else USIP_LAUNCH((gemm_x2e_kernel<EPI_NONE>), grid, block, 0, st, a, pl);
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}

}  // namespace usip_mlp
