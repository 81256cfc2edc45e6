// usip_amd/csrc/shared_mlp_bf16.hip -- the shared-MLP GEMM and weight-gradient kernels with bf16 MULTIPLY:
// tensors stay fp32 in HBM, the prologue (BatchNorm+ReLU / BatchNorm-backward) runs in fp32, operands are
// rounded to bf16 (RNE, v_cvt_pk_bf16_f32) when they are written to LDS, v_mfma_f32_32x32x16_bf16 accumulates
// in fp32, and bias / row bias / BatchNorm statistics in the epilogue are fp32 as in shared_mlp.hip.
//
// NS = 1 is the perf mode of BASELINE.json configs[1] ("bf16"); it is NOT the parity mode: products carry
// 2^-9 relative rounding per operand.
//
// NS = 3 ("f32x3") is an fp32-ACCURATE product on the bf16 matrix cores: every fp32 operand is split exactly into
// three bf16 planes x = x0 + x1 + x2 (x0 = rne(x), x1 = rne(x - x0), x2 = rne(x - x0 - x1); the residual is below
// 2^-26 |x|) and the product is formed from the six plane pairs whose weight is >= 2^-18:
//     x.y ~= x0y0 + (x0y1 + x1y0) + (x1y1 + x0y2 + x2y0),     dropped: x1y2, x2y1, x2y2 <= 3 * 2^-27 |x||y|
// Each plane pair is an exact bf16 x bf16 product accumulated in fp32 by v_mfma_f32_32x32x16_bf16: six matrix
// instructions of 32 cycles each do the work of eight v_mfma_f32_32x32x2_f32 of 64 cycles (2.7x the fp32-MFMA
// rate at the same K), with a relative error at or below the fp32 FMA chain's own rounding (tests/ compare both
// with fp64).  gfx950 has no xf32/TF32 path; this is how the wide (compute-bound) layers leave the 157 TFLOP/s
// fp32 ceiling without giving up fp32 results.  Reference: models/layers.py:208-216, :293-303 under
// torch.autocast-style bf16, which the reference itself never ran -- tolerance is documented in DESIGN.md.
//
// LDS layout: both operands [row][k] with k contiguous (the MFMA lane (row = l & 31, h = l >> 5) reads the
// 8 k-values 8h..8h+7 of its row as one ds_read_b128).  Row pitch BK + 8 bf16 (48 B at BK = 16): eight
// consecutive rows cover all 32 banks, so b128 reads and writes are conflict free.
// Global -> register mapping: lane <-> consecutive row (m for the matrix operand, position p for the
// streamed one), thread owns K consecutive k of its row, so every load instruction of a wave is one
// contiguous 256-B segment and the LDS write of a thread is one or two 16-B vectors.
#include "mlp_common.h"

using namespace usip_mlp;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

namespace {

// n consecutive fp32 -> bf16, stored as 8- or 16-byte vectors
template <int N>
__device__ __forceinline__ void store_bf16_run(__bf16* dst, const float (&v)[N])
{
    if constexpr (N % 8 == 0) {
#pragma unroll
        for (int g = 0; g < N / 8; ++g) {
            bf16x8 pk;
#pragma unroll
            for (int i = 0; i < 8; ++i) pk[i] = (__bf16)v[g * 8 + i];
            *reinterpret_cast<bf16x8*>(dst + g * 8) = pk;
        }
    } else {
        static_assert(N % 4 == 0, "run length must be a multiple of 4");
#pragma unroll
        for (int g = 0; g < N / 4; ++g) {
            bf16x4 pk;
#pragma unroll
            for (int i = 0; i < 4; ++i) pk[i] = (__bf16)v[g * 4 + i];
            *reinterpret_cast<bf16x4*>(dst + g * 4) = pk;
        }
    }
}

// n consecutive fp32 -> NS bf16 planes (plane stride `plane` elements); NS = 1: plain RNE rounding
template <int NS, int N>
__device__ __forceinline__ void store_planes(__bf16* dst, int plane, const float (&v)[N])
{
    float r[N];
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = v[i];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        float q[N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const __bf16 h = (__bf16)r[i];                   // RNE
            q[i] = r[i];
            if (s + 1 < NS) r[i] = r[i] - (float)h;          // exact: the difference fits the fp32 mantissa
        }
        store_bf16_run<N>(dst + s * plane, q);
    }
}

// The plane pairs of the f32x3 product, smallest terms first.
__device__ constexpr int X3_PA[6] = {2, 0, 1, 1, 0, 0};
__device__ constexpr int X3_PB[6] = {0, 2, 1, 0, 1, 0};

template <int WM, int WN, int BK, int PRO, int EPI, int NS>
__global__ __launch_bounds__(256, NS == 1 ? 4 : 2) void gemm_bf16_kernel(const GemmArgs a)
{
    constexpr int BM = WM * 64, BN = WN * 64;
    constexpr int KA = BM * BK / 256;           // consecutive k per thread, matrix operand
    constexpr int KX = BN * BK / 256;           // consecutive k per thread, streamed operand
    constexpr int PITCH = BK + 8;
    constexpr bool POOL = (PRO == PRO_BN_BWD_POOL);
    constexpr bool TWO = (PRO == PRO_BN_BWD) || POOL;
    constexpr int LDS_BYTES = 2 * NS * (BM + BN) * PITCH * 2;
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
    // [buffer][plane][row][k]
    __bf16 (*As)[NS][BM][PITCH] = reinterpret_cast<__bf16 (*)[NS][BM][PITCH]>(smem);
    __bf16 (*Bs)[NS][BN][PITCH] = reinterpret_cast<__bf16 (*)[NS][BN][PITCH]>(smem + 2 * NS * BM * PITCH * 2);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // same logical tile order and XCD remap as the fp32 kernel
    const int tpc = (a.P + BN - 1) / BN, nmt = (a.M + BM - 1) / BM;
    const int total = a.nb * tpc * nmt;
    int L = blockIdx.x;
    if ((total & 7) == 0) L = (blockIdx.x & 7) * (total >> 3) + (blockIdx.x >> 3);
    const int mt = L % nmt, tn = L / nmt;
    const int b = tn / tpc, pt = tn % tpc;
    const int m0 = mt * BM, p0 = pt * BN;
    const float* Xb = a.X + (long long)b * a.K * a.P;
    const float* X2b = TWO ? a.X2 + (long long)b * a.K * a.P : nullptr;
    const int pgrp = POOL ? a.P / a.pool_group : 0;
    const float* pdp = POOL ? a.pool_dp + (long long)b * a.K * pgrp : nullptr;
    const int* parg = POOL ? a.pool_arg + (long long)b * a.K * pgrp : nullptr;

    // thread -> (row, k-group); the k-group is uniform over a wave (BM, BN >= 64), so everything indexed
    // by k (prologue coefficients) is a scalar load
    const int am = tid % BM, akg = __builtin_amdgcn_readfirstlane(tid / BM) * KA;
    const int xp = tid % BN, xkg = __builtin_amdgcn_readfirstlane(tid / BN) * KX;
    const int amc = min(m0 + am, a.M - 1);
    const int xpc = min(p0 + xp, a.P - 1);
    const bool a_ok = m0 + am < a.M, x_ok = p0 + xp < a.P;
    const int xgrp = POOL ? xpc / a.pool_group : 0, xkin = POOL ? xpc % a.pool_group : 0;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    float ra[KA], rx[KX], ry[TWO ? KX : 1];
    int rarg[POOL ? KX : 1];

    // raw loads from clamped (always valid) addresses; masking and the prologue happen at LDS-store time
    auto load_stage = [&](int k0) {
#pragma unroll
        for (int i = 0; i < KA; ++i) {
            const int kc = min(k0 + akg + i, a.K - 1);
            ra[i] = a.a_trans ? a.At[(long long)amc * a.lda + kc] : a.At[(long long)kc * a.lda + amc];
        }
#pragma unroll
        for (int i = 0; i < KX; ++i) {
            const int kc = min(k0 + xkg + i, a.K - 1);
            const long long off = (long long)kc * a.P + xpc;
            if (POOL) {
                const long long g = (long long)kc * pgrp + xgrp;
                rx[i] = pdp[g];
                rarg[i] = parg[g];
            } else {
                rx[i] = Xb[off];
            }
            if (TWO) ry[i] = X2b[off];
        }
    };
    auto store_stage = [&](int buf, int k0) {
        float va[KA], vx[KX];
#pragma unroll
        for (int i = 0; i < KA; ++i) va[i] = (a_ok && k0 + akg + i < a.K) ? ra[i] : 0.0f;
        store_planes<NS, KA>(&As[buf][0][am][akg], BM * PITCH, va);
#pragma unroll
        for (int i = 0; i < KX; ++i) {
            const bool ok = x_ok && (k0 + xkg + i < a.K);
            float v = rx[i];
            if (PRO != PRO_NONE) {
                const int kc = min(k0 + xkg + i, a.K - 1);
                const float c0 = a.coef[kc], c1 = a.coef[a.K + kc];
                float c2 = 0.f, c3 = 0.f, w = v;
                if (TWO) { c2 = a.coef[2 * a.K + kc]; c3 = a.coef[3 * a.K + kc]; w = ry[i]; }
                if (POOL) v = (rarg[POOL ? i : 0] == xkin) ? v : 0.f;
                constexpr int PA = POOL ? PRO_BN_BWD : PRO;
                v = pro_apply<PA>(v, w, c0, c1, c2, c3);
            }
            vx[i] = ok ? v : 0.0f;
        }
        store_planes<NS, KX>(&Bs[buf][0][xp][xkg], BN * PITCH, vx);
    };

    const int nk = (a.K + BK - 1) / BK;
    load_stage(0);
    store_stage(0, 0);
    __syncthreads();
    int cur = 0;
    const int kh = (lane >> 5) * 8, c = lane & 31;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) load_stage((kt + 1) * BK);          // in flight under the MFMAs below
#pragma unroll
        for (int kk = 0; kk < BK; kk += 16) {
            bf16x8 fa[NS][2], fb[NS][2];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    fa[s][t] = *reinterpret_cast<const bf16x8*>(&As[cur][s][wm * 64 + t * 32 + c][kk + kh]);
                    fb[s][t] = *reinterpret_cast<const bf16x8*>(&Bs[cur][s][wn * 64 + t * 32 + c][kk + kh]);
                }
            }
            // operands swapped: D'[position][channel], see gemm_epilogue
            if constexpr (NS == 1) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[0][0], fa[0][0], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[0][1], fa[0][0], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[0][0], fa[0][1], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[0][1], fa[0][1], acc[1][1], 0, 0, 0);
            } else {
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    const int pa = X3_PA[q], pb = X3_PB[q];
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[pb][0], fa[pa][0], acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[pb][1], fa[pa][0], acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[pb][0], fa[pa][1], acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[pb][1], fa[pa][1], acc[1][1], 0, 0, 0);
                }
            }
        }
        if (kt + 1 < nk) store_stage(cur ^ 1, (kt + 1) * BK);
        __syncthreads();
        cur ^= 1;
    }
    gemm_epilogue<WM, WN, EPI>(a, acc, reinterpret_cast<float*>(smem), LDS_BYTES / 4, b, m0, p0, tn, tpc);
}

template <int WM, int WN, int BK, int NS>
int launch_gemm_bf16_t(const GemmArgs& a, int pro, hipStream_t st)
{
    constexpr int BM = WM * 64, BN = WN * 64;
    const int tpc = (a.P + BN - 1) / BN, nmt = (a.M + BM - 1) / BM;
    const long long total = (long long)a.nb * tpc * nmt;
    if (total > 0x7fffffffLL) return USIP_EINVAL;
    const int epi = a.stats == nullptr ? EPI_NONE : EPI_STATS;
    dim3 grid((unsigned)total), block(256);
#define USIP_GEMM_CASE(P_, E_)                                                                  \
    if (pro == P_ && epi == E_) {                                                               \
        USIP_LAUNCH((gemm_bf16_kernel<WM, WN, BK, P_, E_, NS>), grid, block, 0, st, a);         \
        USIP_LAUNCH_CHECK();                                                                    \
        return USIP_OK;                                                                         \
    }
    USIP_GEMM_CASE(PRO_NONE, EPI_STATS)
    USIP_GEMM_CASE(PRO_NONE, EPI_NONE)
    USIP_GEMM_CASE(PRO_AFFINE_RELU, EPI_STATS)
    USIP_GEMM_CASE(PRO_AFFINE_RELU, EPI_NONE)
    USIP_GEMM_CASE(PRO_BN_BWD, EPI_NONE)
    USIP_GEMM_CASE(PRO_BN_BWD_POOL, EPI_NONE)
#undef USIP_GEMM_CASE
    return USIP_EINVAL;
}

// ------------------------------------------------------------------------------------------------
// weight gradient: both operands are [row][positions] with the contraction (positions) contiguous, so the
// float4 a thread loads becomes four consecutive k of its LDS row directly (no transposing scatter).
template <int TM, int TN, int PRO, bool XPRO, bool VEC, int NS>
__global__ __launch_bounds__(256, NS == 1 ? 1 : 2) void wgrad_bf16_kernel(const WgradArgs a)
{
    // NS = 3 stages 16 positions at a time: three planes of a 32-position stage would leave one workgroup per CU
    constexpr int WN = 2, BM = 2 * TM * 32, BN = 2 * TN * 32, BKP = (NS == 1) ? 32 : 16, PITCH = BKP + 8;
    __shared__ __attribute__((aligned(16))) __bf16 Gs[2][NS][BM][PITCH];
    __shared__ __attribute__((aligned(16))) __bf16 Xs[2][NS][BN][PITCH];
    constexpr int NG4 = BM * BKP / 4 / 256, NX4 = BN * BKP / 4 / 256;

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
    constexpr bool POOL = (PRO == PRO_BN_BWD_POOL);
    constexpr bool TWO = (PRO == PRO_BN_BWD) || POOL;
    const float* Gb = POOL ? nullptr : a.G + (long long)b * a.M * a.P;
    const float* G2b = TWO ? a.G2 + (long long)b * a.M * a.P : nullptr;
    const float* Xb = a.X + (long long)b * a.N * a.P;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int pgrp = POOL ? a.P / a.pool_group : 0;
    float4 rg[NG4], rg2[TWO ? NG4 : 1], rx[NX4], rdummy[1];
    auto load_pool = [&](int p) {
#pragma unroll
        for (int i = 0; i < NG4; ++i) {
            const int f = tid + i * 256, row = f / (BKP / 4), kq = (f % (BKP / 4)) * 4;
            const int rc = min(m0 + row, a.M - 1), pc = min(p + kq, a.P - 4);
            const long long g = ((long long)b * a.M + rc) * pgrp + pc / a.pool_group;
            rg[i] = make_float4(a.pool_dp[g], __int_as_float(a.pool_arg[g]), __int_as_float(pc % a.pool_group), 0.f);
            rg2[i] = *reinterpret_cast<const float4*>(G2b + (long long)rc * a.P + pc);
        }
    };
    auto to_lds = [&](__bf16* dst, int plane, const float4& v) {
        const float q[4] = {v.x, v.y, v.z, v.w};
        store_planes<NS, 4>(dst, plane, q);
    };
    // per-row prologue coefficients, loaded once (a thread's rows are the same in every stage)
    float gc[TWO ? NG4 : 1][4], xc[XPRO ? NX4 : 1][2];
    if (TWO) {
#pragma unroll
        for (int i = 0; i < NG4; ++i) {
            const int ch = min(m0 + (tid + i * 256) / (BKP / 4), a.M - 1);
            gc[i][0] = a.coef[ch]; gc[i][1] = a.coef[a.M + ch];
            gc[i][2] = a.coef[2 * a.M + ch]; gc[i][3] = a.coef[3 * a.M + ch];
        }
    }
    if (XPRO) {
#pragma unroll
        for (int i = 0; i < NX4; ++i) {
            const int ch = min(n0 + (tid + i * 256) / (BKP / 4), a.N - 1);
            xc[i][0] = a.xcoef[ch]; xc[i][1] = a.xcoef[a.N + ch];
        }
    }
    auto store_stage = [&](int buf, int p) {
#pragma unroll
        for (int i = 0; i < NG4; ++i) {
            const int f = tid + i * 256, row = f / (BKP / 4), kq = (f % (BKP / 4)) * 4;
            float4 v = rg[i];
            if (!TWO && VEC && !(m0 + row < a.M && p + kq < pend)) v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (TWO) {
                const bool rok = m0 + row < a.M;
                const float c0 = gc[TWO ? i : 0][0], c1 = gc[TWO ? i : 0][1], c2 = gc[TWO ? i : 0][2],
                            c3 = gc[TWO ? i : 0][3];
                const float4 w = rg2[i];
                if (POOL) {
                    const int hit = __float_as_int(v.y) - __float_as_int(v.z);
                    const float g = v.x;
                    v = make_float4(hit == 0 ? g : 0.f, hit == 1 ? g : 0.f, hit == 2 ? g : 0.f, hit == 3 ? g : 0.f);
                }
                v.x = (rok && p + kq + 0 < pend) ? pro_apply<PRO_BN_BWD>(v.x, w.x, c0, c1, c2, c3) : 0.f;
                v.y = (rok && p + kq + 1 < pend) ? pro_apply<PRO_BN_BWD>(v.y, w.y, c0, c1, c2, c3) : 0.f;
                v.z = (rok && p + kq + 2 < pend) ? pro_apply<PRO_BN_BWD>(v.z, w.z, c0, c1, c2, c3) : 0.f;
                v.w = (rok && p + kq + 3 < pend) ? pro_apply<PRO_BN_BWD>(v.w, w.w, c0, c1, c2, c3) : 0.f;
            }
            to_lds(&Gs[buf][0][row][kq], BM * PITCH, v);
        }
#pragma unroll
        for (int i = 0; i < NX4; ++i) {
            const int f = tid + i * 256, row = f / (BKP / 4), kq = (f % (BKP / 4)) * 4;
            float4 v = rx[i];
            if (XPRO) {
                const float s0 = xc[XPRO ? i : 0][0], s1 = xc[XPRO ? i : 0][1];
                v.x = fmaxf(__builtin_fmaf(v.x, s0, s1), 0.f); v.y = fmaxf(__builtin_fmaf(v.y, s0, s1), 0.f);
                v.z = fmaxf(__builtin_fmaf(v.z, s0, s1), 0.f); v.w = fmaxf(__builtin_fmaf(v.w, s0, s1), 0.f);
                if (!VEC) {
                    if (!(n0 + row < a.N && p + kq + 0 < pend)) v.x = 0.f;
                    if (!(n0 + row < a.N && p + kq + 1 < pend)) v.y = 0.f;
                    if (!(n0 + row < a.N && p + kq + 2 < pend)) v.z = 0.f;
                    if (!(n0 + row < a.N && p + kq + 3 < pend)) v.w = 0.f;
                }
            }
            if (VEC && !(n0 + row < a.N && p + kq < pend)) v = make_float4(0.f, 0.f, 0.f, 0.f);
            to_lds(&Xs[buf][0][row][kq], BN * PITCH, v);
        }
    };

    const int nst = (pend - pbeg + BKP - 1) / BKP;
    if (nst > 0) {
        if (POOL) load_pool(pbeg);
        else wgrad_load_rows<NG4, TWO, VEC, BKP / 4>(Gb, G2b, a.M, a.P, m0, pbeg, pend, tid, rg, rg2);
        wgrad_load_rows<NX4, false, VEC, BKP / 4>(Xb, nullptr, a.N, a.P, n0, pbeg, pend, tid, rx, rdummy);
        store_stage(0, pbeg);
    }
    __syncthreads();
    int cur = 0;
    const int kh = (lane >> 5) * 8, c = lane & 31;
    for (int s = 0; s < nst; ++s) {
        if (s + 1 < nst) {
            if (POOL) load_pool(pbeg + (s + 1) * BKP);
            else wgrad_load_rows<NG4, TWO, VEC, BKP / 4>(Gb, G2b, a.M, a.P, m0, pbeg + (s + 1) * BKP, pend, tid, rg, rg2);
            wgrad_load_rows<NX4, false, VEC, BKP / 4>(Xb, nullptr, a.N, a.P, n0, pbeg + (s + 1) * BKP, pend, tid, rx, rdummy);
        }
        constexpr int NH = BKP / 16;
        bf16x8 fa[NH][NS][TM], fb[NH][NS][TN];          // all 16-position groups of the stage up front
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int pl = 0; pl < NS; ++pl) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    fa[h][pl][i] = *reinterpret_cast<const bf16x8*>(&Gs[cur][pl][(wm * TM + i) * 32 + c][h * 16 + kh]);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    fb[h][pl][j] = *reinterpret_cast<const bf16x8*>(&Xs[cur][pl][(wn * TN + j) * 32 + c][h * 16 + kh]);
            }
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            if constexpr (NS == 1) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[h][0][i], fb[h][0][j], acc[i][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int q = 0; q < 6; ++q)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[h][X3_PA[q]][i], fb[h][X3_PB[q]][j],
                                                                                acc[i][j], 0, 0, 0);
            }
        }
        if (s + 1 < nst) store_stage(cur ^ 1, pbeg + (s + 1) * BKP);
        __syncthreads();
        cur ^= 1;
    }
    wgrad_store_partial<TM, TN>(a, acc, slice, m0, n0, wm, wn, lane);
}

}  // namespace

namespace usip_mlp {

int launch_gemm_bf16(const GemmArgs& a, int pro, hipStream_t st)
{
    // K-step 16 (one MFMA group per stage): 32 was measured slower on every layer shape of the detector
    // (7.70 vs 7.16 ms per step) -- the kernels are bound by L2 -> L1 operand traffic, not by barriers.
    return (a.M <= 64) ? launch_gemm_bf16_t<1, 4, 16, 1>(a, pro, st) : launch_gemm_bf16_t<2, 2, 16, 1>(a, pro, st);
}

int launch_gemm_x3(const GemmArgs& a, int pro, hipStream_t st)
{
    return launch_gemm_bf16_t<2, 2, 16, 3>(a, pro, st);
}

template <int NS>
static int launch_wgrad_planes(const WgradArgs& a, int pro, bool xpro, bool vec, int small, unsigned blocks, hipStream_t st)
{
    dim3 grid(blocks), block(256);
#define USIP_WGRAD_CASE(T_, P_, X_, V_)                                                        \
    if (small == (T_ == 1) && pro == P_ && xpro == X_ && vec == V_) {                          \
        USIP_LAUNCH((wgrad_bf16_kernel<T_, T_, P_, X_, V_, NS>), grid, block, 0, st, a);       \
        USIP_LAUNCH_CHECK();                                                                   \
        return USIP_OK;                                                                        \
    }
#define USIP_WGRAD_CASES(T_, P_) \
    USIP_WGRAD_CASE(T_, P_, false, true) USIP_WGRAD_CASE(T_, P_, false, false) \
    USIP_WGRAD_CASE(T_, P_, true, true) USIP_WGRAD_CASE(T_, P_, true, false)
    if constexpr (NS == 1) {
        USIP_WGRAD_CASES(1, PRO_NONE)
        USIP_WGRAD_CASES(1, PRO_BN_BWD)
        USIP_WGRAD_CASE(1, PRO_BN_BWD_POOL, false, true) USIP_WGRAD_CASE(1, PRO_BN_BWD_POOL, true, true)
    }
    USIP_WGRAD_CASES(2, PRO_NONE)
    USIP_WGRAD_CASES(2, PRO_BN_BWD)
    USIP_WGRAD_CASE(2, PRO_BN_BWD_POOL, false, true) USIP_WGRAD_CASE(2, PRO_BN_BWD_POOL, true, true)
#undef USIP_WGRAD_CASES
#undef USIP_WGRAD_CASE
    return USIP_EINVAL;
}

int launch_wgrad_bf16(const WgradArgs& a, int pro, bool xpro, bool vec, int small, unsigned blocks, hipStream_t st)
{
    return launch_wgrad_planes<1>(a, pro, xpro, vec, small, blocks, st);
}

int launch_wgrad_x3(const WgradArgs& a, int pro, bool xpro, bool vec, unsigned blocks, hipStream_t st)
{
    return launch_wgrad_planes<3>(a, pro, xpro, vec, 0, blocks, st);
}

}  // namespace usip_mlp
