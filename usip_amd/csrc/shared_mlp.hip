// usip_amd/csrc/shared_mlp.hip -- the shared-MLP (1x1 convolution + BatchNorm + ReLU) of the
// USIP detector as hand-written fp32-MFMA kernels for gfx950.
//
// Reference: models/layers.py:208-216 (MyConv2d.forward), :293-303 (EquivariantLayer.forward):
// cuDNN/MIOpen convolution + native batch-norm + clamp as three kernels per layer forward and
// ~six per layer backward, every one a full round trip over B x C x M x K activations.
//
// Layout.  Activations stay in the reference's channel-major layout [cloud][C][P] (P = M*K
// grouped positions, or N points): for  Y[b] = W . X[b]  the positions are the GEMM's N
// dimension and are contiguous, so the MFMA B operand (v_mfma_f32_32x32x2_f32: lane l holds
// B[k = l>>5][n = l&31]) and the C/D rows are read and written as 128-B coalesced segments
// with no transposes anywhere.  The matrix operand is handed over K-major ("At" = [K][M]).
//
// Kernels
//   gemm_kernel     Y = At^T . pro(X) + bias, epilogue: per-channel (sum, sum^2) partials of Y for
//                   the BatchNorm that follows (deterministic: one partial per position tile).
//                   pro = identity | relu(x*s+t) | BatchNorm-backward of (dZ, Y) -> dY.
//                   The same kernel is the forward GEMM (At = W^T) and the data-gradient GEMM
//                   (At = W, X = dZ/Y of the layer's output).
//   wgrad_kernel    dW[co][ci] = sum_p pro(G)[co][p] * X[ci][p]: split over position ranges,
//                   partial tiles to a workspace, summed in fixed order by wgrad_reduce_kernel.
//   bn_* kernels    statistics finalisation (+ running stats), BN+ReLU apply, backward reductions.
//
// fp32 in, fp32 accumulate on the matrix cores: v_mfma_f32_32x32x2_f32 is bit-for-bit an fmaf
// chain, so parity with the fp32 reference is a matter of summation order only (<= 1e-6).
#include "mlp_common.h"

using namespace usip_mlp;

namespace {

// TM = 32-row MFMA tiles per wave: 2 (64 x 64 per wave, the default) or 1 (32 x 64 per wave, block 64 x 128) for
// launches whose 128-row tiling would leave most CUs idle -- a workgroup's time is its waves' K-loop, so the only
// way to shorten an under-filled launch is less work per wave (tools/small_gemm_bench.py).
template <int WM, int WN, int BK, int PRO, int EPI, bool VEC, int TM>
__global__ __launch_bounds__(256, 4) void gemm_kernel(const GemmArgs a)
{
    constexpr int BM = WM * 32 * TM, BN = WN * 64;
    constexpr int NA = BM * BK / 256;           // A elements per thread per stage
    constexpr int NB4 = BK * BN / 4 / 256;      // X float4 per thread per stage (VEC)
    constexpr int NBS = BK * BN / 256;          // X scalars per thread per stage (!VEC)
    constexpr bool POOL = (PRO == PRO_BN_BWD_POOL);
    constexpr bool TWO = (PRO == PRO_BN_BWD) || POOL;   // second streamed tensor (the layer's pre-BN output)
    __shared__ __attribute__((aligned(16))) float As[2][BK][BM];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][BN];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // logical tile id: co-tiles of one position tile are consecutive; XCD-aware remap so that
    // they also land on ONE XCD (shared L2 for the X tile) -- speed only, never correctness.
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

    f32x16 acc[TM][2];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // Raw register staging: the loads of stage t+1 are issued before the MFMAs of stage t and are
    // first touched (prologue math + ds_write) after them, so their latency hides under the MFMAs.
    float ra[NA];
    float4 rx[VEC ? NB4 : 1], ry[(VEC && TWO) ? NB4 : 1];
    // BN+ReLU prologue: the float4's channel coefficients are fetched WITH the operand, a stage ahead (loaded
    // where they are used they put a vector-load round trip between the MFMAs and the LDS refill of every
    // stage).  Not for the BatchNorm-backward prologues: their four coefficients per float4 push the kernel over
    // the 128-VGPR budget (24-180 B of scratch, the narrow data-gradient GEMMs 2x slower).
    constexpr bool RC = VEC && (PRO == PRO_AFFINE_RELU);
    float rc[RC ? NB4 : 1][2];
    // BatchNorm-backward prologues (four coefficients per channel): a wave's 64 float4 cover KW = 256 / BN
    // consecutive K rows (lanes [32h', ...) of a 128-wide tile, one row of a 256-wide tile), so the coefficients
    // are fetched with SCALAR loads -- a stage ahead like the operands, in SGPRs, at no VGPR cost -- and picked
    // per lane with one select each.
    constexpr int KW = (64 * 4) / BN;                        // K rows per wave per float4 index: 2 or 1
    constexpr bool SC = VEC && TWO;
    float sc[SC ? NB4 : 1][4][KW];
    float rxs[VEC ? 1 : NBS], rys[(!VEC && TWO) ? NBS : 1];

    auto load_stage = [&](int k0) {
        // branch-free: out-of-range elements load from a clamped (valid) address and are zeroed when
        // the registers are written to LDS (store_stage), so the loads issue back to back and nothing
        // touches their results until after the MFMAs
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int e = tid + i * 256, k = e / BM, m = e % BM;
            const int kc = min(k0 + k, a.K - 1), mc = min(m0 + m, a.M - 1);
            ra[i] = a.a_trans ? a.At[(long long)mc * a.lda + kc] : a.At[(long long)kc * a.lda + mc];
        }
        if (VEC) {
#pragma unroll
            for (int i = 0; i < NB4; ++i) {
                const int f = tid + i * 256, k = f / (BN / 4), col = (f % (BN / 4)) * 4;
                const int kc = min(k0 + k, a.K - 1), pc = min(p0 + col, a.P - 4);
                const long long off = (long long)kc * a.P + pc;
                if (POOL) {
                    // raw (dpooled, arg) of this float4's neighbourhood; dZ is formed when the registers go to LDS
                    const long long g = (long long)kc * pgrp + pc / a.pool_group;
                    rx[i] = make_float4(pdp[g], __int_as_float(parg[g]), __int_as_float(pc % a.pool_group), 0.f);
                } else {
                    rx[i] = *reinterpret_cast<const float4*>(Xb + off);
                }
                if (TWO) ry[i] = *reinterpret_cast<const float4*>(X2b + off);
                if (RC) { rc[i][0] = a.coef[kc]; rc[i][1] = a.coef[a.K + kc]; }
                if (SC) {
                    const int kw = __builtin_amdgcn_readfirstlane(k);     // the wave's first K row for this i
#pragma unroll
                    for (int h = 0; h < KW; ++h) {
                        const int ks = min(k0 + kw + h, a.K - 1);
#pragma unroll
                        for (int j = 0; j < 4; ++j) sc[i][j][h] = a.coef[j * a.K + ks];
                    }
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < NBS; ++i) {
                const int e = tid + i * 256, k = e / BN, col = e % BN;
                float v = 0.f, w = 0.f;
                if (k0 + k < a.K && p0 + col < a.P) {
                    const long long off = (long long)(k0 + k) * a.P + p0 + col;
                    v = Xb[off];
                    if (TWO) w = X2b[off];
                }
                rxs[i] = v;
                if (TWO) rys[i] = w;
            }
        }
    };
    auto store_stage = [&](int buf, int k0) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int e = tid + i * 256, k = e / BM, m = e % BM;
            As[buf][k][m] = (k0 + k < a.K && m0 + m < a.M) ? ra[i] : 0.0f;
        }
        if (VEC) {
#pragma unroll
            for (int i = 0; i < NB4; ++i) {
                const int f = tid + i * 256, k = f / (BN / 4), col = (f % (BN / 4)) * 4;
                float4 v = rx[i];
                const bool ok = (k0 + k < a.K && p0 + col < a.P);
                if (PRO == PRO_NONE) {
                    if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
                } else {
                    const int kc = min(k0 + k, a.K - 1);
                    float c0, c1, c2 = 0.f, c3 = 0.f;
                    float4 w = v;
                    if (SC) {
                        const bool hi = KW > 1 && (lane >= 32);          // second K row of the wave
                        c0 = hi ? sc[SC ? i : 0][0][KW - 1] : sc[SC ? i : 0][0][0];
                        c1 = hi ? sc[SC ? i : 0][1][KW - 1] : sc[SC ? i : 0][1][0];
                        c2 = hi ? sc[SC ? i : 0][2][KW - 1] : sc[SC ? i : 0][2][0];
                        c3 = hi ? sc[SC ? i : 0][3][KW - 1] : sc[SC ? i : 0][3][0];
                        w = ry[i];
                    } else {
                        c0 = RC ? rc[RC ? i : 0][0] : a.coef[kc];
                        c1 = RC ? rc[RC ? i : 0][1] : a.coef[a.K + kc];
                        if (TWO) { c2 = a.coef[2 * a.K + kc]; c3 = a.coef[3 * a.K + kc]; w = ry[i]; }
                    }
                    if (POOL) {
                        const int hit = __float_as_int(v.y) - __float_as_int(v.z);   // arg - (first k of the float4)
                        const float g = v.x;
                        v = make_float4(hit == 0 ? g : 0.f, hit == 1 ? g : 0.f, hit == 2 ? g : 0.f, hit == 3 ? g : 0.f);
                    }
                    constexpr int PA = POOL ? PRO_BN_BWD : PRO;
                    v.x = ok ? pro_apply<PA>(v.x, w.x, c0, c1, c2, c3) : 0.f;
                    v.y = ok ? pro_apply<PA>(v.y, w.y, c0, c1, c2, c3) : 0.f;
                    v.z = ok ? pro_apply<PA>(v.z, w.z, c0, c1, c2, c3) : 0.f;
                    v.w = ok ? pro_apply<PA>(v.w, w.w, c0, c1, c2, c3) : 0.f;
                }
                *reinterpret_cast<float4*>(&Bs[buf][k][col]) = v;
            }
        } else {
#pragma unroll
            for (int i = 0; i < NBS; ++i) {
                const int e = tid + i * 256, k = e / BN, col = e % BN;
                float v = rxs[i];
                if (PRO != PRO_NONE) {
                    const bool ok = (k0 + k < a.K && p0 + col < a.P);
                    const int kc = min(k0 + k, a.K - 1);
                    const float c0 = a.coef[kc], c1 = a.coef[a.K + kc];
                    float c2 = 0.f, c3 = 0.f, w = v;
                    if (TWO) { c2 = a.coef[2 * a.K + kc]; c3 = a.coef[3 * a.K + kc]; w = rys[i]; }
                    v = ok ? pro_apply<PRO>(v, w, c0, c1, c2, c3) : 0.f;
                }
                Bs[buf][k][col] = v;
            }
        }
    };

    const int nk = (a.K + BK - 1) / BK;
    load_stage(0);
    store_stage(0, 0);
    __syncthreads();
    int cur = 0;
    const int kr = lane >> 5, c = lane & 31;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) load_stage((kt + 1) * BK);         // in flight under the MFMAs below
        // fragments of step kk+2 are read from LDS while the four MFMAs of step kk run
        float fa[TM], fb0 = Bs[cur][kr][wn * 64 + c], fb1 = Bs[cur][kr][wn * 64 + 32 + c];
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = As[cur][kr][(wm * TM + i) * 32 + c];
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float av[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) av[i] = fa[i];
            const float b0 = fb0, b1 = fb1;
            if (kk + 2 < BK) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[i] = As[cur][kk + 2 + kr][(wm * TM + i) * 32 + c];
                fb0 = Bs[cur][kk + 2 + kr][wn * 64 + c];
                fb1 = Bs[cur][kk + 2 + kr][wn * 64 + 32 + c];
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                // operands swapped: D'[position][channel], see gemm_epilogue
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0, av[i], acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b1, av[i], acc[i][1], 0, 0, 0);
            }
            // pin the order: next step's LDS reads first, then this step's MFMAs cover them
            __builtin_amdgcn_sched_group_barrier(0x100, TM + 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * TM, 0);
        }
        if (kt + 1 < nk) store_stage(cur ^ 1, (kt + 1) * BK);
        __syncthreads();
        cur ^= 1;
    }

    gemm_epilogue<WM, WN, EPI, TM>(a, acc, &As[0][0][0], 2 * BK * BM, b, m0, p0, tn, tpc);   // LDS is free now
}

template <int WM, int WN, int BK, int TM>
int launch_gemm(const GemmArgs& a, int pro, hipStream_t st)
{
    constexpr int BM = WM * 32 * TM, BN = WN * 64;
    const int tpc = (a.P + BN - 1) / BN, nmt = (a.M + BM - 1) / BM;
    const long long total = (long long)a.nb * tpc * nmt;
    if (total > 0x7fffffffLL) return USIP_EINVAL;
    const bool vec = (a.P % 4 == 0) && (pro == PRO_BN_BWD_POOL || (reinterpret_cast<uintptr_t>(a.X) & 15u) == 0) &&
                     ((pro != PRO_BN_BWD && pro != PRO_BN_BWD_POOL) || (reinterpret_cast<uintptr_t>(a.X2) & 15u) == 0);
    const int epi = a.stats == nullptr ? EPI_NONE : EPI_STATS;
    dim3 grid((unsigned)total), block(256);
#define USIP_GEMM_CASE(P_, E_, V_)                                                              \
    if (pro == P_ && epi == E_ && vec == V_) {                                                  \
        USIP_LAUNCH((gemm_kernel<WM, WN, BK, P_, E_, V_, TM>), grid, block, 0, st, a);          \
        USIP_LAUNCH_CHECK();                                                                    \
        return USIP_OK;                                                                         \
    }
    USIP_GEMM_CASE(PRO_NONE, EPI_STATS, true)
    USIP_GEMM_CASE(PRO_NONE, EPI_STATS, false)
    USIP_GEMM_CASE(PRO_NONE, EPI_NONE, true)
    USIP_GEMM_CASE(PRO_NONE, EPI_NONE, false)
    USIP_GEMM_CASE(PRO_AFFINE_RELU, EPI_STATS, true)
    USIP_GEMM_CASE(PRO_AFFINE_RELU, EPI_STATS, false)
    USIP_GEMM_CASE(PRO_AFFINE_RELU, EPI_NONE, true)
    USIP_GEMM_CASE(PRO_AFFINE_RELU, EPI_NONE, false)
    USIP_GEMM_CASE(PRO_BN_BWD, EPI_NONE, true)
    USIP_GEMM_CASE(PRO_BN_BWD, EPI_NONE, false)
    USIP_GEMM_CASE(PRO_BN_BWD_POOL, EPI_NONE, true)
#undef USIP_GEMM_CASE
    return USIP_EINVAL;
}

template <int TM, int TN, int PRO, bool XPRO, bool VEC>
__global__ __launch_bounds__(256, 4) void wgrad_kernel(const WgradArgs a)
{
    // 2 x 2 waves, each TM x TN MFMA tiles of 32 x 32: block tile 128 x 128 (TM = TN = 2) or 64 x 64
    constexpr int WN = 2, BM = 2 * TM * 32, BN = 2 * TN * 32, BKP = 16;
    constexpr int LDM = BM + 2, LDN = BN + 2;
    __shared__ float Gs[2][BKP][LDM];
    __shared__ float Xs[2][BKP][LDN];
    constexpr int NG4 = BM * BKP / 4 / 256, NX4 = BN * BKP / 4 / 256;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int nmt = (a.M + BM - 1) / BM, nnt = (a.N + BN - 1) / BN;
    // All output tiles of one position slice stream the SAME rows of G and X in lockstep: put them on
    // ONE XCD (workgroup id mod 8 selects the XCD) so that the re-reads are L2 hits instead of
    // 8 separate fetches.  Speed only; any placement gives the same result.
    const int total = gridDim.x;
    int L = blockIdx.x;
    if ((total & 7) == 0) L = (blockIdx.x & 7) * (total >> 3) + (blockIdx.x >> 3);
    const int tile = L % (nmt * nnt), slice = L / (nmt * nnt);
    const int m0 = (tile / nnt) * BM, n0 = (tile % nnt) * BN;
    const int b = slice / a.segs, seg = slice % a.segs;
    const int pbeg = seg * a.seglen, pend = min(a.P, pbeg + a.seglen);
    const float* Gb = (PRO == PRO_BN_BWD_POOL) ? nullptr : a.G + (long long)b * a.M * a.P;
    const float* G2b = (PRO == PRO_BN_BWD || PRO == PRO_BN_BWD_POOL) ? a.G2 + (long long)b * a.M * a.P : nullptr;
    const float* Xb = a.X + (long long)b * a.N * a.P;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    constexpr bool POOL = (PRO == PRO_BN_BWD_POOL);
    constexpr bool TWO = (PRO == PRO_BN_BWD) || POOL;
    const int pgrp = POOL ? a.P / a.pool_group : 0;
    float4 rg[NG4], rg2[TWO ? NG4 : 1], rx[NX4], rdummy[1];
    // POOL: raw (dpooled, arg, first k) of the float4's neighbourhood instead of a dZ load
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
    // A thread's rows are the same in every stage (row = f / 8): their per-channel prologue coefficients are
    // loaded ONCE here, not in every stage where the vector loads' latency would sit between the MFMAs and the
    // LDS refill.
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
            Gs[buf][kq][row] = v.x; Gs[buf][kq + 1][row] = v.y;
            Gs[buf][kq + 2][row] = v.z; Gs[buf][kq + 3][row] = v.w;
        }
#pragma unroll
        for (int i = 0; i < NX4; ++i) {
            const int f = tid + i * 256, row = f / (BKP / 4), kq = (f % (BKP / 4)) * 4;
            float4 v = rx[i];
            if (XPRO) {
                const float s0 = xc[XPRO ? i : 0][0], s1 = xc[XPRO ? i : 0][1];
                v.x = fmaxf(__builtin_fmaf(v.x, s0, s1), 0.f); v.y = fmaxf(__builtin_fmaf(v.y, s0, s1), 0.f);
                v.z = fmaxf(__builtin_fmaf(v.z, s0, s1), 0.f); v.w = fmaxf(__builtin_fmaf(v.w, s0, s1), 0.f);
                if (!VEC) {                                  // scalar path zero-filled invalid lanes before the affine
                    if (!(n0 + row < a.N && p + kq + 0 < pend)) v.x = 0.f;
                    if (!(n0 + row < a.N && p + kq + 1 < pend)) v.y = 0.f;
                    if (!(n0 + row < a.N && p + kq + 2 < pend)) v.z = 0.f;
                    if (!(n0 + row < a.N && p + kq + 3 < pend)) v.w = 0.f;
                }
            }
            if (VEC && !(n0 + row < a.N && p + kq < pend)) v = make_float4(0.f, 0.f, 0.f, 0.f);
            Xs[buf][kq][row] = v.x; Xs[buf][kq + 1][row] = v.y;
            Xs[buf][kq + 2][row] = v.z; Xs[buf][kq + 3][row] = v.w;
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
    const int kr = lane >> 5, c = lane & 31;
    for (int s = 0; s < nst; ++s) {
        if (s + 1 < nst) {
            if (POOL) load_pool(pbeg + (s + 1) * BKP);
            else wgrad_load_rows<NG4, TWO, VEC, BKP / 4>(Gb, G2b, a.M, a.P, m0, pbeg + (s + 1) * BKP, pend, tid, rg, rg2);
            wgrad_load_rows<NX4, false, VEC, BKP / 4>(Xb, nullptr, a.N, a.P, n0, pbeg + (s + 1) * BKP, pend, tid, rx, rdummy);
        }
        float fa[TM], fb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = Gs[cur][kr][(wm * TM + i) * 32 + c];
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[j] = Xs[cur][kr][(wn * TN + j) * 32 + c];
#pragma unroll
        for (int kk = 0; kk < BKP; kk += 2) {
            float av[TM], bv[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) av[i] = fa[i];
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[j] = fb[j];
            if (kk + 2 < BKP) {                       // next step's fragments, under this step's MFMAs
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[i] = Gs[cur][kk + 2 + kr][(wm * TM + i) * 32 + c];
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[j] = Xs[cur][kk + 2 + kr][(wn * TN + j) * 32 + c];
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
        }
        if (s + 1 < nst) store_stage(cur ^ 1, pbeg + (s + 1) * BKP);
        __syncthreads();
        cur ^= 1;
    }
    wgrad_store_partial<TM, TN>(a, acc, slice, m0, n0, wm, wn, lane);
}

// dW[e] = sum over slices, in a fixed order: 64 elements x Q slice lanes per block, every lane sums its share of the
// slices (4 independent chains), the shares are combined through LDS in a fixed order.  Q = 16 for the launches with
// many slices (one tile of 64 x 7 .. 128 x 128 outputs cut into ~1024 position slices): with Q = 4 every lane walked
// 256 slices and the kernel was that chain of dependent loads (23 us for 1.8 MB at 64 x 7, 31 us at 128 x 128).
template <int Q>
__device__ __forceinline__ void wgrad_reduce_body(const float* __restrict__ part, float* __restrict__ dW, long long elems,
                                                  int slices, int N, int ldw, int coloff, long long block)
{
    __shared__ float red[Q][64];
    const int e = threadIdx.x & 63, q = threadIdx.x >> 6;
    const long long i = block * 64 + e;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < elems) {
        const int per = (slices + Q - 1) / Q, k0 = q * per, k1 = min(slices, k0 + per);
        int k = k0;
        for (; k + 3 < k1; k += 4) {
            s0 += part[(long long)k * elems + i];
            s1 += part[(long long)(k + 1) * elems + i];
            s2 += part[(long long)(k + 2) * elems + i];
            s3 += part[(long long)(k + 3) * elems + i];
        }
        for (; k < k1; ++k) s0 += part[(long long)k * elems + i];
    }
    red[q][e] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (q == 0 && i < elems) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < Q; g += 4) t += (red[g][e] + red[g + 1][e]) + (red[g + 2][e] + red[g + 3][e]);
        dW[(i / N) * ldw + coloff + (i % N)] = t;
    }
}

template <int Q>
__global__ __launch_bounds__(64 * Q) void wgrad_reduce_kernel(const float* __restrict__ part,
                                                              float* __restrict__ dW, long long elems, int slices,
                                                              int N, int ldw, int coloff)
{
    wgrad_reduce_body<Q>(part, dW, elems, slices, N, ldw, coloff, blockIdx.x);
}

// Round 5: ALL reductions of a backward pass in one launch per Q.  A training step's weight gradients are read by the
// optimizer (or the all-reduce) only, so the fifteen fixed-order sums of partial tiles it used to launch one by one
// behind their producers (5-12 us each, most of it the launch) are recorded (usip_wgrad_defer) and issued together at
// the end of backward (usip_wgrad_flush).  The job table travels BY VALUE in the kernel arguments, so a captured HIP
// graph holds it; every block finds its job by a scan over at most USIP_REDUCE_JOBS block offsets.  Same body, same
// per-lane slice shares, same summation order as wgrad_reduce_kernel<Q>: the same bits.
constexpr int USIP_REDUCE_JOBS = 24;
struct ReduceJob { const float* part; float* dW; long long elems; int slices, N, ldw, coloff, block0, pad; };
struct ReduceJobs { int n, pad; ReduceJob j[USIP_REDUCE_JOBS]; };

template <int Q>
__global__ __launch_bounds__(64 * Q) void wgrad_reduce_multi_kernel(const ReduceJobs J)
{
    int k = 0;
    for (int t = 1; t < J.n; ++t) if ((int)blockIdx.x >= J.j[t].block0) k = t;
    const ReduceJob jb = J.j[k];
    wgrad_reduce_body<Q>(jb.part, jb.dW, jb.elems, jb.slices, jb.N, jb.ldw, jb.coloff, (long long)blockIdx.x - jb.block0);
}

// ------------------------------------------------------------------------------------------------
// BatchNorm helpers.  One wave per channel; partials are summed in double in a fixed order.
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    return v;
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(
    const float* __restrict__ stats, int ntn, int C, double count, const float* __restrict__ gamma,
    const float* __restrict__ beta, float eps, float momentum, float* __restrict__ running_mean,
    float* __restrict__ running_var, float* __restrict__ mean_out, float* __restrict__ invstd_out,
    float* __restrict__ coef)
{
    // one workgroup of four waves per channel (a single wave spent 26 us on the 4096 partials of the widest
    // layers, latency bound); fixed partition and fixed combination order, so the result is deterministic
    __shared__ double red[2][4];
    const int ch = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double s = 0.0, q = 0.0;
    const float* ps = stats + (long long)ch * ntn;
    const float* pq = stats + (long long)ntn * C + (long long)ch * ntn;
    int t = threadIdx.x;
    // eight loads in flight per pass (the kernel is a chain of load latencies: 5-6 us for 16 serial passes); the order
    // of the additions is the same as with one load per pass
    for (; t + 768 < ntn; t += 1024) {
        const float s0 = ps[t], s1 = ps[t + 256], s2 = ps[t + 512], s3 = ps[t + 768];
        const float q0 = pq[t], q1 = pq[t + 256], q2 = pq[t + 512], q3 = pq[t + 768];
        s += (double)s0; s += (double)s1; s += (double)s2; s += (double)s3;
        q += (double)q0; q += (double)q1; q += (double)q2; q += (double)q3;
    }
    for (; t < ntn; t += 256) { s += (double)ps[t]; q += (double)pq[t]; }
    s = wave_sum(s); q = wave_sum(q);
    if (lane == 0) { red[0][wave] = s; red[1][wave] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        s = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        q = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        const double mean = s / count;
        double var = q / count - mean * mean;                 // biased, as F.batch_norm normalises
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        const float g = gamma ? gamma[ch] : 1.0f, bt = beta ? beta[ch] : 0.0f;
        const float sc = g * invstd;
        mean_out[ch] = (float)mean;
        invstd_out[ch] = invstd;
        coef[ch] = sc;                                        // z = relu(fma(y, sc, sh))
        coef[C + ch] = bt - (float)mean * sc;
        coef[2 * C + ch] = (float)mean;                       // rows 2, 3: what a consumer's backward epilogue
        coef[3 * C + ch] = invstd;                            // needs to form yhat
        if (running_mean) {
            const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
            running_mean[ch] = (1.0f - momentum) * running_mean[ch] + momentum * (float)mean;
            running_var[ch] = (1.0f - momentum) * running_var[ch] + momentum * (float)unbiased;
        }
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ Y, const float* __restrict__ coef,
                                                       float* __restrict__ Z, int relu, int C, int P)
{
    const long long rowid = blockIdx.y;                       // b*C + c
    const int ch = (int)(rowid % C);
    const float sc = coef[ch], sh = coef[C + ch];
    const float* y = Y + rowid * P;
    float* z = Z + rowid * P;
    if (VEC) {
        const int p = (blockIdx.x * 256 + threadIdx.x) * 4;
        if (p >= P) return;
        float4 v = *reinterpret_cast<const float4*>(y + p);
        v.x = __builtin_fmaf(v.x, sc, sh); v.y = __builtin_fmaf(v.y, sc, sh);
        v.z = __builtin_fmaf(v.z, sc, sh); v.w = __builtin_fmaf(v.w, sc, sh);
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        *reinterpret_cast<float4*>(z + p) = v;
    } else {
        const int p = blockIdx.x * 256 + threadIdx.x;
        if (p >= P) return;
        float v = __builtin_fmaf(y[p], sc, sh);
        z[p] = relu ? fmaxf(v, 0.f) : v;
    }
}

// Per (cloud, channel) row: s1 = sum dYhat, s2 = sum dYhat * yhat with
// dYhat = dZ * [fma(y, sc, sh) > 0] (or dZ when !relu), yhat = (y - mean) * invstd.
// plain (no BN): s1 = sum dZ only (bias gradient of a layer without normalisation).
// GROUP (K consecutive positions form one neighbourhood, K % 4 == 0, K/4 a power of two <= 64):
// also gsum[0][row][m] = sum_k dYhat, gsum[1][row][m] = sum_k y -- what the pooled-concat layer
// needs to push the gradient through its broadcast input without a second pass.
template <bool GROUP, int UNR>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(
    const float* __restrict__ dZ, const float* __restrict__ Y, const float* __restrict__ coef,
    const float* __restrict__ mean, const float* __restrict__ invstd, float* __restrict__ partial,
    float* __restrict__ gsum, int relu, int plain, int C, int P, int nrows, int K, int want_max)
{
    __shared__ float red[3][4];
    const long long rowid = blockIdx.x;
    const int ch = (int)(rowid % C);
    const float* dz = dZ + rowid * P;
    float s1 = 0.f, s2 = 0.f, mx = 0.f;                      // mx: max |dYhat| of the row (want_max: partial[2][row])
    if (plain) {
        for (int p = threadIdx.x; p < P; p += 256) s1 += dz[p];
    } else if (GROUP) {
        const float* y = Y + rowid * P;
        const float sc = coef[ch], sh = coef[C + ch], mu = mean[ch], is = invstd[ch];
        const int lpg = K / 4;                                   // lanes per group
        const int G = P / K;
        float* g0 = gsum + rowid * G;
        float* g1 = gsum + ((long long)nrows + rowid) * G;
        // UNR iterations' loads (UNR x 2 x 16 B per lane; UNR = 2: 4 KiB per wave) are in flight before the first is used -- round 5: the
        // loop issued one pair of loads and waited for it (vmcnt(0)) eight times per row; the arithmetic and its order are
        // unchanged (same bits).  Ordinary (temporal) loads: the pair (dZ, Y) is next read by the data / weight gradient kernels.
        const int pend4 = ((P + 1023) / 1024) * 1024;
        for (int p0 = threadIdx.x * 4; p0 < pend4; p0 += UNR * 1024) {
            float4 yq[UNR], dq[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int p = min(p0 + u * 1024, P - 4);         // clamped: branch-free loads (P % 4 == 0)
                yq[u] = ld_in4<LD_BN_REDUCE_Y>(y + p);
                dq[u] = ld_in4<LD_BN_REDUCE_Z>(dz + p);
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int p = p0 + u * 1024;
                if (p >= pend4) break;
                float gd = 0.f, gy = 0.f;
                if (p < P) {
                    const float4 yv = yq[u];
                    const float4 dv = dq[u];
                    const float d0 = (!relu || __builtin_fmaf(yv.x, sc, sh) > 0.f) ? dv.x : 0.f;
                    const float d1 = (!relu || __builtin_fmaf(yv.y, sc, sh) > 0.f) ? dv.y : 0.f;
                    const float d2 = (!relu || __builtin_fmaf(yv.z, sc, sh) > 0.f) ? dv.z : 0.f;
                    const float d3 = (!relu || __builtin_fmaf(yv.w, sc, sh) > 0.f) ? dv.w : 0.f;
                    gd = (d0 + d1) + (d2 + d3);
                    gy = (yv.x + yv.y) + (yv.z + yv.w);
                    mx = fmaxf(fmaxf(mx, fmaxf(fabsf(d0), fabsf(d1))), fmaxf(fabsf(d2), fabsf(d3)));
                    s1 += gd;
                    s2 = __builtin_fmaf(d0, (yv.x - mu) * is, s2);
                    s2 = __builtin_fmaf(d1, (yv.y - mu) * is, s2);
                    s2 = __builtin_fmaf(d2, (yv.z - mu) * is, s2);
                    s2 = __builtin_fmaf(d3, (yv.w - mu) * is, s2);
                }
                for (int off = lpg / 2; off > 0; off >>= 1) {    // lpg lanes = one neighbourhood
                    gd += __shfl_xor(gd, off);
                    gy += __shfl_xor(gy, off);
                }
                if (p < P && (threadIdx.x % lpg) == 0) { g0[p / K] = gd; g1[p / K] = gy; }
            }
        }
    } else {
        const float* y = Y + rowid * P;
        const float sc = coef[ch], sh = coef[C + ch], mu = mean[ch], is = invstd[ch];
        const bool vec = (P % 4 == 0) && (((reinterpret_cast<uintptr_t>(dZ) | reinterpret_cast<uintptr_t>(Y)) & 15u) == 0);
        if (vec) {
            // (two iterations' loads in flight, as in the GROUP form above; same arithmetic, same order)
            for (int p0 = threadIdx.x * 4; p0 < P; p0 += UNR * 1024) {
                float4 yq[UNR], dq[UNR];
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int p = min(p0 + u * 1024, P - 4);     // clamped: branch-free loads
                    yq[u] = ld_in4<LD_BN_REDUCE_Y>(y + p);
                    dq[u] = ld_in4<LD_BN_REDUCE_Z>(dz + p);
                }
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    // a sub-iteration beyond the row adds exact zeros (the clamped load returned finite data): no branch
                    // for the compiler to sink the second pair of loads into
                    const bool in = p0 + u * 1024 < P;
                    const float4 yv = yq[u];
                    const float4 dv = dq[u];
                    const float d0 = (in && (!relu || __builtin_fmaf(yv.x, sc, sh) > 0.f)) ? dv.x : 0.f;
                    const float d1 = (in && (!relu || __builtin_fmaf(yv.y, sc, sh) > 0.f)) ? dv.y : 0.f;
                    const float d2 = (in && (!relu || __builtin_fmaf(yv.z, sc, sh) > 0.f)) ? dv.z : 0.f;
                    const float d3 = (in && (!relu || __builtin_fmaf(yv.w, sc, sh) > 0.f)) ? dv.w : 0.f;
                    mx = fmaxf(fmaxf(mx, fmaxf(fabsf(d0), fabsf(d1))), fmaxf(fabsf(d2), fabsf(d3)));
                    s1 += (d0 + d1) + (d2 + d3);
                    s2 = __builtin_fmaf(d0, (yv.x - mu) * is, s2);
                    s2 = __builtin_fmaf(d1, (yv.y - mu) * is, s2);
                    s2 = __builtin_fmaf(d2, (yv.z - mu) * is, s2);
                    s2 = __builtin_fmaf(d3, (yv.w - mu) * is, s2);
                }
            }
        } else {
            for (int p = threadIdx.x; p < P; p += 256) {
                const float yv = y[p];
                const float d = (!relu || __builtin_fmaf(yv, sc, sh) > 0.f) ? dz[p] : 0.f;
                mx = fmaxf(mx, fabsf(d));
                s1 += d;
                s2 = __builtin_fmaf(d, (yv - mu) * is, s2);
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s1 += __shfl_down(s1, off); s2 += __shfl_down(s2, off); mx = fmaxf(mx, __shfl_down(mx, off));
    }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s1; red[1][threadIdx.x >> 6] = s2; red[2][threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[rowid] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        partial[nrows + rowid] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        if (want_max) partial[2LL * nrows + rowid] = fmaxf(fmaxf(red[2][0], red[2][1]), fmaxf(red[2][2], red[2][3]));
    }
}

// The same two sums when dZ = (k == arg) ? dpooled : 0: only the arg-max element of every neighbourhood
// contributes, so the pass touches B*C*M elements (and gathers one y each) instead of B*C*M*K.
__global__ __launch_bounds__(256) void bn_bwd_pool_reduce_kernel(
    const float* __restrict__ dpooled, const int* __restrict__ arg, const float* __restrict__ Y,
    const float* __restrict__ coef, const float* __restrict__ mean, const float* __restrict__ invstd,
    float* __restrict__ partial, int relu, int C, int M, int K, int nrows, const float* __restrict__ yarg, int want_max)
{
    __shared__ float red[3][4];
    const long long rowid = blockIdx.x;
    const int ch = (int)(rowid % C);
    const float sc = coef[ch], sh = coef[C + ch], mu = mean[ch], is = invstd[ch];
    const float* y = Y + rowid * M * K;
    float s1 = 0.f, s2 = 0.f, mx = 0.f;
    for (int m = threadIdx.x; m < M; m += 256) {
        // yarg: y at the arg-max, kept by the forward pooling pass (otherwise one 128-B line fetched per 4-B value)
        const float yv = yarg ? yarg[rowid * M + m] : y[(long long)m * K + arg[rowid * M + m]];
        const float d = (!relu || __builtin_fmaf(yv, sc, sh) > 0.f) ? dpooled[rowid * M + m] : 0.f;
        mx = fmaxf(mx, fabsf(d));
        s1 += d;
        s2 = __builtin_fmaf(d, (yv - mu) * is, s2);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s1 += __shfl_down(s1, off); s2 += __shfl_down(s2, off); mx = fmaxf(mx, __shfl_down(mx, off));
    }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s1; red[1][threadIdx.x >> 6] = s2; red[2][threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[rowid] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        partial[nrows + rowid] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        if (want_max) partial[2LL * nrows + rowid] = fmaxf(fmaxf(red[2][0], red[2][1]), fmaxf(red[2][2], red[2][3]));
    }
}

// dgamma = sum2, dbeta = sum1, and the PRO_BN_BWD coefficients
//   dY = gamma*invstd * (dYhat - mean(dYhat) - yhat * mean(dYhat*yhat)) = a1*dYhat + q1*y + q0
__global__ __launch_bounds__(64) void bn_bwd_finalize_kernel(
    const float* __restrict__ partial, int nb, int C, double count, const float* __restrict__ gamma,
    const float* __restrict__ coef_fwd, const float* __restrict__ mean, const float* __restrict__ invstd,
    float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ coef4, int want_bound)
{
    const int ch = blockIdx.x * 64 + threadIdx.x;
    const long long nrows = (long long)nb * C;
    float bound = 0.f;
    if (ch < C) {
        double s1 = 0.0, s2 = 0.0;
        float mx = 0.f;
        int b = 0;
        // eight rows per pass, all loads issued before the first addition (one row per pass was a chain of nb load
        // latencies: 8-10 us at nb = 16); same order of additions
        for (; b + 7 < nb; b += 8) {
            float u[8], v[8], w[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                u[i] = partial[(long long)(b + i) * C + ch];
                v[i] = partial[nrows + (long long)(b + i) * C + ch];
                w[i] = want_bound ? partial[2 * nrows + (long long)(b + i) * C + ch] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) { s1 += (double)u[i]; s2 += (double)v[i]; mx = fmaxf(mx, w[i]); }
        }
        for (; b < nb; ++b) {
            s1 += (double)partial[(long long)b * C + ch];
            s2 += (double)partial[nrows + (long long)b * C + ch];
            if (want_bound) mx = fmaxf(mx, partial[2 * nrows + (long long)b * C + ch]);
        }
        if (dbeta) dbeta[ch] = (float)s1;
        if (dgamma) dgamma[ch] = (float)s2;
        if (coef4) {
            const float a1 = coef_fwd[ch], a0 = coef_fwd[C + ch];
            const float is = invstd[ch], mu = mean[ch];
            const float c1m = (float)(s1 / count), c2m = (float)(s2 / count);
            coef4[ch] = a1;
            coef4[C + ch] = a0;
            coef4[2 * C + ch] = -a1 * c2m * is;
            coef4[3 * C + ch] = a1 * (c2m * is * mu - c1m);
            // |dY| = |a1| |dYhat - c1m - yhat c2m| <= |a1| (max|dYhat| + |c1m| + |c2m| sqrt(n)):  |yhat| <= sqrt(n) for
            // batch statistics over n samples (n (y - mean)^2 <= n sum (y - mean)^2 = n^2 var)
            bound = fabsf(a1) * (mx + fabsf(c1m) + fabsf(c2m) * (float)sqrt(count));
        }
    }
    if (want_bound && coef4) {
        // row 4 of coef4 ([5][C]): entry i = bound of |dY| over the channels [64 i, 64 i + 64) -- what the split-fp16
        // kernels scale their streamed operand by (they take the max over the ceil(C/64) entries)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) bound = fmaxf(bound, __shfl_down(bound, off));
        if (threadIdx.x == 0) coef4[4 * C + blockIdx.x] = bound;
    }
    (void)gamma;
}

// The same finalisation for partial sums with MANY rows per channel (one row per workgroup of the fused narrow
// backward: ~1000): 16 row groups x 64 channels per workgroup, every thread sums its rows in double with four
// independent chains, the 16 group sums are combined in a fixed order.  (bn_bwd_finalize_kernel walks the rows with one
// thread per channel: 283 us for 1024 rows.)
__global__ __launch_bounds__(1024) void bn_bwd_finalize_rows_kernel(
    const float* __restrict__ partial, int rows, int C, double count, const float* __restrict__ coef_fwd,
    const float* __restrict__ mean, const float* __restrict__ invstd, float* __restrict__ dgamma,
    float* __restrict__ dbeta, float* __restrict__ coef4, const float* __restrict__ max0, int n0,
    const float* __restrict__ max1, int n1)
{
    __shared__ double red[2][16][64];
    __shared__ float rmax[2][16];
    // maxima of |dYhat| of the gradient's (up to two) parts: loaded first, reduced BEHIND the row sums (reducing them
    // here put two more load latencies in front of the row loop: 10 -> 16.7 us)
    float m0 = 0.f, m1 = 0.f;
    if (max0) {
        for (int i = threadIdx.x; i < n0; i += 1024) m0 = fmaxf(m0, max0[i]);
        for (int i = threadIdx.x; i < n1; i += 1024) m1 = fmaxf(m1, max1[i]);
    }
    const int cl = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int ch = blockIdx.x * 64 + cl;
    const long long plane = (long long)rows * C;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0, b0 = 0, b1 = 0, b2 = 0, b3 = 0;
    if (ch < C) {
        const int per = (rows + 15) / 16, r0 = grp * per, r1 = min(rows, r0 + per);
        int r = r0;
        for (; r + 7 < r1; r += 8) {                               // 16 loads in flight; the four chains add in the same order
            float u[8], v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { u[i] = partial[(long long)(r + i) * C + ch]; v[i] = partial[plane + (long long)(r + i) * C + ch]; }
            a0 += (double)u[0]; b0 += (double)v[0]; a1 += (double)u[1]; b1 += (double)v[1];
            a2 += (double)u[2]; b2 += (double)v[2]; a3 += (double)u[3]; b3 += (double)v[3];
            a0 += (double)u[4]; b0 += (double)v[4]; a1 += (double)u[5]; b1 += (double)v[5];
            a2 += (double)u[6]; b2 += (double)v[6]; a3 += (double)u[7]; b3 += (double)v[7];
        }
        for (; r + 3 < r1; r += 4) {
            a0 += (double)partial[(long long)r * C + ch];           b0 += (double)partial[plane + (long long)r * C + ch];
            a1 += (double)partial[(long long)(r + 1) * C + ch];     b1 += (double)partial[plane + (long long)(r + 1) * C + ch];
            a2 += (double)partial[(long long)(r + 2) * C + ch];     b2 += (double)partial[plane + (long long)(r + 2) * C + ch];
            a3 += (double)partial[(long long)(r + 3) * C + ch];     b3 += (double)partial[plane + (long long)(r + 3) * C + ch];
        }
        for (; r < r1; ++r) { a0 += (double)partial[(long long)r * C + ch]; b0 += (double)partial[plane + (long long)r * C + ch]; }
    }
    red[0][grp][cl] = (a0 + a1) + (a2 + a3);
    red[1][grp][cl] = (b0 + b1) + (b2 + b3);
    if (max0) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { m0 = fmaxf(m0, __shfl_xor(m0, off)); m1 = fmaxf(m1, __shfl_xor(m1, off)); }
        if ((threadIdx.x & 63) == 0) { rmax[0][threadIdx.x >> 6] = m0; rmax[1][threadIdx.x >> 6] = m1; }
    }
    __syncthreads();
    float bound = 0.f;
    if (grp == 0 && ch < C) {
        double s1 = 0, s2 = 0;
#pragma unroll
        for (int g = 0; g < 16; ++g) { s1 += red[0][g][cl]; s2 += red[1][g][cl]; }
        if (dbeta) dbeta[ch] = (float)s1;
        if (dgamma) dgamma[ch] = (float)s2;
        const float a1f = coef_fwd[ch], a0f = coef_fwd[C + ch];
        const float is = invstd[ch], mu = mean[ch];
        const float c1m = (float)(s1 / count), c2m = (float)(s2 / count);
        coef4[ch] = a1f;
        coef4[C + ch] = a0f;
        coef4[2 * C + ch] = -a1f * c2m * is;
        coef4[3 * C + ch] = a1f * (c2m * is * mu - c1m);
        if (max0) {
            float t0 = 0.f, t1 = 0.f;
#pragma unroll
            for (int w = 0; w < 16; ++w) { t0 = fmaxf(t0, rmax[0][w]); t1 = fmaxf(t1, rmax[1][w]); }
            bound = fabsf(a1f) * ((t0 + t1) + fabsf(c1m) + fabsf(c2m) * (float)sqrt(count));     // as bn_bwd_finalize_kernel
        }
    }
    if (max0 && grp == 0) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) bound = fmaxf(bound, __shfl_down(bound, off));
        if (cl == 0) coef4[4 * C + blockIdx.x] = bound;
    }
}

}  // namespace

// Deferred reductions (see wgrad_reduce_multi_kernel): recorded under a lock because autograd runs backward nodes on its
// own thread while the step object switches the mode from the caller's.
#include <mutex>
#include <vector>
namespace {
std::mutex g_defer_lock;
bool g_defer_on = false;
bool g_defer_any_stream = true;                               // usip_wgrad_defer(1): whatever stream; _on(stream): that stream only
hipStream_t g_defer_stream = nullptr;
int g_defer_hold = 0;                                         // > 0: reductions are launched at once although the mode is on
std::vector<ReduceJob> g_defer_jobs;

template <int Q>
int flush_q(std::vector<ReduceJob>& jobs, hipStream_t st)
{
    size_t at = 0;
    while (at < jobs.size()) {
        ReduceJobs J{};
        long long blocks = 0;
        while (at < jobs.size() && J.n < USIP_REDUCE_JOBS) {
            ReduceJob jb = jobs[at];
            const long long nb = (jb.elems + 63) / 64;
            if (blocks + nb > 0x7fffffffLL) break;
            jb.block0 = (int)blocks;
            blocks += nb;
            J.j[J.n++] = jb;
            ++at;
        }
        if (J.n == 0) return USIP_EINVAL;
        USIP_LAUNCH((wgrad_reduce_multi_kernel<Q>), dim3((unsigned)blocks), dim3(64 * Q), 0, st, J);
        USIP_LAUNCH_CHECK();
    }
    return USIP_OK;
}
}  // namespace

extern "C" int usip_wgrad_defer(int on)
{
    std::lock_guard<std::mutex> g(g_defer_lock);
    g_defer_on = on != 0;
    g_defer_any_stream = true;
    g_defer_stream = nullptr;
    g_defer_hold = 0;
    g_defer_jobs.clear();                                     // entering or leaving the mode: nothing stale survives
    return USIP_OK;
}

// The mode for ONE stream: only reductions whose entry point was called with `stream` are recorded; a call on
// any other stream (another thread, another device's step) launches its reduction at once, as outside the mode.
extern "C" int usip_wgrad_defer_on(void* stream)
{
    std::lock_guard<std::mutex> g(g_defer_lock);
    g_defer_on = true;
    g_defer_any_stream = false;
    g_defer_stream = (hipStream_t)stream;
    g_defer_hold = 0;
    g_defer_jobs.clear();
    return USIP_OK;
}

// hold != 0: until usip_wgrad_defer_hold(0) every reduction is launched at once although the mode is on -- for a weight
// gradient whose destination the caller hands straight to a consumer that runs BEFORE the flush (a dW that is not a view of
// the step's gradient bucket).  Returns the previous value.
extern "C" int usip_wgrad_defer_hold(int hold)
{
    std::lock_guard<std::mutex> g(g_defer_lock);
    const int was = g_defer_hold;
    g_defer_hold = hold != 0;
    return was;
}

extern "C" int usip_wgrad_flush(void* stream)
{
    std::vector<ReduceJob> many, few;
    {
        std::lock_guard<std::mutex> g(g_defer_lock);
        for (const ReduceJob& jb : g_defer_jobs) (jb.slices > 64 ? many : few).push_back(jb);
        g_defer_jobs.clear();
    }
    const int n = (int)(many.size() + few.size());
    hipStream_t st = (hipStream_t)stream;
    if (!many.empty()) { const int rc = flush_q<16>(many, st); if (rc != USIP_OK) return rc < 0 ? rc : -rc; }
    if (!few.empty()) { const int rc = flush_q<4>(few, st); if (rc != USIP_OK) return rc < 0 ? rc : -rc; }
    return n;
}

// fixed-order sum of weight-gradient partial tiles, for the other translation units of the shared MLP
int usip_mlp::launch_wgrad_reduce(const float* part, float* dW, long long elems, int slices, int N, int ldw, int coloff,
                                  hipStream_t st)
{
    {
        std::lock_guard<std::mutex> g(g_defer_lock);
        if (g_defer_on && !g_defer_hold && (g_defer_any_stream || st == g_defer_stream)) {   // the caller keeps `part` alive until usip_wgrad_flush
            g_defer_jobs.push_back(ReduceJob{part, dW, elems, slices, N, ldw, coloff, 0, 0});
            return USIP_OK;
        }
    }
    if (slices > 64)
        USIP_LAUNCH(wgrad_reduce_kernel<16>, dim3((unsigned)((elems + 63) / 64)), dim3(1024), 0, st, part, dW, elems,
                    slices, N, ldw, coloff);
    else
        USIP_LAUNCH(wgrad_reduce_kernel<4>, dim3((unsigned)((elems + 63) / 64)), dim3(256), 0, st, part, dW, elems,
                    slices, N, ldw, coloff);
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}

// ================================================================================================
extern "C" int usip_mlp_gemm_tiles(int M, int P, int nb)
{
    const int BN = (M <= 64) ? 256 : 128;
    return nb * ((P + BN - 1) / BN);
}

// mode: 0 = fp32 MFMA, 1 = bf16 multiply (perf mode), 2 = f32x3 (fp32-accurate split products on the bf16 matrix
// cores) for the launches that are matrix-bound -- the others run the fp32 kernel, whose result is the same to
// rounding and which is faster where the layer is HBM-bound (C <= 64) or too small to fill the chip.
static bool x3_gemm_pays(int M, int K, int P, int nb)
{
    const int t = usip_tuning_value(USIP_TUNE_GEMM_SPLIT3);          // 1: never, 2: whenever the shape allows
    if (t == 1) return false;
    const long long tiles = (long long)nb * ((P + 127) / 128) * ((M + 127) / 128);
    if ((long long)K * P >= (1LL << 30)) return false;        // the split kernels address a cloud with 32-bit byte offsets
    if (t == 2) return M > 64 && K >= 16;
    // 128 tiles of 128 x 128 (r03; 512 in r02): with 128-row tiles for small launches (usip_mlp_x3p_tile_rows) the
    // second-stage and head layers -- 16 x 512 positions -- run 1.3-2.1x faster on the split kernel than on fp32 MFMAs
    // (73 -> 36 us for 640 -> 512); in the step 5.195 -> 5.145 ms, same box, two runs each
    return M >= 128 && K >= 128 && tiles >= (t >= 16 ? t : 128);     // t >= 16: measurement, the tile threshold itself
}

static int mlp_gemm_impl(int mode, const float* At, int lda, const float* X, const float* X2,
                         const float* coef, int pro, const float* bias, const float* rowbias,
                         int rb_group, const float* pool_dp, const int32_t* pool_arg, int pool_group,
                         float* Y, int y_rows, float* stats, int M, int K, int P, int nb, void* stream)
{
    // lda < 0 selects the transposed storage of the matrix operand: At is [M][K] with row stride -lda
    // (the forward product can then read W itself instead of a transposed copy made every step)
    const int a_trans = lda < 0 ? 1 : 0;
    if (a_trans) lda = -lda;
    if (M < 1 || K < 1 || P < 0 || nb < 0 || lda < (a_trans ? K : M)) return USIP_EINVAL;
    if ((long long)P * nb == 0) return USIP_OK;
    if (!At || !Y || pro < 0 || pro > 3) return USIP_EINVAL;
    if (pro != PRO_BN_BWD_POOL && !X) return USIP_EINVAL;
    if (pro != PRO_NONE && !coef) return USIP_EINVAL;
    if ((pro == PRO_BN_BWD || pro == PRO_BN_BWD_POOL) && (!X2 || stats)) return USIP_EINVAL;
    if (pro == PRO_BN_BWD_POOL && (!pool_dp || !pool_arg || pool_group < 4 || pool_group % 4 != 0 ||
                                   P % pool_group != 0 || P % 4 != 0)) return USIP_EINVAL;
    if (rowbias && (rb_group < 1 || P % rb_group != 0)) return USIP_EINVAL;
    if (y_rows == 0) y_rows = M;                           // Y is a dense [nb][M][P] tensor
    if (y_rows < M) return USIP_EINVAL;
    GemmArgs a{At, lda, X, X2, coef, bias, Y, stats, M, K, P, nb, rowbias, rb_group, pool_dp, pool_arg, pool_group,
               a_trans, y_rows, (P % 4 == 0 && (reinterpret_cast<uintptr_t>(Y) & 15u) == 0) ? 1 : 0};
    hipStream_t st = (hipStream_t)stream;
    if (mode == 1) return launch_gemm_bf16(a, pro, st);
    if (mode == 2 && x3_gemm_pays(M, K, P, nb)) return launch_gemm_x3(a, pro, st);
    // K-step 16: 32 was measured slower (LDS per workgroup doubles, occupancy halves)
    // fewer than two workgroups per CU: halve the rows per wave (32 x 256 resp. 64 x 128 workgroups)
    if (M <= 64)
        return ((long long)nb * ((P + 255) / 256) < 512 && M <= 32) ? launch_gemm<1, 4, 16, 1>(a, pro, st)
                                                                    : launch_gemm<1, 4, 16, 2>(a, pro, st);
    if ((long long)nb * ((P + 127) / 128) * ((M + 127) / 128) < 512) return launch_gemm<2, 2, 16, 1>(a, pro, st);
    return launch_gemm<2, 2, 16, 2>(a, pro, st);
}

#define USIP_GEMM_PARAMS                                                                                     \
    const float *At, int lda, const float *X, const float *X2, const float *coef, int pro, const float *bias, \
        const float *rowbias, int rb_group, const float *pool_dp, const int32_t *pool_arg, int pool_group,    \
        float *Y, int y_rows, float *stats, int M, int K, int P, int nb,                                      \
        void *stream
#define USIP_GEMM_ARGS \
    At, lda, X, X2, coef, pro, bias, rowbias, rb_group, pool_dp, pool_arg, pool_group, Y, y_rows, stats, \
        M, K, P, nb, stream
extern "C" int usip_mlp_gemm_f32(USIP_GEMM_PARAMS) { return mlp_gemm_impl(0, USIP_GEMM_ARGS); }
extern "C" int usip_mlp_gemm_bf16(USIP_GEMM_PARAMS) { return mlp_gemm_impl(1, USIP_GEMM_ARGS); }
extern "C" int usip_mlp_gemm_f32x3(USIP_GEMM_PARAMS) { return mlp_gemm_impl(2, USIP_GEMM_ARGS); }
// 1 when usip_mlp_gemm_f32x3 runs this shape on the split-product kernel, 0 when it hands it to the fp32 kernel
extern "C" int usip_mlp_gemm_f32x3_used(int M, int K, int P, int nb) { return x3_gemm_pays(M, K, P, nb) ? 1 : 0; }
#undef USIP_GEMM_PARAMS
#undef USIP_GEMM_ARGS

// Workspace (floats) the weight-gradient needs, and the slicing it will use.
static void wgrad_plan(int M, int N, int P, int nb, int* seglen, int* segs, int* small, int* tiles)
{
    *small = (M <= 64 && N <= 64) ? 1 : 0;
    const int BM = *small ? 64 : 128, BN = *small ? 64 : 128;
    *tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    // aim at ~1024 workgroups (4 per CU); a position segment is a multiple of 32 and >= 512 so that
    // the partial tile a workgroup writes stays small next to what it streams
    long long want = 1024 / (*tiles);
    if (want < 1) want = 1;
    long long per_cloud = (want + nb - 1) / nb;
    if (per_cloud < 1) per_cloud = 1;
    long long sl = (P + per_cloud - 1) / per_cloud;
    sl = ((sl + 31) / 32) * 32;
    if (sl < 512) {
        // Long segments keep the partial tiles small next to what a workgroup streams -- but only while they
        // still fill the chip: the M-sized products of the pooled-concat layers and the head (P = 512 per
        // cloud) ran as 16-128 workgroups of 16 serial stages each (45-70 us for < 60 MB).  Take the longest
        // segment in {512, 256, 128, 64} that still gives >= 256 workgroups.
        long long floor_sl = 512;
        while (floor_sl > 64 && (long long)(*tiles) * nb * ((P + floor_sl - 1) / floor_sl) < 256) floor_sl /= 2;
        if (sl < floor_sl) sl = floor_sl;
    }
    *seglen = (int)sl;
    *segs = (int)((P + sl - 1) / sl);
}

extern "C" long long usip_mlp_wgrad_workspace(int M, int N, int P, int nb)
{
    int seglen, segs, small, tiles, sl3, sg3, t3;
    wgrad_plan(M, N, P, nb, &seglen, &segs, &small, &tiles);
    wgrad_x3_plan(M, N, P, nb, &sl3, &sg3, &t3);              // the f32x3 kernel slices differently: cover both
    return (long long)nb * (segs > sg3 ? segs : sg3) * M * N;
}

// Number of workgroups usip_mlp_wgrad_f32 launches (lets a profiler match launches to layer shapes).
extern "C" int usip_mlp_wgrad_blocks(int M, int N, int P, int nb)
{
    int seglen, segs, small, tiles;
    wgrad_plan(M, N, P, nb, &seglen, &segs, &small, &tiles);
    return tiles * nb * segs;
}

static bool x3_wgrad_pays(int M, int N, int P, int nb)
{
    const int t = usip_tuning_value(USIP_TUNE_GEMM_SPLIT3);
    if (t == 1) return false;
    if (t == 2) return M > 64 || N > 64;
    // few positions (the head: 16 x 512): with position segments down to 128 (wgrad_x3_plan) 512 x 640 runs 102 -> 42 us,
    // 512 x 512 66 -> 40, 256 x 512 39 -> 29; 256 x 256 is a tie (27 us) and stays on the fp32 kernel
    const bool few_ok = (long long)M * N >= 256LL * 512 && usip_tuning_value(USIP_TUNE_X3_WGRAD_TILE) != 3;
    return M >= 128 && N >= 128 && ((long long)nb * P >= 32768 || few_ok);
}
extern "C" int usip_mlp_wgrad_f32x3_used(int M, int N, int P, int nb) { return x3_wgrad_pays(M, N, P, nb) ? 1 : 0; }

// Workgroups usip_mlp_wgrad_f32x3 launches, negative when it runs the 256 x 256-tile kernel (profiling aid).
extern "C" int usip_mlp_wgrad_f32x3_blocks(int M, int N, int P, int nb)
{
    int seglen, segs, small, tiles;
    wgrad_plan(M, N, P, nb, &seglen, &segs, &small, &tiles);
    const bool x3 = !small && x3_wgrad_pays(M, N, P, nb);
    if (x3 && P % 4 == 0 && M > 128 && N > 128 && usip_tuning_value(USIP_TUNE_X3_WGRAD_TILE) != 1) {
        wgrad_x3_plan(M, N, P, nb, &seglen, &segs, &tiles);
        return -(tiles * nb * segs);
    }
    return tiles * nb * segs;
}

static int mlp_wgrad_impl(int mode, const float* G, const float* G2, const float* coef, int pro,
                          const float* X, const float* xcoef, const float* pool_dp, const int32_t* pool_arg,
                          int pool_group, float* workspace, float* dW, int ldw,
                          int coloff, int M, int N, int P, int nb, void* stream)
{
    if (M < 1 || N < 1 || P < 1 || nb < 1 || ldw < N + coloff || coloff < 0) return USIP_EINVAL;
    if (!dW) return USIP_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (!X || !workspace || (pro != PRO_NONE && pro != PRO_BN_BWD && pro != PRO_BN_BWD_POOL)) return USIP_EINVAL;
    if (pro != PRO_BN_BWD_POOL && !G) return USIP_EINVAL;
    if ((pro == PRO_BN_BWD || pro == PRO_BN_BWD_POOL) && (!G2 || !coef)) return USIP_EINVAL;
    if (pro == PRO_BN_BWD_POOL && (!pool_dp || !pool_arg || pool_group < 4 || pool_group % 4 != 0 ||
                                   P % pool_group != 0 || P % 4 != 0 || (reinterpret_cast<uintptr_t>(G2) & 15u)))
        return USIP_EINVAL;
    int seglen, segs, small, tiles;
    wgrad_plan(M, N, P, nb, &seglen, &segs, &small, &tiles);
    const bool bf16 = (mode == 1), x3 = (mode == 2 || mode == 3) && !small && x3_wgrad_pays(M, N, P, nb);
    // 256 x 256 tiles for the wide layers (every streamed element is prepared for twice as many partners)
    const bool vec0 = (P % 4 == 0) && (pro == PRO_BN_BWD_POOL || (reinterpret_cast<uintptr_t>(G) & 15u) == 0) &&
                      ((reinterpret_cast<uintptr_t>(X) & 15u) == 0) &&
                      (pro == PRO_NONE || (reinterpret_cast<uintptr_t>(G2) & 15u) == 0);
    const bool x3big = x3 && vec0 && M > 128 && N > 128 && usip_tuning_value(USIP_TUNE_X3_WGRAD_TILE) != 1;
    // mode 3: two fp16 planes per operand where both operands have a bound (see usip_mlp_wgrad_x2h_f32); f32x3 otherwise
    const bool x2h = (mode == 3) && x3big && (pro == PRO_BN_BWD || pro == PRO_BN_BWD_POOL) && xcoef != nullptr;
    if (x3big) wgrad_x3_plan(M, N, P, nb, &seglen, &segs, &tiles);
    WgradArgs a{G, G2, coef, X, xcoef, pool_dp, pool_arg, pool_group, workspace, M, N, P, nb, seglen, segs};
    const bool xpro = xcoef != nullptr;
    const bool vec = (P % 4 == 0) && (pro == PRO_BN_BWD_POOL || (reinterpret_cast<uintptr_t>(G) & 15u) == 0) &&
                     ((reinterpret_cast<uintptr_t>(X) & 15u) == 0) &&
                     (pro == PRO_NONE || (reinterpret_cast<uintptr_t>(G2) & 15u) == 0);
    if (pro == PRO_BN_BWD_POOL && !vec) return USIP_EINVAL;
    const long long blocks = (long long)tiles * nb * segs;
    if (blocks > 0x7fffffffLL) return USIP_EINVAL;
    dim3 grid((unsigned)blocks), block(256);
    if (bf16) {
        const int rc = launch_wgrad_bf16(a, pro, xpro, vec, small, (unsigned)blocks, st);
        if (rc != USIP_OK) return rc;
    }
    if (x3) {
        const int rc = x2h ? launch_wgrad_x2h_256(a, pro, (unsigned)blocks, st)
                     : x3big ? launch_wgrad_x3_256(a, pro, xpro, (unsigned)blocks, st)
                             : launch_wgrad_x3(a, pro, xpro, vec, (unsigned)blocks, st);
        if (rc != USIP_OK) return rc;
    }
#define USIP_WGRAD_CASE(T_, P_, X_, V_)                                                        \
    if (!bf16 && !x3 && small == (T_ == 1) && pro == P_ && xpro == X_ && vec == V_) {         \
        USIP_LAUNCH((wgrad_kernel<T_, T_, P_, X_, V_>), grid, block, 0, st, a);                \
        USIP_LAUNCH_CHECK();                                                                   \
    }
#define USIP_WGRAD_CASES(T_, P_) \
    USIP_WGRAD_CASE(T_, P_, false, true) USIP_WGRAD_CASE(T_, P_, false, false) \
    USIP_WGRAD_CASE(T_, P_, true, true) USIP_WGRAD_CASE(T_, P_, true, false)
    USIP_WGRAD_CASES(1, PRO_NONE)
    USIP_WGRAD_CASES(1, PRO_BN_BWD)
    USIP_WGRAD_CASES(2, PRO_NONE)
    USIP_WGRAD_CASES(2, PRO_BN_BWD)
    USIP_WGRAD_CASE(1, PRO_BN_BWD_POOL, false, true) USIP_WGRAD_CASE(1, PRO_BN_BWD_POOL, true, true)
    USIP_WGRAD_CASE(2, PRO_BN_BWD_POOL, false, true) USIP_WGRAD_CASE(2, PRO_BN_BWD_POOL, true, true)
#undef USIP_WGRAD_CASES
#undef USIP_WGRAD_CASE
    const long long elems = (long long)M * N;
    return usip_mlp::launch_wgrad_reduce(workspace, dW, elems, nb * segs, N, ldw, coloff, st);
}

#define USIP_WGRAD_PARAMS                                                                                      \
    const float *G, const float *G2, const float *coef, int pro, const float *X, const float *xcoef,            \
        const float *pool_dp, const int32_t *pool_arg, int pool_group, float *workspace, float *dW, int ldw,    \
        int coloff, int M, int N, int P, int nb, void *stream
#define USIP_WGRAD_ARGS \
    G, G2, coef, pro, X, xcoef, pool_dp, pool_arg, pool_group, workspace, dW, ldw, coloff, M, N, P, nb, stream
extern "C" int usip_mlp_wgrad_f32(USIP_WGRAD_PARAMS) { return mlp_wgrad_impl(0, USIP_WGRAD_ARGS); }
extern "C" int usip_mlp_wgrad_bf16(USIP_WGRAD_PARAMS) { return mlp_wgrad_impl(1, USIP_WGRAD_ARGS); }
extern "C" int usip_mlp_wgrad_f32x3(USIP_WGRAD_PARAMS) { return mlp_wgrad_impl(2, USIP_WGRAD_ARGS); }
// f32x2: as usip_mlp_wgrad_f32x3, and launches with pro 2 / 3, M, N > 128, coef = the [5][M] array of
// usip_bn_backward_reduce_f32(want_bound) and xcoef = the [4][N] (scale, shift, mean, invstd) of a training-mode BatchNorm
// over the nb * P samples of this launch use two fp16 planes and three plane products (usip_mlp_gemm_x2h_f32's scheme).
extern "C" int usip_mlp_wgrad_x2h_f32(USIP_WGRAD_PARAMS) { return mlp_wgrad_impl(3, USIP_WGRAD_ARGS); }
#undef USIP_WGRAD_PARAMS
#undef USIP_WGRAD_ARGS

extern "C" int usip_bn_finalize_f32(const float* stats, int ntiles, int C, long long count,
                                    const float* gamma, const float* beta, float eps, float momentum,
                                    float* running_mean, float* running_var, float* mean, float* invstd,
                                    float* coef, void* stream)
{
    if (!stats || ntiles < 1 || C < 1 || count < 1 || !mean || !invstd || !coef) return USIP_EINVAL;
    if ((running_mean == nullptr) != (running_var == nullptr)) return USIP_EINVAL;
    USIP_LAUNCH(bn_finalize_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, stats, ntiles, C, (double)count,
                gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, coef);
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}

extern "C" int usip_bn_apply_f32(const float* Y, const float* coef, float* Z, int relu,
                                 int nb, int C, int P, void* stream)
{
    if (nb < 0 || C < 1 || P < 0) return USIP_EINVAL;
    if ((long long)nb * P == 0) return USIP_OK;
    if (!Y || !coef || !Z || (long long)nb * C > 65535LL * 65535LL) return USIP_EINVAL;
    const bool vec = (P % 4 == 0) && ((reinterpret_cast<uintptr_t>(Y) & 15u) == 0) &&
                     ((reinterpret_cast<uintptr_t>(Z) & 15u) == 0);
    hipStream_t st = (hipStream_t)stream;
    // grid.y <= 65535: fold rows if needed
    const long long rows = (long long)nb * C;
    if (rows > 65535) {
        // process in slabs of <= 65535 rows (C divides the slab start only if slab % C == 0)
        const long long slab = (65535 / C) * (long long)C;
        if (slab == 0) return USIP_EINVAL;
        for (long long r0 = 0; r0 < rows; r0 += slab) {
            const long long n = (rows - r0 < slab) ? rows - r0 : slab;
            if (vec) USIP_LAUNCH((bn_apply_kernel<true>), dim3(usip_ceil_div(P, 1024), (unsigned)n), dim3(256), 0, st,
                                 Y + r0 * P, coef, Z + r0 * P, relu, C, P);
            else USIP_LAUNCH((bn_apply_kernel<false>), dim3(usip_ceil_div(P, 256), (unsigned)n), dim3(256), 0, st,
                             Y + r0 * P, coef, Z + r0 * P, relu, C, P);
            USIP_LAUNCH_CHECK();
        }
        return USIP_OK;
    }
    if (vec) USIP_LAUNCH((bn_apply_kernel<true>), dim3(usip_ceil_div(P, 1024), (unsigned)rows), dim3(256), 0, st,
                         Y, coef, Z, relu, C, P);
    else USIP_LAUNCH((bn_apply_kernel<false>), dim3(usip_ceil_div(P, 256), (unsigned)rows), dim3(256), 0, st,
                     Y, coef, Z, relu, C, P);
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}

extern "C" int usip_bn_backward_reduce_f32(const float* dZ, const float* Y, const float* coef_fwd,
                                           const float* mean, const float* invstd, const float* gamma,
                                           int relu, float* partial, float* dgamma, float* dbeta,
                                           float* coef4, float* gsum, int group,
                                           int nb, int C, int P, int want_bound, void* stream)
{
    if (nb < 1 || C < 1 || P < 1 || !dZ || !partial) return USIP_EINVAL;
    const int plain = (Y == nullptr);
    if (gsum) {
        // vector path only: K % 4 == 0, K/4 a power of two <= 64, 16-B aligned rows
        const int lpg = group / 4;
        if (plain || group < 4 || group % 4 != 0 || (lpg & (lpg - 1)) != 0 || lpg > 64 || P % group != 0 ||
            (reinterpret_cast<uintptr_t>(dZ) & 15u) || (reinterpret_cast<uintptr_t>(Y) & 15u))
            return USIP_EINVAL;
    }
    if (!plain && (!coef_fwd || !mean || !invstd)) return USIP_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const long long rows = (long long)nb * C;
    if (rows > 0x7fffffffLL) return USIP_EINVAL;
    // loads of UNR loop iterations in flight per lane (the sums are the same bits for every UNR).  Round 5 measured 2 and 4
    // against 1 inside the step, same box, alternating (profiles/r05b_forms_ab.txt): 46.1-48.0 / 46.1 / 45.3 us per launch --
    // this pass is not short of bytes in flight (32 waves per CU already hold 64 KiB), so 1 stays; knob r5_forms bit 1 -> 2,
    // bit 2 -> 4
    const int forms = usip_tuning_value(USIP_TUNE_R5_FORMS);
    const int unr = (forms & 2) ? 2 : (forms & 4) ? 4 : 1;
#define USIP_BNRED(G_, U_)                                                                                        \
    USIP_LAUNCH((bn_bwd_reduce_kernel<G_, U_>), dim3((unsigned)rows), dim3(256), 0, st, dZ, Y, coef_fwd, mean, invstd, \
                partial, gsum, relu, plain, C, P, (int)rows, group, want_bound)
    if (gsum) { if (unr == 1) USIP_BNRED(true, 1); else if (unr == 4) USIP_BNRED(true, 4); else USIP_BNRED(true, 2); }
    else { if (unr == 1) USIP_BNRED(false, 1); else if (unr == 4) USIP_BNRED(false, 4); else USIP_BNRED(false, 2); }
#undef USIP_BNRED
    USIP_LAUNCH_CHECK();
    USIP_LAUNCH(bn_bwd_finalize_kernel, dim3(usip_ceil_div(C, 64)), dim3(64), 0, st, partial, nb, C,
                (double)nb * (double)P, gamma, coef_fwd, mean, invstd, dgamma, dbeta, plain ? nullptr : coef4,
                want_bound);
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}

extern "C" int usip_bn_pool_backward_reduce_f32(const float* dpooled, const int32_t* arg, const float* Y,
                                                const float* yarg, const float* coef_fwd, const float* mean,
                                                const float* invstd, const float* gamma, int relu, float* partial,
                                                float* dgamma, float* dbeta, float* coef4, int nb, int C, int M,
                                                int K, int want_bound, void* stream)
{
    if (nb < 1 || C < 1 || M < 1 || K < 1 || !dpooled || !arg || !Y || !coef_fwd || !mean || !invstd || !partial)
        return USIP_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const long long rows = (long long)nb * C;
    if (rows > 0x7fffffffLL) return USIP_EINVAL;
    USIP_LAUNCH(bn_bwd_pool_reduce_kernel, dim3((unsigned)rows), dim3(256), 0, st, dpooled, arg, Y, coef_fwd, mean,
                invstd, partial, relu, C, M, K, (int)rows, yarg, want_bound);
    USIP_LAUNCH_CHECK();
    if (!dgamma && !dbeta && !coef4) return USIP_OK;          // partial sums only (combined by usip_bn_backward_finalize_f32)
    USIP_LAUNCH(bn_bwd_finalize_kernel, dim3(usip_ceil_div(C, 64)), dim3(64), 0, st, partial, nb, C,
                (double)nb * (double)M * (double)K, gamma, coef_fwd, mean, invstd, dgamma, dbeta, coef4, want_bound);
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}

// dgamma, dbeta and the BatchNorm-backward coefficients from partial sums produced elsewhere (the fused narrow
// backward, usip_bn_pool_backward_reduce_f32 with NULL outputs): partial = [2][rows][C] (s1 rows, then s2 rows), summed
// over `rows` in order, in double.  count = elements per channel the statistics were taken over.
extern "C" int usip_bn_backward_finalize_f32(const float* partial, int rows, int C, long long count,
                                             const float* coef_fwd, const float* mean, const float* invstd,
                                             float* dgamma, float* dbeta, float* coef4, void* stream)
{
    if (!partial || rows < 1 || C < 1 || count < 1 || !coef_fwd || !mean || !invstd || !coef4) return USIP_EINVAL;
    USIP_LAUNCH(bn_bwd_finalize_rows_kernel, dim3(usip_ceil_div(C, 64)), dim3(1024), 0, (hipStream_t)stream, partial,
                rows, C, (double)count, coef_fwd, mean, invstd, dgamma, dbeta, coef4, (const float*)nullptr, 0,
                (const float*)nullptr, 0);
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}

// The same, also writing row 4 of coef4 ([5][C]): the gradient whose sums these are is the sum of up to two parts with
// maxima max0[n0], max1[n1] (max1 may be NULL) of |dYhat|; |dYhat| <= max(max0) + max(max1).
extern "C" int usip_bn_backward_finalize_max_f32(const float* partial, int rows, int C, long long count,
                                                 const float* coef_fwd, const float* mean, const float* invstd,
                                                 float* dgamma, float* dbeta, float* coef4, const float* max0, int n0,
                                                 const float* max1, int n1, void* stream)
{
    if (!partial || rows < 1 || C < 1 || count < 1 || !coef_fwd || !mean || !invstd || !coef4 || !max0 || n0 < 1 ||
        n1 < 0 || (n1 > 0 && !max1))
        return USIP_EINVAL;
    USIP_LAUNCH(bn_bwd_finalize_rows_kernel, dim3(usip_ceil_div(C, 64)), dim3(1024), 0, (hipStream_t)stream, partial,
                rows, C, (double)count, coef_fwd, mean, invstd, dgamma, dbeta, coef4, max0, n0, max1, max1 ? n1 : 0);
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}
