// usip_amd/csrc/narrow_bwd.hip -- backward of a NARROW shared-MLP layer (64 inputs, 64 or 128 outputs: conv2, conv3 and
// the feature half of conv4 of RPN_Detector_Ball, models/networks.py:705-709) as ONE kernel.
//
// These layers are HBM-bound: 2*Cin*Cout flops per position against 4*(2*Cout + 2*Cin) bytes -- 16 flop/B at 64 x 64.
// The generic path reads the layer's (dZ, Y) pair three times (BatchNorm-backward reduction, data-gradient GEMM,
// weight-gradient GEMM).  Here a workgroup stages one position tile of (dZ, Y, X) ONCE, forms dY = BN'(ReLU'(dZ)) in
// LDS, and produces from it both
//     dX[ci][p]  = sum_co W[co][ci] * dY[co][p]          (written tile by tile)
//     dW[co][ci] += sum_p dY[co][p] * act(X)[ci][p]      (accumulated in registers over the workgroup's segment)
// with v_mfma_f32_32x32x2_f32 (exact fp32: same arithmetic class as the generic kernels).  One LDS layout serves
// both products without bank conflicts: dY and act(X) are stored TRANSPOSED, [position][channel] with a row of
// Cout+1 / Cin+1 floats -- the weight-gradient reads walk the channels (consecutive banks), the data-gradient reads
// walk the positions (stride Cout+1, odd).
#include "mlp_common.h"

using namespace usip_mlp;

namespace {

struct NarrowArgs {
    const float* dZ; const float* Y; const float* coef4;   // [nb][COUT][P] x2, [4][COUT]
    const float* X; const float* xcoef;                    // [nb][x_rows][P] (rows [0, CIN) used), [2][CIN] or null
    const float* W; int ldw;                               // W[co * ldw + ci]
    float* dX; int dx_rows;                                // [nb][dx_rows][P], rows [0, CIN) written
    float* part;                                           // [nb * segs][COUT][CIN]
    int x_rows, P, nb, seglen, segs;
    float* red;                                            // RED: [2][nb * segs][CIN] BatchNorm-backward sums of the INPUT layer, then [nb * segs] maxima
};

// RED (needs XPRO; xcoef is then the producing layer's [4][CIN] forward coefficients: scale, shift, mean, invstd): the
// dX this kernel writes is the incoming gradient dZ' of the layer that produced X.  That layer's BatchNorm backward
// starts with s1 = sum dZ' * [relu on], s2 = sum dZ' * [relu on] * xhat over all positions -- another full pass over
// (dZ', X) in the generic path.  Here the tile of dX is in registers and the tile of X in LDS, so the sums are
// accumulated on the way out (one partial pair per workgroup, combined in fixed order by usip_bn_backward_finalize_f32).
template <int COUT, bool XPRO, bool RED>
__global__ __launch_bounds__(256) void narrow_bwd_kernel(const NarrowArgs a)
{
    constexpr int CIN = 64;
    constexpr int BP = (COUT == 64) ? 64 : 32;                // positions per tile
    constexpr int LG = COUT + 1, LX = CIN + 1;                // LDS row lengths (odd: conflict-free both ways)
    constexpr int Q = BP / 4;                                 // float4 per row of a tile
    constexpr int NG = COUT * Q / 256, NX = CIN * Q / 256;    // float4 per thread per tile: (4, 4) or (4, 2)
    __shared__ __attribute__((aligned(16))) float Gt[BP][LG];  // dY^T
    __shared__ __attribute__((aligned(16))) float Xt[BP][LX];  // act(X)^T (RED: X^T raw, activated when read)
    __shared__ __attribute__((aligned(16))) float Ws[COUT][CIN];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x / a.segs, seg = blockIdx.x % a.segs;
    const int pbeg = seg * a.seglen, pend = min(a.P, pbeg + a.seglen);
    const int ntile = (pend - pbeg) / BP;

    for (int i = tid; i < COUT * CIN; i += 256) Ws[i / CIN][i % CIN] = a.W[(long long)(i / CIN) * a.ldw + i % CIN];

    // thread -> NG (row, 4 positions) pieces of dZ / Y and NX of X: f = tid + i*256, row = f / Q, kq = (f % Q) * 4
    const int kq = (tid % Q) * 4;
    const float* gz[NG];
    const float* gy[NG];
    float gc[NG][4];
    int grow[NG];
#pragma unroll
    for (int i = 0; i < NG; ++i) {
        grow[i] = (tid + i * 256) / Q;
        gz[i] = a.dZ + ((long long)b * COUT + grow[i]) * a.P + pbeg + kq;
        gy[i] = a.Y + ((long long)b * COUT + grow[i]) * a.P + pbeg + kq;
#pragma unroll
        for (int j = 0; j < 4; ++j) gc[i][j] = a.coef4[j * COUT + grow[i]];
    }
    const float* gx[NX];
    float xc[NX][2];
    int xrow[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        xrow[i] = (tid + i * 256) / Q;
        gx[i] = a.X + ((long long)b * a.x_rows + xrow[i]) * a.P + pbeg + kq;
        xc[i][0] = XPRO ? a.xcoef[xrow[i]] : 1.f;
        xc[i][1] = XPRO ? a.xcoef[CIN + xrow[i]] : 0.f;
    }

    // MFMA roles.  COUT = 64 (BP = 64): every wave owns one 32 x 32 tile of dX^T (positions x inputs) and one of dW.
    // COUT = 128 (BP = 32): waves 0, 1 own the two dX^T tiles (64 k-steps each), waves 2, 3 four dW tiles each.
    constexpr int NDW = (COUT == 64) ? 1 : 4;
    const bool does_dx = (COUT == 64) || wave < 2;
    const bool does_dw = (COUT == 64) || wave >= 2;
    const int dx_pt = (COUT == 64) ? (wave >> 1) : 0, dx_ct = (COUT == 64) ? (wave & 1) : wave;   // position / input tile
    const int dw_ct = (COUT == 64) ? (wave & 1) : (wave & 1);                                      // input tile of dW
    const int dw_ot0 = (COUT == 64) ? (wave >> 1) : 0;                                            // first output tile
    f32x16 acc_dw[NDW];
#pragma unroll
    for (int t = 0; t < NDW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_dw[t][r] = 0.f;

    float4 rz[NG], ry[NG], rx[NX];
    auto load_tile = [&](int t) {
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            rz[i] = *reinterpret_cast<const float4*>(gz[i] + (long long)t * BP);
            ry[i] = *reinterpret_cast<const float4*>(gy[i] + (long long)t * BP);
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) rx[i] = *reinterpret_cast<const float4*>(gx[i] + (long long)t * BP);
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const float c0 = gc[i][0], c1 = gc[i][1], c2 = gc[i][2], c3 = gc[i][3];
            Gt[kq + 0][grow[i]] = pro_apply<PRO_BN_BWD>(rz[i].x, ry[i].x, c0, c1, c2, c3);
            Gt[kq + 1][grow[i]] = pro_apply<PRO_BN_BWD>(rz[i].y, ry[i].y, c0, c1, c2, c3);
            Gt[kq + 2][grow[i]] = pro_apply<PRO_BN_BWD>(rz[i].z, ry[i].z, c0, c1, c2, c3);
            Gt[kq + 3][grow[i]] = pro_apply<PRO_BN_BWD>(rz[i].w, ry[i].w, c0, c1, c2, c3);
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            float4 v = rx[i];
            if (XPRO && !RED) {
                v.x = fmaxf(__builtin_fmaf(v.x, xc[i][0], xc[i][1]), 0.f); v.y = fmaxf(__builtin_fmaf(v.y, xc[i][0], xc[i][1]), 0.f);
                v.z = fmaxf(__builtin_fmaf(v.z, xc[i][0], xc[i][1]), 0.f); v.w = fmaxf(__builtin_fmaf(v.w, xc[i][0], xc[i][1]), 0.f);
            }
            Xt[kq + 0][xrow[i]] = v.x; Xt[kq + 1][xrow[i]] = v.y; Xt[kq + 2][xrow[i]] = v.z; Xt[kq + 3][xrow[i]] = v.w;
        }
    };

    const int c = lane & 31, kr = lane >> 5;
    float* dXb = a.dX + (long long)b * a.dx_rows * a.P;
    // RED: per-lane coefficients of the lane's input channel in the two roles
    float wsc = 1.f, wsh = 0.f, rsc = 1.f, rsh = 0.f, rmu = 0.f, ris = 0.f, s1 = 0.f, s2 = 0.f, mx = 0.f;
    if (RED) {
        wsc = a.xcoef[dw_ct * 32 + c]; wsh = a.xcoef[CIN + dw_ct * 32 + c];
        const int ci = (dx_ct & 1) * 32 + c;                  // waves without a dX tile (COUT = 128: 2, 3) stay in range
        rsc = a.xcoef[ci]; rsh = a.xcoef[CIN + ci]; rmu = a.xcoef[2 * CIN + ci]; ris = a.xcoef[3 * CIN + ci];
    }
    if (ntile > 0) load_tile(0);
    for (int t = 0; t < ntile; ++t) {
        __syncthreads();                                      // everyone is done with the previous tile's LDS (and Ws is loaded)
        store_tile();
        if (t + 1 < ntile) load_tile(t + 1);                  // in flight under the MFMAs below
        __syncthreads();
        if (does_dx) {
            // dX^T[pos][ci] = sum_co dY^T[pos][co] * W[co][ci]
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll 8
            for (int k = 0; k < COUT; k += 2)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Gt[dx_pt * 32 + c][k + kr], Ws[k + kr][dx_ct * 32 + c], acc, 0, 0, 0);
            // lane holds input channel dx_ct*32 + c, four runs of four consecutive positions
            float* orow = dXb + (long long)(dx_ct * 32 + c) * a.P + pbeg + (long long)t * BP + dx_pt * 32 + 4 * kr;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(orow + 8 * g) = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
            if (RED) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float xv = Xt[dx_pt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kr][dx_ct * 32 + c];
                    const float d = (__builtin_fmaf(xv, rsc, rsh) > 0.f) ? acc[r] : 0.f;
                    s1 += d;
                    s2 = __builtin_fmaf(d, (xv - rmu) * ris, s2);
                    mx = fmaxf(mx, fabsf(d));                 // bound of the producing layer's dY for its f32x2 backward
                }
            }
        }
        if (does_dw) {
            // dW[co][ci] += sum_p dY^T[p][co] * act(X)^T[p][ci]
#pragma unroll 4
            for (int k = 0; k < BP; k += 2) {
                float xb = Xt[k + kr][dw_ct * 32 + c];
                if (RED) xb = fmaxf(__builtin_fmaf(xb, wsc, wsh), 0.f);
#pragma unroll
                for (int u = 0; u < NDW; ++u) {
                    const int ot = (COUT == 64) ? dw_ot0 : u;
                    acc_dw[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(Gt[k + kr][ot * 32 + c], xb, acc_dw[u], 0, 0, 0);
                }
            }
        }
    }
    if (RED) {
        // the two half-waves hold the same channel at different positions; then the waves that share an input tile
        __syncthreads();
        float* rs = &Gt[0][0];                                // [4 waves][2][32]
        s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
        if (does_dx && kr == 0) { rs[(wave * 2 + 0) * 32 + c] = s1; rs[(wave * 2 + 1) * 32 + c] = s2; }
        __syncthreads();
        if (tid < CIN) {
            const int ct = tid >> 5, cc = tid & 31;
            float t1, t2;
            if (COUT == 64) {                                 // waves ct and ct + 2 own input tile ct
                t1 = rs[(ct * 2 + 0) * 32 + cc] + rs[((ct + 2) * 2 + 0) * 32 + cc];
                t2 = rs[(ct * 2 + 1) * 32 + cc] + rs[((ct + 2) * 2 + 1) * 32 + cc];
            } else {                                          // wave ct alone
                t1 = rs[(ct * 2 + 0) * 32 + cc];
                t2 = rs[(ct * 2 + 1) * 32 + cc];
            }
            const long long nblk = (long long)a.nb * a.segs;
            a.red[(long long)blockIdx.x * CIN + tid] = t1;
            a.red[(nblk + blockIdx.x) * CIN + tid] = t2;
        }
        __syncthreads();
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        if (lane == 0) rs[wave] = mx;
        __syncthreads();
        if (tid == 0) a.red[2LL * a.nb * a.segs * CIN + blockIdx.x] = fmaxf(fmaxf(rs[0], rs[1]), fmaxf(rs[2], rs[3]));
    }
    if (does_dw) {
        float* out = a.part + (long long)blockIdx.x * COUT * CIN;
#pragma unroll
        for (int u = 0; u < NDW; ++u) {
            const int ot = (COUT == 64) ? dw_ot0 : u;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = ot * 32 + (r & 3) + 8 * (r >> 2) + 4 * kr;
                out[row * CIN + dw_ct * 32 + c] = acc_dw[u][r];
            }
        }
    }
}

}  // namespace

// Position segments per cloud of the fused narrow backward (a multiple of 64 positions each): as many workgroups as
// the chip holds AT ONCE (3 per CU at 64 outputs, 2 at 128: LDS), so that the launch is a single round -- with 1024
// workgroups on 768 slots the second round ran a third full and cost 20 % (MFMA pipe 48 % busy).
static void narrow_plan(int Cout, int P, int nb, int* seglen, int* segs)
{
    const long long slots = (Cout == 64) ? 768 : 512;
    long long per_cloud = slots / nb;
    if (per_cloud < 1) per_cloud = 1;
    const long long tiles = (P + 63) / 64;
    long long tps = (tiles + per_cloud - 1) / per_cloud;      // 64-position tiles per segment
    if (tps < 4) tps = 4;
    *seglen = (int)(tps * 64);
    *segs = (int)((P + *seglen - 1) / *seglen);
}

extern "C" long long usip_mlp_narrow_backward_workspace(int Cout, int P, int nb)
{
    int seglen, segs;
    narrow_plan(Cout, P, nb, &seglen, &segs);
    return (long long)nb * segs * Cout * 64;
}

// 1 when usip_mlp_narrow_backward_f32 supports the shape (64 inputs, 64 or 128 outputs, positions a multiple of 64).
extern "C" int usip_mlp_narrow_backward_supported(int Cin, int Cout, int P)
{
    return (Cin == 64 && (Cout == 64 || Cout == 128) && P > 0 && P % 64 == 0) ? 1 : 0;
}

// dX[b][ci][p] = sum_co W[co*ldw + ci] * dY[b][co][p]  and  dW[co*lddw + ci] = sum_{b,p} dY[b][co][p] * act(X)[b][ci][p]
// with dY = BatchNorm'(ReLU'(dZ)) rebuilt from (dZ, Y, coef4) as in usip_mlp_gemm_f32 (pro = 2) and
// act(X) = relu(X * xcoef[0] + xcoef[1]) (xcoef may be NULL: X is used as is).  X points at the first of the 64 input
// rows inside a [nb][x_rows][P] tensor, dX likewise inside [nb][dx_rows][P]; all pointers 16-B aligned.
// red_partial (may be NULL; needs xcoef = the producing layer's [4][64] forward coefficients): receives
// [2][blocks][64] partial BatchNorm-backward sums of dX against X (see RED above) and behind them [blocks] maxima of
// |dX [relu on]|, blocks = usip_mlp_narrow_backward_blocks(Cout, P, nb).
extern "C" int usip_mlp_narrow_backward_blocks(int Cout, int P, int nb)
{
    int seglen, segs;
    narrow_plan(Cout, P, nb, &seglen, &segs);
    return nb * segs;
}

extern "C" int usip_mlp_narrow_backward_f32(const float* dZ, const float* Y, const float* coef4, const float* X,
                                            int x_rows, const float* xcoef, const float* W, int ldw, float* dX,
                                            int dx_rows, float* workspace, float* dW, int lddw, float* red_partial,
                                            int Cin, int Cout, int P, int nb, void* stream)
{
    if (red_partial && !xcoef) return USIP_EINVAL;
    if (!usip_mlp_narrow_backward_supported(Cin, Cout, P) || nb < 1 || x_rows < Cin || dx_rows < Cin || ldw < Cin ||
        lddw < Cin)
        return USIP_EINVAL;
    if (!dZ || !Y || !coef4 || !X || !W || !dX || !workspace || !dW) return USIP_EINVAL;
    if ((reinterpret_cast<uintptr_t>(dZ) | reinterpret_cast<uintptr_t>(Y) | reinterpret_cast<uintptr_t>(X) |
         reinterpret_cast<uintptr_t>(dX)) & 15u)
        return USIP_EINVAL;
    int seglen, segs;
    narrow_plan(Cout, P, nb, &seglen, &segs);
    NarrowArgs a{dZ, Y, coef4, X, xcoef, W, ldw, dX, dx_rows, workspace, x_rows, P, nb, seglen, segs, red_partial};
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)(nb * segs)), block(256);
    if (Cout == 64) {
        if (red_partial) USIP_LAUNCH((narrow_bwd_kernel<64, true, true>), grid, block, 0, st, a);
        else if (xcoef) USIP_LAUNCH((narrow_bwd_kernel<64, true, false>), grid, block, 0, st, a);
        else USIP_LAUNCH((narrow_bwd_kernel<64, false, false>), grid, block, 0, st, a);
    } else {
        if (red_partial) USIP_LAUNCH((narrow_bwd_kernel<128, true, true>), grid, block, 0, st, a);
        else if (xcoef) USIP_LAUNCH((narrow_bwd_kernel<128, true, false>), grid, block, 0, st, a);
        else USIP_LAUNCH((narrow_bwd_kernel<128, false, false>), grid, block, 0, st, a);
    }
    USIP_LAUNCH_CHECK();
    return usip_mlp::launch_wgrad_reduce(workspace, dW, (long long)Cout * Cin, nb * segs, Cin, lddw, 0, st);
}
