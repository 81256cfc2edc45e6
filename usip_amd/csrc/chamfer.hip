// usip_amd/csrc/chamfer.hip -- the sigma arithmetic of the probabilistic chamfer loss as
// one forward and one backward kernel.
//
// Reference: models/losses.py:68-99 (ChamferLoss_Brute.forward after the two min/arg-min reductions):
//     s_f = (sigma_src + sigma_dst[J]) / 2        loss_f = mean(log s_f + a / s_f)
//     s_b = (sigma_dst + sigma_src[I]) / 2        loss_b = mean(log s_b + c / s_b)
//     pure = mean(a) + mean(c)        weighted = mean(w_f a) + mean(w_b c),  w = (1/s) / mean(1/s)
// which ATen runs as ~50 element-wise / reduction launches over B x 512 values forward and ~60 backward.
// a / J (c / I) are the row (column) minima and arg-minima of the pairwise keypoint distance from
// nearest.hip.  Sums are accumulated in double in a fixed order (deterministic); the scatter of the
// gathered sigma's gradient is a deterministic segmented sum, not atomics.
#include "common.h"

namespace {

// four block sums at once: wave sums by shuffles, then the waves' partials in fixed order (red: [4][16] doubles)
__device__ __forceinline__ void block_sum4(double (&v)[4], double* red)
{
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[i] += __shfl_down(v[i], off);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();                                   // red may still be read from the previous call
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) red[i * 16 + wave] = v[i];
    }
    __syncthreads();
    const int nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        double s = 0.0;
        for (int w = 0; w < nw; ++w) s += red[i * 16 + w];
        v[i] = s;
    }
}

// one direction: sums of (log s + d/s), d, 1/s, d/s over B*M elements
__device__ __forceinline__ void side_sums(const float* __restrict__ d, const int* __restrict__ arg,
                                          const float* __restrict__ s_own, const float* __restrict__ s_other,
                                          int B, int M, int N, double (&out)[4], double* red)
{
    double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
    const long long total = (long long)B * M;
    long long e = threadIdx.x;
    // four elements per pass, their (dependent: arg -> sigma) loads issued together: the kernel is one workgroup and
    // was a chain of 2 x 8 x two load latencies (26 us); the additions keep their order
    for (; e + 3LL * blockDim.x < total; e += 4LL * blockDim.x) {
        float so[4], st[4], dv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long long ei = e + (long long)i * blockDim.x;
            const int b = (int)(ei / M);
            so[i] = s_own[ei];
            st[i] = s_other[(long long)b * N + arg[ei]];
            dv[i] = d[ei];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float s = (so[i] + st[i]) / 2.0f;
            const float r = 1.0f / s, q = dv[i] / s;
            t0 += (double)(logf(s) + q);
            t1 += (double)dv[i];
            t2 += (double)r;
            t3 += (double)(r * dv[i]);
        }
    }
    for (; e < total; e += blockDim.x) {
        const int b = (int)(e / M);
        const float s = (s_own[e] + s_other[(long long)b * N + arg[e]]) / 2.0f;
        const float dv = d[e];
        const float r = 1.0f / s, q = dv / s;
        t0 += (double)(logf(s) + q);
        t1 += (double)dv;
        t2 += (double)r;
        t3 += (double)(r * dv);            // the reference forms (1/s) * d for the weighted chamfer
    }
    out[0] = t0; out[1] = t1; out[2] = t2; out[3] = t3;
    block_sum4(out, red);
}

__global__ __launch_bounds__(1024) void chamfer_prob_fwd_kernel(
    const float* __restrict__ a, const int* __restrict__ J, const float* __restrict__ c,
    const int* __restrict__ I, const float* __restrict__ ss, const float* __restrict__ sd,
    float* __restrict__ out, int B, int M, int N)
{
    __shared__ double red[64];
    double f[4], g[4];
    side_sums(a, J, ss, sd, B, M, N, f, red);
    side_sums(c, I, sd, ss, B, N, M, g, red);
    if (threadIdx.x == 0) {
        const double nf = (double)B * M, nb = (double)B * N;
        out[0] = (float)(f[0] / nf) + (float)(g[0] / nb);            // forward_loss + backward_loss
        out[1] = (float)(f[1] / nf) + (float)(g[1] / nb);            // chamfer_pure
        out[2] = (float)(f[3] / f[2]) + (float)(g[3] / g[2]);        // mean(w a) = sum(a/s) / sum(1/s)
    }
}

// Backward.  With h_f[m] = (dL/ds_f[m]) / 2 and h_b[n] = (dL/ds_b[n]) / 2:
//   d sigma_src[m] = h_f[m] + sum_{n: I[n]==m} h_b[n]          d sigma_dst[n] = h_b[n] + sum_{m: J[m]==n} h_f[m]
// A workgroup owns 64 targets of one pair; its four waves each scan a quarter of the sources (staged in LDS,
// read as 16-B vectors) and the quarters are combined in a fixed order: deterministic, no atomics, no
// workspace (the halves are recomputed from (d, arg, sigma) where they are needed).
constexpr int CHUNK = 1024;

struct Side {                                  // one direction of the loss, row b
    const float* d; const int* arg; const float* s_own; const float* s_other; float scale;
    __device__ __forceinline__ float half(int i, float* dd) const
    {
        const float s = (s_own[i] + s_other[arg[i]]) / 2.0f;
        const float r = 1.0f / s;
        if (dd) dd[i] = scale * r;                                        // dL/dd
        return 0.5f * (scale * (r - d[i] * r * r));
    }
};

__global__ __launch_bounds__(256) void chamfer_prob_bwd_kernel(
    const float* __restrict__ gloss, const float* __restrict__ a, const int* __restrict__ J,
    const float* __restrict__ c, const int* __restrict__ I, const float* __restrict__ ss,
    const float* __restrict__ sd, float* __restrict__ da, float* __restrict__ dc, float* __restrict__ dss,
    float* __restrict__ dsd, int B, int M, int N)
{
    __shared__ __attribute__((aligned(16))) int s_arg[CHUNK];
    __shared__ __attribute__((aligned(16))) float s_val[CHUNK];
    __shared__ float part[4][64];
    const int b = blockIdx.y, tl = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int t = blockIdx.x * 64 + tl;
    const float g = gloss[0];
    const long long om = (long long)b * M, on = (long long)b * N;
    const Side fwd{a + om, J + om, ss + om, sd + on, g / ((float)B * (float)M)};
    const Side bwd{c + on, I + on, sd + on, ss + om, g / ((float)B * (float)N)};
    for (int pass = 0; pass < 2; ++pass) {
        const Side& src = pass == 0 ? fwd : bwd;          // pass 0: h_f[m] lands on sigma_dst[J[m]]
        const Side& own = pass == 0 ? bwd : fwd;
        const int ns = pass == 0 ? M : N, nd = pass == 0 ? N : M;
        float* dd = (pass == 0 ? dc + on : da + om);       // the target side's distance gradient
        float* dst = (pass == 0 ? dsd + on : dss + om);
        float acc = 0.f;
        for (int c0 = 0; c0 < ns; c0 += CHUNK) {
            const int len = min(CHUNK, ns - c0);
            const int len4 = ((len + 15) / 16) * 4;        // sources per wave, a multiple of 4
            __syncthreads();
            for (int i = threadIdx.x; i < 4 * len4; i += 256) {
                s_arg[i] = (i < len) ? src.arg[c0 + i] : -1;
                s_val[i] = (i < len) ? src.half(c0 + i, nullptr) : 0.f;
            }
            __syncthreads();
            for (int i = q * len4; i < (q + 1) * len4; i += 4) {
                const int4 k = *reinterpret_cast<const int4*>(&s_arg[i]);
                const float4 v = *reinterpret_cast<const float4*>(&s_val[i]);
                acc += (k.x == t) ? v.x : 0.f;
                acc += (k.y == t) ? v.y : 0.f;
                acc += (k.z == t) ? v.z : 0.f;
                acc += (k.w == t) ? v.w : 0.f;
            }
        }
        part[q][tl] = acc;
        __syncthreads();
        if (q == 0 && t < nd)
            dst[t] = own.half(t, dd) + ((part[0][tl] + part[1][tl]) + (part[2][tl] + part[3][tl]));
        __syncthreads();
    }
}

}  // namespace

extern "C" int usip_chamfer_prob_f32(const float* a, const int32_t* J, const float* c, const int32_t* I,
                                     const float* sigma_src, const float* sigma_dst, float* out3,
                                     int B, int M, int N, void* stream)
{
    if (B < 1 || M < 1 || N < 1 || !a || !J || !c || !I || !sigma_src || !sigma_dst || !out3) return USIP_EINVAL;
    USIP_LAUNCH(chamfer_prob_fwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a, J, c, I, sigma_src,
                sigma_dst, out3, B, M, N);
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}

extern "C" int usip_chamfer_prob_backward_f32(const float* gloss, const float* a, const int32_t* J, const float* c,
                                              const int32_t* I, const float* sigma_src, const float* sigma_dst,
                                              float* da, float* dc, float* dsigma_src, float* dsigma_dst,
                                              int B, int M, int N, void* stream)
{
    if (B < 1 || M < 1 || N < 1 || !gloss || !a || !J || !c || !I || !sigma_src || !sigma_dst || !da || !dc ||
        !dsigma_src || !dsigma_dst || B > 65535)
        return USIP_EINVAL;
    const int mx = M > N ? M : N;
    USIP_LAUNCH(chamfer_prob_bwd_kernel, dim3(usip_ceil_div(mx, 64), B), dim3(256), 0, (hipStream_t)stream, gloss, a,
                J, c, I, sigma_src, sigma_dst, da, dc, dsigma_src, dsigma_dst, B, M, N);
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}
