// usip_amd/csrc/head.hip -- the element-wise tail of the detector step as three small kernels (gfx950).
//
// Between the last shared-MLP layer and the nearest-neighbour reductions the reference runs a dozen element-wise ATen
// launches over B x 4 x M values forward and as many backward -- split, add, softplus, add, the scaled rigid transform
// (mul + baddbmm), means, a sum -- each ~5 us of launch latency inside the captured step:
//     keypoints = mlp3[:, :3] + centre,  sigmas = softplus(mlp3[:, 3]) + lower_bound        (models/networks.py:150-154)
//     kp_t = (R * scale) . keypoints_src + shift                                             (keypoint_detector.py:182-184)
//     loss = loss_chamfer + alpha * (mean(d_src) + mean(d_dst))                               (keypoint_detector.py:196-204)
// Same arithmetic, one launch each way per line.
#include "common.h"

namespace {

constexpr float SOFTPLUS_THRESHOLD = 20.0f;                   // torch.nn.Softplus default (beta = 1)

__global__ __launch_bounds__(256) void head_fwd_kernel(
    const float* __restrict__ ks, const float* __restrict__ centre, float lower, float* __restrict__ kp,
    float* __restrict__ sigma, int M)
{
    const int b = blockIdx.y, m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const float* k = ks + (long long)b * 4 * M;
    const float* c = centre + (long long)b * 3 * M;
    float* o = kp + (long long)b * 3 * M;
    o[m] = k[m] + c[m];
    o[M + m] = k[M + m] + c[M + m];
    o[2 * M + m] = k[2 * M + m] + c[2 * M + m];
    const float x = k[3 * M + m];
    sigma[(long long)b * M + m] = (x > SOFTPLUS_THRESHOLD ? x : log1pf(expf(x))) + lower;
}

__global__ __launch_bounds__(256) void head_bwd_kernel(
    const float* __restrict__ gkp, const float* __restrict__ gsigma, const float* __restrict__ ks,
    float* __restrict__ gks, int M)
{
    const int b = blockIdx.y, m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    float* o = gks + (long long)b * 4 * M;
    const float* g = gkp ? gkp + (long long)b * 3 * M : nullptr;
    o[m] = g ? g[m] : 0.f;
    o[M + m] = g ? g[M + m] : 0.f;
    o[2 * M + m] = g ? g[2 * M + m] : 0.f;
    float gs = 0.f;
    if (gsigma) {
        const float x = ks[(long long)b * 4 * M + 3 * M + m];
        const float z = expf(x);                              // softplus_backward: grad * z / (z + 1)
        gs = gsigma[(long long)b * M + m] * (x > SOFTPLUS_THRESHOLD ? 1.0f : z / (z + 1.0f));
    }
    o[3 * M + m] = gs;
}

// out[b][i][m] = shift[b][i] + sum_j (R[b][i][j] * scale[b]) * x[b][j][m]        (TRANSPOSE: R^T, no shift -- the backward)
template <bool TRANSPOSE>
__global__ __launch_bounds__(256) void rigid_kernel(
    const float* __restrict__ x, const float* __restrict__ R, const float* __restrict__ scale,
    const float* __restrict__ shift, float* __restrict__ out, int M)
{
    const int b = blockIdx.y, m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const float s = scale[b];
    const float* r = R + b * 9;
    const float* xb = x + (long long)b * 3 * M;
    const float x0 = xb[m], x1 = xb[M + m], x2 = xb[2 * M + m];
    float* o = out + (long long)b * 3 * M;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float a0 = (TRANSPOSE ? r[i] : r[3 * i]) * s, a1 = (TRANSPOSE ? r[3 + i] : r[3 * i + 1]) * s,
                    a2 = (TRANSPOSE ? r[6 + i] : r[3 * i + 2]) * s;
        float t = a0 * x0;
        t = __builtin_fmaf(a1, x1, t);
        t = __builtin_fmaf(a2, x2, t);
        o[i * M + m] = TRANSPOSE ? t : shift[b * 3 + i] + t;
    }
}

// out3 = (chamfer + alpha * (mean(d[0:half]) + mean(d[half:])), alpha * mean(d[0:half]), alpha * mean(d[half:]))
__global__ __launch_bounds__(1024) void loss_combine_kernel(
    const float* __restrict__ d, const float* __restrict__ chamfer, float alpha, float* __restrict__ out3,
    long long half)
{
    __shared__ double red[2][16];
    double s0 = 0.0, s1 = 0.0;
    for (long long e = threadIdx.x; e < half; e += 1024) { s0 += (double)d[e]; s1 += (double)d[half + e]; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { s0 += __shfl_down(s0, off); s1 += __shfl_down(s1, off); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s0; red[1][threadIdx.x >> 6] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t0 = 0.0, t1 = 0.0;
        for (int w = 0; w < 16; ++w) { t0 += red[0][w]; t1 += red[1][w]; }
        const float m0 = (float)(t0 / (double)half) * alpha, m1 = (float)(t1 / (double)half) * alpha;
        out3[1] = m0;
        out3[2] = m1;
        out3[0] = chamfer[0] + (m0 + m1);
    }
}

__global__ __launch_bounds__(256) void fill_scaled_kernel(const float* __restrict__ g, float factor,
                                                          float* __restrict__ out, long long n)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = g[0] * factor;
}

// sum over the K neighbours of dY = a1 * dYhat + q1 * y + q0 from the per-neighbourhood sums the BatchNorm-backward
// reduction took (gsum[0] = sum dYhat, gsum[1] = sum y): out = K * q0 + a1 * gsum[0] + q1 * gsum[1], per channel.
__global__ __launch_bounds__(256) void group_dy_sum_kernel(
    const float* __restrict__ g0, const float* __restrict__ g1, const float* __restrict__ coef4,
    float* __restrict__ out, long long total, int C, int G, float K)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)((i / G) % C);
    const float t = K * coef4[3 * C + c];
    out[i] = (t + coef4[c] * g0[i]) + coef4[2 * C + c] * g1[i];
}

}  // namespace

extern "C" int usip_bn_group_dy_sum_f32(const float* gsum0, const float* gsum1, const float* coef4, float* out,
                                        int nb, int C, int G, int K, void* stream)
{
    if (nb < 0 || C < 1 || G < 0 || K < 1) return USIP_EINVAL;
    const long long total = (long long)nb * C * G;
    if (total == 0) return USIP_OK;
    if (!gsum0 || !gsum1 || !coef4 || !out || (total + 255) / 256 > 0x7fffffffLL) return USIP_EINVAL;
    USIP_LAUNCH(group_dy_sum_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                gsum0, gsum1, coef4, out, total, C, G, (float)K);
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}

extern "C" int usip_detector_head_f32(const float* ks, const float* centre, float sigma_lower_bound, float* keypoints,
                                      float* sigmas, int B, int M, void* stream)
{
    if (B < 0 || M < 0) return USIP_EINVAL;
    if ((long long)B * M == 0) return USIP_OK;
    if (!ks || !centre || !keypoints || !sigmas || B > 65535) return USIP_EINVAL;
    USIP_LAUNCH(head_fwd_kernel, dim3(usip_ceil_div(M, 256), B), dim3(256), 0, (hipStream_t)stream,
                ks, centre, sigma_lower_bound, keypoints, sigmas, M);
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}

extern "C" int usip_detector_head_backward_f32(const float* g_keypoints, const float* g_sigmas, const float* ks,
                                               float* g_ks, int B, int M, void* stream)
{
    if (B < 0 || M < 0) return USIP_EINVAL;
    if ((long long)B * M == 0) return USIP_OK;
    if (!ks || !g_ks || B > 65535) return USIP_EINVAL;
    USIP_LAUNCH(head_bwd_kernel, dim3(usip_ceil_div(M, 256), B), dim3(256), 0, (hipStream_t)stream,
                g_keypoints, g_sigmas, ks, g_ks, M);
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}

extern "C" int usip_rigid_transform_f32(const float* x, const float* R, const float* scale, const float* shift,
                                        float* out, int transpose, int B, int M, void* stream)
{
    if (B < 0 || M < 0) return USIP_EINVAL;
    if ((long long)B * M == 0) return USIP_OK;
    if (!x || !R || !scale || !out || (!transpose && !shift) || B > 65535) return USIP_EINVAL;
    const dim3 grid(usip_ceil_div(M, 256), B), block(256);
    if (transpose) USIP_LAUNCH(rigid_kernel<true>, grid, block, 0, (hipStream_t)stream, x, R, scale, shift, out, M);
    else USIP_LAUNCH(rigid_kernel<false>, grid, block, 0, (hipStream_t)stream, x, R, scale, shift, out, M);
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}

extern "C" int usip_detector_loss_combine_f32(const float* d, const float* chamfer, float alpha, float* out3,
                                              long long half, void* stream)
{
    if (half < 1 || !d || !chamfer || !out3) return USIP_EINVAL;
    USIP_LAUNCH(loss_combine_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, d, chamfer, alpha, out3, half);
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}

extern "C" int usip_fill_scaled_f32(const float* g, float factor, float* out, long long n, void* stream)
{
    if (n < 0) return USIP_EINVAL;
    if (n == 0) return USIP_OK;
    if (!g || !out || (n + 255) / 256 > 0x7fffffffLL) return USIP_EINVAL;
    USIP_LAUNCH(fill_scaled_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, factor,
                out, n);
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}

// ------------------------------------------------------------------------------------------------
// Adam over ONE flat fp32 parameter buffer (models/keypoint_detector.py:42-45, :207: torch.optim.Adam(lr,
// betas=(0.9, 0.999)), eps 1e-8, no weight decay, no amsgrad), the arithmetic of torch's own single-tensor update:
//     m = lerp(m, g, 1 - b1);  v = b2 v + (1 - b2) g g;  p -= (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// The step count t lives on the device (the update is part of a captured HIP graph): a one-thread launch bumps it,
// then every workgroup derives the two bias corrections from it in double precision (as the Python implementation
// does) and broadcasts them through LDS.  torch's fused multi-tensor kernel spread a single 1.2 M-element tensor over
// 19 workgroups (48 us); this is one pass over five arrays at full width.
namespace {

__global__ void adam_bump_kernel(float* step) { step[0] += 1.0f; }

// hyper != null: (lr, beta1, beta2, eps) are read from that device array instead of the launch arguments, so that a
// captured launch follows param_groups[...]['lr'] (models/keypoint_detector.py:356-366 changes it every lr_decay_step
// epochs) without a new capture
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   const float* __restrict__ step, float lr, float b1, float b2,
                                                   float eps, long long n, const float* __restrict__ hyper)
{
    if (hyper) { lr = hyper[0]; b1 = hyper[1]; b2 = hyper[2]; eps = hyper[3]; }
    __shared__ float cf[2];
    if (threadIdx.x == 0) {
        const double t = (double)step[0];
        const double bc1 = 1.0 - pow((double)b1, t), bc2 = 1.0 - pow((double)b2, t);
        cf[0] = (float)((double)lr / bc1);                     // step_size
        cf[1] = (float)sqrt(bc2);                             // bias_correction2_sqrt
    }
    __syncthreads();
    const float step_size = cf[0], bc2s = cf[1];
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
        const float4 gv = *reinterpret_cast<const float4*>(g + i);
        float4 mv = *reinterpret_cast<float4*>(m + i), vv = *reinterpret_cast<float4*>(v + i), pv = *reinterpret_cast<float4*>(p + i);
        const float gg[4] = {gv.x, gv.y, gv.z, gv.w};
        float mm[4] = {mv.x, mv.y, mv.z, mv.w}, vs[4] = {vv.x, vv.y, vv.z, vv.w}, pp[4] = {pv.x, pv.y, pv.z, pv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            mm[e] = mm[e] + (1.0f - b1) * (gg[e] - mm[e]);
            vs[e] = b2 * vs[e] + (1.0f - b2) * gg[e] * gg[e];
            pp[e] = pp[e] - step_size * (mm[e] / (sqrtf(vs[e]) / bc2s + eps));
        }
        *reinterpret_cast<float4*>(m + i) = make_float4(mm[0], mm[1], mm[2], mm[3]);
        *reinterpret_cast<float4*>(v + i) = make_float4(vs[0], vs[1], vs[2], vs[3]);
        *reinterpret_cast<float4*>(p + i) = make_float4(pp[0], pp[1], pp[2], pp[3]);
    } else {
        for (long long j = i; j < n; ++j) {
            const float ge = g[j];
            const float me = m[j] + (1.0f - b1) * (ge - m[j]);
            const float ve = b2 * v[j] + (1.0f - b2) * ge * ge;
            m[j] = me; v[j] = ve;
            p[j] = p[j] - step_size * (me / (sqrtf(ve) / bc2s + eps));
        }
    }
}

}  // namespace

// One Adam step on flat fp32 buffers (16-B aligned): param, grad, exp_avg, exp_avg_sq [n], step_count [1] (device float,
// incremented first).  Replaces optimizer.step() of models/keypoint_detector.py:207 for the flat parameter buffer.
extern "C" int usip_adam_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* step_count,
                                  float lr, float beta1, float beta2, float eps, long long n, void* stream)
{
    if (n < 0) return USIP_EINVAL;
    if (n == 0) return USIP_OK;
    if (!param || !grad || !exp_avg || !exp_avg_sq || !step_count) return USIP_EINVAL;
    if ((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(exp_avg) |
         reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15u)
        return USIP_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    USIP_LAUNCH(adam_bump_kernel, dim3(1), dim3(1), 0, st, step_count);
    USIP_LAUNCH_CHECK();
    USIP_LAUNCH(adam_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, st, param, grad, exp_avg, exp_avg_sq,
                step_count, lr, beta1, beta2, eps, n, (const float*)nullptr);
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}

// The same with the hyper-parameters in device memory: hyper[4] = (lr, beta1, beta2, eps).
extern "C" int usip_adam_step_hyper_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                                        float* step_count, const float* hyper, long long n, void* stream)
{
    if (n < 0) return USIP_EINVAL;
    if (n == 0) return USIP_OK;
    if (!param || !grad || !exp_avg || !exp_avg_sq || !step_count || !hyper) return USIP_EINVAL;
    if ((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(exp_avg) |
         reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15u)
        return USIP_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    USIP_LAUNCH(adam_bump_kernel, dim3(1), dim3(1), 0, st, step_count);
    USIP_LAUNCH_CHECK();
    USIP_LAUNCH(adam_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, st, param, grad, exp_avg, exp_avg_sq,
                step_count, 0.f, 0.f, 0.f, 0.f, n, hyper);
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}
