// usip_amd/csrc/fps.hip -- farthest-point sampling of SOM nodes on gfx950 and the
// inference-side non-maximum suppression of keypoints.
//
// f-3  Reference: FarthestSampler.sample (data/kitti_detector_loader.py:69-83), run per cloud in DataLoader
// workers on a random N/3 subset: start from a given point, then k-1 times take the arg-max of the running
// minimum squared distance.  numpy evaluates it in float64 (the zeros() buffer is float64), first maximum
// wins.  One workgroup per cloud keeps the running minima in registers (float64, same arithmetic order:
// (dx*dx + dy*dy) + dz*dz, no FMA) and does k-1 block-wide arg-max reductions; the selected indices are
// bit-identical to numpy's.
//
// f-4  Reference: nms() + top-k by sigma (evaluation/save_keypoints.py:180-216, :346-351): repeatedly keep the
// remaining keypoint with the smallest sigma and drop everything within NMS_radius of it (float32
// np.linalg.norm: sqrt((dx*dx + dy*dy) + dz*dz) > radius survives).
#include "common.h"

namespace {

constexpr int FPS_T = 1024;          // threads per cloud
constexpr int FPS_PPT = 16;          // points per thread -> n <= 16384

__global__ __launch_bounds__(FPS_T) void fps_kernel(
    const float* __restrict__ pts, const int32_t* __restrict__ first, int32_t* __restrict__ out, int n, int k)
{
    __shared__ double red_v[FPS_T / 64];
    __shared__ int red_i[FPS_T / 64];
    __shared__ int sel;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* p = pts + (long long)b * 3 * n;
    double dist[FPS_PPT];
#pragma unroll
    for (int t = 0; t < FPS_PPT; ++t) dist[t] = __builtin_inf();
    int cur = first[b];
    if (tid == 0) out[(long long)b * k] = cur;
    for (int it = 1; it < k; ++it) {
        const double cx = (double)p[cur], cy = (double)p[n + cur], cz = (double)p[2 * n + cur];
        double best = -1.0;
        int bi = 0x7fffffff;
#pragma unroll
        for (int t = 0; t < FPS_PPT; ++t) {
            const int j = t * FPS_T + tid;
            if (j < n) {
                const double dx = cx - (double)p[j], dy = cy - (double)p[n + j], dz = cz - (double)p[2 * n + j];
                const double d = (dx * dx + dy * dy) + dz * dz;          // contraction is off
                dist[t] = d < dist[t] ? d : dist[t];                     // np.minimum
                if (dist[t] > best) { best = dist[t]; bi = j; }          // ascending j: first maximum
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double ov = __shfl_xor(best, off);
            const int oi = __shfl_xor(bi, off);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if (lane == 0) { red_v[wave] = best; red_i[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            double v = red_v[0];
            int i = red_i[0];
            for (int w = 1; w < FPS_T / 64; ++w)
                if (red_v[w] > v || (red_v[w] == v && red_i[w] < i)) { v = red_v[w]; i = red_i[w]; }
            sel = i;
            out[(long long)b * k + it] = i;
        }
        __syncthreads();
        cur = sel;
    }
}

// One workgroup per cloud, M <= 1024 keypoints: alive flags in LDS, greedy loop.
__global__ __launch_bounds__(1024) void nms_kernel(
    const float* __restrict__ kp, const float* __restrict__ sigma, float radius, int32_t* __restrict__ order,
    int32_t* __restrict__ count, int M)
{
    __shared__ float red_v[16];
    __shared__ int red_i[16];
    __shared__ int sel;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* p = kp + (long long)b * 3 * M;
    const bool has = tid < M;
    const float x = has ? p[tid] : 0.f, y = has ? p[M + tid] : 0.f, z = has ? p[2 * M + tid] : 0.f;
    const float s = has ? sigma[(long long)b * M + tid] : 0.f;
    bool alive = has;
    int n_out = 0;
    for (;;) {
        float best = alive ? s : __builtin_inff();
        int bi = alive ? tid : 0x7fffffff;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(best, off);
            const int oi = __shfl_xor(bi, off);
            if (oi != 0x7fffffff && (bi == 0x7fffffff || ov < best || (ov == best && oi < bi))) { best = ov; bi = oi; }
        }
        if (lane == 0) { red_v[wave] = best; red_i[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            float v = red_v[0];
            int i = red_i[0];
            for (int w = 1; w < 16; ++w)
                if (red_i[w] != 0x7fffffff && (i == 0x7fffffff || red_v[w] < v || (red_v[w] == v && red_i[w] < i))) {
                    v = red_v[w]; i = red_i[w];
                }
            sel = i;
            if (i != 0x7fffffff) order[(long long)b * M + n_out] = i;
        }
        __syncthreads();
        const int c = sel;
        if (c == 0x7fffffff) break;
        ++n_out;
        const float dx = p[c] - x, dy = p[M + c] - y, dz = p[2 * M + c] - z;
        const float d = sqrtf((dx * dx + dy * dy) + dz * dz);
        if (!(d > radius)) alive = false;                     // the selected point itself has d == 0
        __syncthreads();
    }
    if (tid == 0) count[b] = n_out;
}

}  // namespace

extern "C" int usip_fps_f32(const float* pts, const int32_t* first_idx, int32_t* out_idx,
                            int B, int n, int k, void* stream)
{
    if (B < 0 || n < 1 || k < 1 || k > n) return USIP_EINVAL;
    if (B == 0) return USIP_OK;
    if (!pts || !first_idx || !out_idx || n > FPS_T * FPS_PPT) return USIP_EINVAL;
    USIP_LAUNCH(fps_kernel, dim3(B), dim3(FPS_T), 0, (hipStream_t)stream, pts, first_idx, out_idx, n, k);
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}

extern "C" int usip_nms_f32(const float* keypoints, const float* sigmas, float radius, int32_t* order,
                            int32_t* count, int B, int M, void* stream)
{
    if (B < 0 || M < 1 || M > 1024) return USIP_EINVAL;
    if (!(radius >= 0.0f)) return USIP_EINVAL;        // negative or NaN: the selected point would never suppress itself
    if (B == 0) return USIP_OK;
    if (!keypoints || !sigmas || !order || !count) return USIP_EINVAL;
    USIP_LAUNCH(nms_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, keypoints, sigmas, radius, order, count, M);
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}
