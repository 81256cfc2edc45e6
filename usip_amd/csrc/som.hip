// usip_amd/csrc/som.hip -- SOM front end of RPN_Detector on gfx950.
//
// Replaces util/som.py:31-54 (query_topk, k = 1) and models/networks.py:85-108 of the
// reference, which materialise a B x 3 x N x M difference tensor (1.6 GB at B'=16, N=16384,
// M=512), a B x N x M squared-norm tensor and a dense one-hot mask of the same size, then
// multiply/sum them three times.  Algorithmically the path needs only:
//     min_idx[b,n]  = argmin_m  (dx*dx + dy*dy) + dz*dz        (squared, summed in c order,
//                                                              no FMA: pow then sum in ATen)
//     count[b,m], cluster_mean[b,:,m] = sum_{n in m} x / (count + 1e-5)
//     x_decentered[b,:,n] = x[b,:,n] - cluster_mean[b,:,min_idx[b,n]]
// i.e. ~3.4 MB of HBM traffic instead of gigabytes.
#include "common.h"

namespace {

// (Four points per lane with the nodes read four at a time (ds_read_b128) was tried to cut the LDS reads: 66 us
// instead of 50 -- the loop is bound by VALU issue (11 instructions per point-node pair, ~43 us at full rate), not by
// LDS, and one workgroup per CU hides less latency.)
// One lane per point; the cloud's nodes sit in LDS and are broadcast to all lanes.
__global__ __launch_bounds__(256) void som_assign_kernel(
    const float* __restrict__ x, const float* __restrict__ node, int32_t* __restrict__ min_idx,
    int N, int M)
{
    extern __shared__ __attribute__((aligned(16))) float nodes[];        // [3][M]
    const int b = blockIdx.y;
    const float* nb = node + (long long)b * 3 * M;
    for (int i = threadIdx.x; i < 3 * M; i += 256) nodes[i] = nb[i];
    __syncthreads();
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const float* xb = x + (long long)b * 3 * N;
    const float px = xb[n], py = xb[N + n], pz = xb[2 * N + n];
    float best = __builtin_inff();
    int arg = 0;
    // two nodes per packed instruction (v_pk_add/mul_f32: the same roundings in the same order), then the two
    // comparisons in node order
    const usip_f32x2 qx = {px, px}, qy = {py, py}, qz = {pz, pz};
    int m = 0;
    for (; m + 1 < M; m += 2) {
        const usip_f32x2 dx = qx - usip_f32x2{nodes[m], nodes[m + 1]};
        const usip_f32x2 dy = qy - usip_f32x2{nodes[M + m], nodes[M + m + 1]};
        const usip_f32x2 dz = qz - usip_f32x2{nodes[2 * M + m], nodes[2 * M + m + 1]};
        const usip_f32x2 d2 = (dx * dx + dy * dy) + dz * dz;             // contraction is off
        if (d2.x < best) { best = d2.x; arg = m; }                       // first minimum wins
        if (d2.y < best) { best = d2.y; arg = m + 1; }
    }
    if (m < M) {
        const float dx = px - nodes[m], dy = py - nodes[M + m], dz = pz - nodes[2 * M + m];
        const float d2 = (dx * dx + dy * dy) + dz * dz;
        if (d2 < best) { best = d2; arg = m; }
    }
    min_idx[(long long)b * N + n] = arg;
}

// One wave per (b, m): deterministic segmented sum (fixed lane partition + fixed tree).
__global__ __launch_bounds__(256) void som_cluster_kernel(
    const float* __restrict__ x, const int32_t* __restrict__ min_idx,
    float* __restrict__ cluster_mean, int32_t* __restrict__ count, int N, int M)
{
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.y;
    if (m >= M) return;
    const float* xb = x + (long long)b * 3 * N;
    const int32_t* ib = min_idx + (long long)b * N;
    // sums in double: the mean is then the correctly rounded exact mean, whatever the order of the scan
    double sx = 0.0, sy = 0.0, sz = 0.0;
    int c = 0;
    if ((N & 3) == 0 && (reinterpret_cast<uintptr_t>(ib) & 15u) == 0) {
        // the scan is bound by the latency of the index loads (one in flight per wave otherwise): 16-B loads,
        // four of them issued before the first compare; the coordinate loads happen for ~N/M hits only
        int n0 = lane * 4;
        for (; n0 + 3 * 256 < N; n0 += 4 * 256) {
            int4 id[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) id[u] = *reinterpret_cast<const int4*>(ib + n0 + u * 256);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int n = n0 + u * 256;
                if (id[u].x == m) { sx += xb[n]; sy += xb[N + n]; sz += xb[2 * N + n]; c += 1; }
                if (id[u].y == m) { sx += xb[n + 1]; sy += xb[N + n + 1]; sz += xb[2 * N + n + 1]; c += 1; }
                if (id[u].z == m) { sx += xb[n + 2]; sy += xb[N + n + 2]; sz += xb[2 * N + n + 2]; c += 1; }
                if (id[u].w == m) { sx += xb[n + 3]; sy += xb[N + n + 3]; sz += xb[2 * N + n + 3]; c += 1; }
            }
        }
        for (; n0 < N; n0 += 256) {
            const int4 id = *reinterpret_cast<const int4*>(ib + n0);
            if (id.x == m) { sx += xb[n0]; sy += xb[N + n0]; sz += xb[2 * N + n0]; c += 1; }
            if (id.y == m) { sx += xb[n0 + 1]; sy += xb[N + n0 + 1]; sz += xb[2 * N + n0 + 1]; c += 1; }
            if (id.z == m) { sx += xb[n0 + 2]; sy += xb[N + n0 + 2]; sz += xb[2 * N + n0 + 2]; c += 1; }
            if (id.w == m) { sx += xb[n0 + 3]; sy += xb[N + n0 + 3]; sz += xb[2 * N + n0 + 3]; c += 1; }
        }
    } else {
        for (int n = lane; n < N; n += 64) {
            if (ib[n] == m) { sx += xb[n]; sy += xb[N + n]; sz += xb[2 * N + n]; c += 1; }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sx += __shfl_down(sx, off);
        sy += __shfl_down(sy, off);
        sz += __shfl_down(sz, off);
        c += __shfl_down(c, off);
    }
    if (lane == 0) {
        const float denom = (float)c + 1e-5f;                            // networks.py:95-96
        float* cm = cluster_mean + (long long)b * 3 * M;
        cm[m] = (float)sx / denom; cm[M + m] = (float)sy / denom; cm[2 * M + m] = (float)sz / denom;
        count[(long long)b * M + m] = c;
    }
}

// cluster_mean / count from the segments of min_idx (networks.py:87-107): one wave per (cloud, node) -- the lanes read
// the segment's positions coalesced, gather their points, and the wave adds up in double, so the mean is the correctly
// rounded exact mean whatever the order.
__global__ __launch_bounds__(256) void som_cluster_csr_kernel(
    const float* __restrict__ x, const int32_t* __restrict__ start, const int32_t* __restrict__ perm,
    float* __restrict__ cluster_mean, int32_t* __restrict__ count, int N, int M)
{
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.y, m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const int32_t* st = start + (long long)b * (M + 1);
    const int32_t* pm = perm + (long long)b * N;
    const float* xb = x + (long long)b * 3 * N;
    const int s0 = st[m], s1 = st[m + 1];
    double sx = 0.0, sy = 0.0, sz = 0.0;
    for (int j = s0 + lane; j < s1; j += 64) {
        const int p = pm[j];
        sx += xb[p]; sy += xb[N + p]; sz += xb[2 * N + p];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sx += __shfl_down(sx, off);
        sy += __shfl_down(sy, off);
        sz += __shfl_down(sz, off);
    }
    if (lane == 0) {
        const int c = s1 - s0;
        const float denom = (float)c + 1e-5f;                            // networks.py:95-96
        float* cm = cluster_mean + (long long)b * 3 * M;
        cm[m] = (float)sx / denom; cm[M + m] = (float)sy / denom; cm[2 * M + m] = (float)sz / denom;
        count[(long long)b * M + m] = c;
    }
}

__global__ __launch_bounds__(256) void som_decenter_kernel(
    const float* __restrict__ x, const int32_t* __restrict__ min_idx,
    const float* __restrict__ cluster_mean, float* __restrict__ out, int N, int M)
{
    const int b = blockIdx.y;
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const int m = min_idx[(long long)b * N + n];
    const float* cm = cluster_mean + (long long)b * 3 * M;
    const float* xb = x + (long long)b * 3 * N;
    float* ob = out + (long long)b * 3 * N;
    ob[n] = xb[n] - cm[m];
    ob[N + n] = xb[N + n] - cm[M + m];
    ob[2 * N + n] = xb[2 * N + n] - cm[2 * M + m];
}

}  // namespace

extern "C" int usip_som_assign_f32(const float* x, const float* node, int32_t* min_idx,
                                   int B, int N, int M, void* stream)
{
    if (B < 0 || N < 0 || M < 1) return USIP_EINVAL;
    if ((long long)B * N == 0) return USIP_OK;
    // the node table lives in dynamic LDS: 3*M*4 B within the 64 KiB a launch gets without raising the kernel's
    // MaxDynamicSharedMemorySize attribute (node counts on the path are 64..512)
    if (!x || !node || !min_idx || B > 65535 || M > 5461) return USIP_EINVAL;
    dim3 grid(usip_ceil_div(N, 256), B), block(256);
    USIP_LAUNCH(som_assign_kernel, grid, block, (size_t)3 * M * sizeof(float), (hipStream_t)stream,
                       x, node, min_idx, N, M);
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}

extern "C" int usip_som_cluster_f32(const float* x, const int32_t* min_idx, float* cluster_mean,
                                    int32_t* count, float* x_decentered, int B, int N, int M, void* stream)
{
    if (B < 0 || N < 0 || M < 0) return USIP_EINVAL;
    if ((long long)B * M == 0) return USIP_OK;
    if (!min_idx || !cluster_mean || !count || (N > 0 && !x) || B > 65535) return USIP_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    USIP_LAUNCH(som_cluster_kernel, dim3(usip_ceil_div(M, 4), B), dim3(256), 0, st,
                       x, min_idx, cluster_mean, count, N, M);
    USIP_LAUNCH_CHECK();
    if (x_decentered && N > 0) {
        USIP_LAUNCH(som_decenter_kernel, dim3(usip_ceil_div(N, 256), B), dim3(256), 0, st,
                           x, min_idx, cluster_mean, x_decentered, N, M);
        USIP_LAUNCH_CHECK();
    }
    return USIP_OK;
}

// The same outputs from the destination-sorted form of min_idx (usip_csr_by_index_i32, csrc/segment.hip): O(N) instead
// of every node scanning all N assignments.  min_idx is still needed for the decentering.
extern "C" int usip_som_cluster_csr_f32(const float* x, const int32_t* min_idx, const int32_t* start,
                                        const int32_t* perm, float* cluster_mean, int32_t* count,
                                        float* x_decentered, int B, int N, int M, void* stream)
{
    if (B < 0 || N < 0 || M < 0) return USIP_EINVAL;
    if ((long long)B * M == 0) return USIP_OK;
    if (!start || !perm || !cluster_mean || !count || (N > 0 && !x) || B > 65535) return USIP_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    USIP_LAUNCH(som_cluster_csr_kernel, dim3(usip_ceil_div(M, 4), B), dim3(256), 0, st,
                x, start, perm, cluster_mean, count, N, M);
    USIP_LAUNCH_CHECK();
    if (x_decentered && N > 0) {
        if (!min_idx) return USIP_EINVAL;
        USIP_LAUNCH(som_decenter_kernel, dim3(usip_ceil_div(N, 256), B), dim3(256), 0, st,
                    x, min_idx, cluster_mean, x_decentered, N, M);
        USIP_LAUNCH_CHECK();
    }
    return USIP_OK;
}
