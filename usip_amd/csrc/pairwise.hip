// usip_amd/csrc/pairwise.hip -- materialised node->point distance matrix on gfx950.
//
// Replaces torch.norm(node.unsqueeze(3) - x.unsqueeze(2), dim=1) (models/networks.py:694-696),
// which in the reference first writes a B x 3 x M x N difference tensor (1.6 GB at B'=16,
// M=512, N=16384) and then reduces it.  Here the B x M x N result is produced directly:
// each lane keeps 4 consecutive points' coordinates in registers and sweeps TM node rows over
// them, storing 16 B per lane per row (fully coalesced).  Store-bound: 4*B*M*N bytes out.
#include "common.h"

namespace {

constexpr int TM = 8;

template <bool VEC>
__global__ __launch_bounds__(256) void pairwise_dist_kernel(
    const float* __restrict__ a, const float* __restrict__ x, float* __restrict__ dist, int M, int N)
{
    const int b = blockIdx.z;
    const int m0 = blockIdx.y * TM;
    const float* xb = x + (long long)b * 3 * N;
    const float* ab = a + (long long)b * 3 * M;
    float* db = dist + (long long)b * M * N;
    if (VEC) {
        const int n = (blockIdx.x * 256 + threadIdx.x) * 4;
        if (n >= N) return;
        const float4 px = *reinterpret_cast<const float4*>(xb + n);
        const float4 py = *reinterpret_cast<const float4*>(xb + N + n);
        const float4 pz = *reinterpret_cast<const float4*>(xb + 2 * N + n);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + i;
            if (m < M) {
                const float ax = ab[m], ay = ab[M + m], az = ab[2 * M + m];
                float4 d;
                d.x = usip_dist(ax, ay, az, px.x, py.x, pz.x);
                d.y = usip_dist(ax, ay, az, px.y, py.y, pz.y);
                d.z = usip_dist(ax, ay, az, px.z, py.z, pz.z);
                d.w = usip_dist(ax, ay, az, px.w, py.w, pz.w);
                *reinterpret_cast<float4*>(db + (long long)m * N + n) = d;
            }
        }
    } else {
        const int n = blockIdx.x * 256 + threadIdx.x;
        if (n >= N) return;
        const float px = xb[n], py = xb[N + n], pz = xb[2 * N + n];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + i;
            if (m < M) db[(long long)m * N + n] = usip_dist(ab[m], ab[M + m], ab[2 * M + m], px, py, pz);
        }
    }
}

}  // namespace

extern "C" int usip_pairwise_dist_f32(const float* a, const float* x, float* dist,
                                      int B, int M, int N, void* stream)
{
    if (B < 0 || M < 0 || N < 0) return USIP_EINVAL;
    if ((long long)B * M * N == 0) return USIP_OK;
    if (!a || !x || !dist) return USIP_EINVAL;
    if (B > 65535 || usip_ceil_div(M, TM) > 65535) return USIP_EINVAL;
    const bool vec = (N % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15u) == 0) &&
                     ((reinterpret_cast<uintptr_t>(dist) & 15u) == 0);
    hipStream_t st = (hipStream_t)stream;
    dim3 block(256);
    if (vec) {
        dim3 grid(usip_ceil_div(N, 1024), usip_ceil_div(M, TM), B);
        USIP_LAUNCH((pairwise_dist_kernel<true>), grid, block, 0, st, a, x, dist, M, N);
    } else {
        dim3 grid(usip_ceil_div(N, 256), usip_ceil_div(M, TM), B);
        USIP_LAUNCH((pairwise_dist_kernel<false>), grid, block, 0, st, a, x, dist, M, N);
    }
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}
