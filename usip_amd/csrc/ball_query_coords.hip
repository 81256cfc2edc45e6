// usip_amd/csrc/ball_query_coords.hip -- fused coords-in ball query on gfx950.
//
// Same result, bit for bit, as usip_pairwise_dist_f32 followed by usip_ball_query_f32
// (models/networks.py:694-698 + ball_query_cuda.cu:22-46) without the B x M x N matrix ever
// touching HBM: a wave owns R node rows of one cloud, streams the cloud's coordinates (L2
// resident, 12 B per point) once for all R rows and evaluates the distance test in registers.
//
// Bit-exactness of the fused test: the reference compares sqrt_rn(s) <= radius with
// s = fma(dz,dz,fma(dy,dy,dx*dx)).  sqrt_rn is monotone, so there is a largest float T with
// sqrt_rn(T) <= radius and the test is exactly  s <= T.  T is found once per call by probing
// the neighbours of radius^2 with a correctly rounded sqrt (sqrt_threshold, on the host).
#include "common.h"
#include <cmath>
#include <cstdlib>
#include <limits>

namespace {


// Largest float T with sqrt_rn(T) <= radius (host sqrtf and the device's __fsqrt_rn are both
// correctly rounded IEEE operations, so the threshold can be derived on the host).
static float sqrt_threshold(float radius)
{
    if (!(radius >= 0.0f)) return -1.0f;     // negative or NaN radius: sqrt(s) <= r never holds
    if (std::isinf(radius)) return radius;
    float t = radius * radius;
    if (std::isinf(t)) t = std::numeric_limits<float>::max();
    for (int i = 0; i < 8 && sqrtf(t) > radius; ++i) t = std::nextafterf(t, -1.0f);
    for (int i = 0; i < 8; ++i) {
        const float up = std::nextafterf(t, std::numeric_limits<float>::infinity());
        if (!std::isinf(up) && sqrtf(up) <= radius) t = up; else break;
    }
    return t;
}

// R = node rows per wave: every point a wave loads is tested against R nodes (fewer L2 bytes per test), but a launch has
// only B * M / R waves -- 2048 at R = 4 for the detector's 16 clouds of 512 nodes, two per SIMD, too few to hide the L2
// round trip of the next block of points behind the tests of the current one.
template <bool VEC, int R>
__global__ __launch_bounds__(256) void ball_query_coords_kernel(
    const float* __restrict__ node, const float* __restrict__ x, int32_t* __restrict__ out,
    float T, int K, int M, int N)
{
    extern __shared__ __attribute__((aligned(16))) int smem[];      // [4 waves][R][K]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.y;
    const int m0 = (blockIdx.x * 4 + wave) * R;
    int* lists = smem + wave * R * K;
    const float* xb = x + (long long)b * 3 * N;
    const float* nb = node + (long long)b * 3 * M;

    float ax[R], ay[R], az[R];
    int count[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int m = min(m0 + r, M - 1);
        ax[r] = nb[m]; ay[r] = nb[M + m]; az[r] = nb[2 * M + m];
        count[r] = (m0 + r < M) ? 0 : K;                  // rows past the end are "done"
    }
    if (m0 < M) {
        constexpr int STEP = VEC ? 256 : 64;
        // The scan is bound by load latency (one L2 round trip per 256 points if the loads are issued where
        // they are used): the next block of points is fetched, from a clamped address, before the current one
        // is tested.
        float4 nx = make_float4(0, 0, 0, 0), ny = nx, nz = nx;
        if (VEC) {
            const int i = min(lane * 4, N - 4);
            nx = *reinterpret_cast<const float4*>(xb + i);
            ny = *reinterpret_cast<const float4*>(xb + N + i);
            nz = *reinterpret_cast<const float4*>(xb + 2 * N + i);
        }
        for (int base = 0; base < N; base += STEP) {
            bool done = true;
#pragma unroll
            for (int r = 0; r < R; ++r) done = done && (count[r] >= K);
            if (done) break;
            unsigned bits[R];
            unsigned any = 0;
            if (VEC) {
                const int i = base + lane * 4;
                const float4 px = nx, py = ny, pz = nz;
                const bool ok = i < N;
                {
                    const int in = min(i + STEP, N - 4);
                    nx = *reinterpret_cast<const float4*>(xb + in);
                    ny = *reinterpret_cast<const float4*>(xb + N + in);
                    nz = *reinterpret_cast<const float4*>(xb + 2 * N + in);
                }
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    // two points per packed instruction (same arithmetic, bit for bit)
                    const usip_f32x2 rx = {ax[r], ax[r]}, ry = {ay[r], ay[r]}, rz = {az[r], az[r]};
                    const usip_f32x2 s01 = usip_sqdist2(rx, ry, rz, usip_f32x2{px.x, px.y}, usip_f32x2{py.x, py.y},
                                                        usip_f32x2{pz.x, pz.y});
                    const usip_f32x2 s23 = usip_sqdist2(rx, ry, rz, usip_f32x2{px.z, px.w}, usip_f32x2{py.z, py.w},
                                                        usip_f32x2{pz.z, pz.w});
                    unsigned h = (s01.x <= T ? 1u : 0u) | (s01.y <= T ? 2u : 0u) | (s23.x <= T ? 4u : 0u) |
                                 (s23.y <= T ? 8u : 0u);
                    bits[r] = (ok && count[r] < K) ? h : 0u;
                    any |= bits[r];
                }
            } else {
                const int i = base + lane;
                const bool ok = i < N;
                const float px = ok ? xb[i] : 0.f, py = ok ? xb[N + i] : 0.f, pz = ok ? xb[2 * N + i] : 0.f;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    unsigned h = usip_sqdist(ax[r], ay[r], az[r], px, py, pz) <= T ? 1u : 0u;
                    bits[r] = (ok && count[r] < K) ? h : 0u;
                    any |= bits[r];
                }
            }
            if (__ballot(any != 0u) == 0ull) continue;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                int* list = lists + r * K;
                if (VEC) {
                    unsigned long long m0b = __ballot(bits[r] & 1u), m1b = __ballot(bits[r] & 2u);
                    unsigned long long m2b = __ballot(bits[r] & 4u), m3b = __ballot(bits[r] & 8u);
                    if ((m0b | m1b | m2b | m3b) == 0ull) continue;
                    int p = count[r] + usip_mbcnt(m0b) + usip_mbcnt(m1b) + usip_mbcnt(m2b) + usip_mbcnt(m3b);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (bits[r] & (1u << j)) {
                            if (p < K) list[p] = base + lane * 4 + j;
                            ++p;
                        }
                    }
                    count[r] += __popcll(m0b) + __popcll(m1b) + __popcll(m2b) + __popcll(m3b);
                } else {
                    unsigned long long mb = __ballot(bits[r] & 1u);
                    if (mb == 0ull) continue;
                    int p = count[r] + usip_mbcnt(mb);
                    if ((bits[r] & 1u) && p < K) list[p] = base + lane;
                    count[r] += __popcll(mb);
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int m = m0 + r;
        if (m < M) {
            const int u = min(count[r], K);
            int32_t* orow = out + ((long long)b * M + m) * K;
            for (int j = lane; j < K; j += 64) orow[j] = (u > 0) ? lists[r * K + (j % u)] : 0;
        }
    }
}

}  // namespace

extern "C" int usip_ball_query_coords_f32(const float* node, const float* x, int32_t* out_idx,
                                          float radius, int K, int B, int M, int N, void* stream)
{
    if (B < 0 || M < 0 || N < 0 || K < 0) return USIP_EINVAL;
    if ((long long)B * M == 0 || K == 0) return USIP_OK;
    if (!node || !x || !out_idx) return USIP_EINVAL;
    if (K > 1024 || B > 65535) return USIP_EINVAL;           // LDS: 4*R*K*4 B = 64 KiB at K=1024
    const float T = sqrt_threshold(radius);
    hipStream_t st = (hipStream_t)stream;
    const bool vec = (N >= 4) && (N % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15u) == 0);
    static const int forced = [] { const char* e = getenv("USIP_BQ_ROWS"); return e ? atoi(e) : 0; }();
    // rows per wave: 4 when that still gives every SIMD four waves (>= 4096 waves), else 2, else 1
    const long long rows = (long long)B * M;
    int R = rows >= 4 * 4096 ? 4 : (rows >= 2 * 4096 ? 2 : 1);
    if (forced == 1 || forced == 2 || forced == 4) R = forced;
    dim3 grid(usip_ceil_div(M, 4 * R), B), block(256);
    const size_t lds = (size_t)4 * R * K * sizeof(int);
#define USIP_BQC(V_, R_) USIP_LAUNCH((ball_query_coords_kernel<V_, R_>), grid, block, lds, st, node, x, out_idx, T, K, M, N)
    if (vec) { if (R == 4) USIP_BQC(true, 4); else if (R == 2) USIP_BQC(true, 2); else USIP_BQC(true, 1); }
    else     { if (R == 4) USIP_BQC(false, 4); else if (R == 2) USIP_BQC(false, 2); else USIP_BQC(false, 1); }
#undef USIP_BQC
    USIP_LAUNCH_CHECK();
    return USIP_OK;
}
