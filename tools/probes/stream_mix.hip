// What a kernel that reads R streams and writes W streams of 134 MB each can move on this chip -- the ceiling the
// front-end streaming kernels (layer_bwd_x2: 3 R + 1 W, 4 R + 1 W, 2 R + 1.3 W; narrow_fwd / gemm_x2r: 1 R + 1 W, 1 R + 2 W)
// are to be read against.  No matrix work, no LDS: float4 per lane, U independent loads in
// flight per stream, persistent workgroups or one chunk per workgroup, plain or non-temporal stores.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/stream_mix.hip -o /tmp/stream_mix && /tmp/stream_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Ptrs { const float4* r[4]; float4* w[2]; };

template <int R, int W, int U, bool NT>
__global__ __launch_bounds__(256) void mix_kernel(Ptrs p, long long n4)
{
    const long long stride = (long long)gridDim.x * 256 * U;
    for (long long base = (long long)blockIdx.x * 256 * U + threadIdx.x; base < n4; base += stride) {
        float4 v[R][U];
#pragma unroll
        for (int s = 0; s < R; ++s)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long long i = base + u * 256;
                v[s][u] = i < n4 ? p.r[s][i] : make_float4(0, 0, 0, 0);
            }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float4 a = v[0][u];
#pragma unroll
            for (int s = 1; s < R; ++s) { a.x += v[s][u].x; a.y = fmaf(a.y, 1.0001f, v[s][u].y); a.z += v[s][u].z; a.w += v[s][u].w; }
            const long long i = base + u * 256;
            if (i < n4) {
#pragma unroll
                for (int t = 0; t < W; ++t) {
                    float4 o = a; o.x += (float)t;
                    typedef float f4v __attribute__((ext_vector_type(4)));
                    if (NT) { f4v q = {o.x, o.y, o.z, o.w}; __builtin_nontemporal_store(q, (f4v*)&p.w[t][i]); } else p.w[t][i] = o;
                }
            }
        }
    }
}

template <int R, int W, int U, bool NT>
static int run(const char* label, std::vector<float*>& bufs, long long n, int grid)
{
    // two disjoint sets of (R + W) buffers used alternately: every launch's inputs were last touched 2 launches and
    // >= 1 GB of traffic ago (Infinity Cache 256 MB)
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 20;
    Ptrs p[2];
    for (int h = 0; h < 2; ++h) {
        for (int s = 0; s < R; ++s) p[h].r[s] = (const float4*)bufs[h * 6 + s];
        for (int t = 0; t < W; ++t) p[h].w[t] = (float4*)bufs[h * 6 + 4 + t];
    }
    const long long n4 = n / 4;
    if (grid == 0) grid = (int)((n4 + 256LL * U - 1) / (256LL * U));
    for (int i = 0; i < 3; ++i) mix_kernel<R, W, U, NT><<<grid, 256>>>(p[i & 1], n4);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) mix_kernel<R, W, U, NT><<<grid, 256>>>(p[i & 1], n4);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps, mb = (double)(R + W) * n * 4 / 1e6;
    printf("%-26s R=%d W=%d U=%d nt=%d grid=%7d  %8.1f us  %7.1f MB  %6.2f TB/s\n", label, R, W, U, (int)NT, grid, us, mb, mb / us);
    return 0;
}

int main()
{
    const long long n = 64LL * 524288;                      // one 64-channel activation of the step: 134 MB
    std::vector<float*> bufs(12);
    for (auto& b : bufs) { CK(hipMalloc(&b, n * 4)); CK(hipMemset(b, 0, n * 4)); }
    int cus = 256;
#define ROW(R, W) \
    run<R, W, 1, false>("chunk/wg", bufs, n, 0); run<R, W, 2, false>("chunk/wg", bufs, n, 0); run<R, W, 4, false>("chunk/wg", bufs, n, 0); \
    run<R, W, 4, true>("chunk/wg nt", bufs, n, 0); \
    run<R, W, 2, false>("persistent 2 wg/cu", bufs, n, 2 * cus); run<R, W, 4, false>("persistent 2 wg/cu", bufs, n, 2 * cus); \
    run<R, W, 4, false>("persistent 4 wg/cu", bufs, n, 4 * cus); run<R, W, 4, false>("persistent 8 wg/cu", bufs, n, 8 * cus); \
    run<R, W, 4, true>("persistent 8 wg/cu nt", bufs, n, 8 * cus); printf("\n");
    ROW(1, 1) ROW(1, 2) ROW(2, 1) ROW(3, 1) ROW(4, 1) ROW(4, 2)
    return 0;
}
