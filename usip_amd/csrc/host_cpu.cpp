// usip_amd/csrc/host_cpu.cpp -- host twins of the reference's explicitly-CPU entry points
// (index_max.forward_cpu / forward_multi_thread_cpu, models/index_max_ext/index_max.cpp:33-112).
// These are part of the reference's API surface, not a fallback: the device entry points never
// route here and raise on host tensors.
#include <algorithm>
#include <thread>
#include <vector>
#include <stdint.h>
#include "../../include/usip_hip.h"

static void index_max_rows(const float* data, const int32_t* index, int32_t* out,
                           int B, int C, int N, int K, int c_begin, int c_end)
{
    std::vector<float> best((size_t)K);
    for (int b = 0; b < B; ++b) {
        const int32_t* idx = index + (size_t)b * N;
        for (int c = c_begin; c < c_end; ++c) {
            const float* row = data + ((size_t)b * C + c) * N;
            int32_t* o = out + ((size_t)b * C + c) * K;
            std::fill(best.begin(), best.end(), -1000.0f);
            std::fill(o, o + K, 0);
            for (int n = 0; n < N; ++n) {
                const int k = idx[n];
                if (row[n] > best[k]) { best[k] = row[n]; o[k] = n; }
            }
        }
    }
}

extern "C" int usip_index_max_f32_cpu(const float* data, const int32_t* index, int32_t* max_idx,
                                      int B, int C, int N, int K, int num_threads)
{
    if (B < 0 || C < 0 || N < 0 || K < 0 || num_threads < 1) return USIP_EINVAL;
    if ((long long)B * C * K == 0) return USIP_OK;
    if (!max_idx || (N > 0 && (!data || !index))) return USIP_EINVAL;
    num_threads = std::min(num_threads, C);
    if (num_threads <= 1) {
        index_max_rows(data, index, max_idx, B, C, N, K, 0, C);
        return USIP_OK;
    }
    std::vector<std::thread> pool;
    const int step = C / num_threads;
    for (int t = 0; t < num_threads; ++t) {
        const int c0 = t * step, c1 = (t == num_threads - 1) ? C : (t + 1) * step;
        pool.emplace_back(index_max_rows, data, index, max_idx, B, C, N, K, c0, c1);
    }
    for (auto& th : pool) th.join();
    return USIP_OK;
}

// ------------------------------------------------------------------------------------------------
// Host twins of the Ball front end's two operators, so that BASELINE configs[0] (N = 1024, M = 64, batch 2 on PyTorch CPU:
// plumbing without a GPU) reaches the product's own code.  The reference has no CPU ball_query (ball_query.cpp:23-31 is
// a stub); these follow the device kernels' contracts (ball_query_cuda.cu:22-46; torch.norm over the three coordinates as
// the pinned platform evaluates it: sqrt(fma(dz,dz, fma(dy,dy, dx*dx)))).  HOST pointers; never reached from the device
// entry points.
#include <cmath>
extern "C" int usip_ball_query_f32_cpu(const float* dist, int32_t* out_idx, float radius, int K, int B, int M, int N)
{
    if (B < 0 || M < 0 || N < 0 || K < 1) return USIP_EINVAL;
    if ((long long)B * M == 0) return USIP_OK;
    if (!out_idx || (N > 0 && !dist)) return USIP_EINVAL;
    for (long long row = 0; row < (long long)B * M; ++row) {
        const float* d = dist + row * N;
        int32_t* o = out_idx + row * K;
        int u = 0;
        for (int n = 0; n < N && u < K; ++n)
            if (d[n] <= radius) o[u++] = n;
        if (u == 0) { std::fill(o, o + K, 0); continue; }
        for (int i = 0; u + i < K; ++i) o[u + i] = o[i % u];    // cyclic padding with the genuine hits
    }
    return USIP_OK;
}

extern "C" int usip_pairwise_dist_f32_cpu(const float* a, const float* x, float* dist, int B, int M, int N)
{
    if (B < 0 || M < 0 || N < 0) return USIP_EINVAL;
    if ((long long)B * M * N == 0) return USIP_OK;
    if (!a || !x || !dist) return USIP_EINVAL;
    for (int b = 0; b < B; ++b) {
        const float* ab = a + (size_t)b * 3 * M;
        const float* xb = x + (size_t)b * 3 * N;
        for (int m = 0; m < M; ++m) {
            float* row = dist + ((size_t)b * M + m) * N;
            const float ax = ab[m], ay = ab[M + m], az = ab[2 * M + m];
            for (int n = 0; n < N; ++n) {
                const float dx = ax - xb[n], dy = ay - xb[N + n], dz = az - xb[2 * N + n];
                float s2 = dx * dx;
                s2 = std::fmaf(dy, dy, s2);
                s2 = std::fmaf(dz, dz, s2);
                row[n] = std::sqrt(s2);
            }
        }
    }
    return USIP_OK;
}

// "usip_hip <version> gfx950 abi=<n> flags=<...>": abi counts incompatible changes of include/usip_hip.h (a signature that
// gained an argument is not detectable through ctypes); flags names what the build MUST have been compiled with --
// USIP_BUILD_FLAGS comes from usip_amd/build.py; a foreign build without it reports "unknown" and is refused.
#ifndef USIP_BUILD_FLAGS
#define USIP_BUILD_FLAGS "unknown"
#endif
extern "C" const char* usip_version(void) { return "usip_hip 0.5 gfx950 abi=5 flags=" USIP_BUILD_FLAGS; }

// Launch-geometry knobs (speed only, never results): 0 = the library's own heuristic.  tools/ sweeps set them to
// measure alternatives on the GPU; the product never does.
#include <string.h>
static int g_tuning[USIP_TUNE_COUNT];
static const char* const g_tuning_names[USIP_TUNE_COUNT] = {
    "index_max_ch", "index_max_unroll", "x3_wgrad_tile", "x3_gemm_tile", "gemm_split3", "index_max_threads", "x2_direct", "r5_forms",
};
extern "C" int usip_tuning_value(int knob) { return (knob >= 0 && knob < USIP_TUNE_COUNT) ? g_tuning[knob] : 0; }
extern "C" int usip_set_tuning(const char* name, int value)
{
    if (!name) return USIP_EINVAL;
    for (int i = 0; i < USIP_TUNE_COUNT; ++i)
        if (strcmp(name, g_tuning_names[i]) == 0) { g_tuning[i] = value; return USIP_OK; }
    return USIP_EINVAL;
}
