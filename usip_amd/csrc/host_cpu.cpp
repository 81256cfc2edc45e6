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

extern "C" const char* usip_version(void) { return "usip_hip 0.2 gfx950"; }

// Launch-geometry knobs (speed only, never results): 0 = the library's own heuristic.  tools/ sweeps set them to
// measure alternatives on the GPU; the product never does.
#include <string.h>
static int g_tuning[USIP_TUNE_COUNT];
static const char* const g_tuning_names[USIP_TUNE_COUNT] = {
    "index_max_ch", "index_max_unroll", "x3_wgrad_tile", "x3_gemm_tile", "gemm_split3", "index_max_threads", "x2_direct",
};
extern "C" int usip_tuning_value(int knob) { return (knob >= 0 && knob < USIP_TUNE_COUNT) ? g_tuning[knob] : 0; }
extern "C" int usip_set_tuning(const char* name, int value)
{
    if (!name) return USIP_EINVAL;
    for (int i = 0; i < USIP_TUNE_COUNT; ++i)
        if (strcmp(name, g_tuning_names[i]) == 0) { g_tuning[i] = value; return USIP_OK; }
    return USIP_EINVAL;
}
