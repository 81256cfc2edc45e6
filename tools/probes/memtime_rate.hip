// What does s_memtime count on this chip?  One wave sleeps a known number of shader cycles (s_sleep 127 = 127 x 64
// cycles, 1000 times) and a second kernel spins on dependent VALU adds; ticks vs HIP-event time for both.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/memtime_rate.hip -o /tmp/memtime_rate && /tmp/memtime_rate
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void sleeper(unsigned long long* out, int n)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

__global__ void adder(unsigned long long* out, float* sink, int n)
{
    float x = threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 64; ++j) asm volatile("v_add_f32 %0, %0, %0" : "+v"(x));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (x == 12345.f) sink[0] = x;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

int main()
{
    unsigned long long* d; float* s;
    hipMalloc(&d, 8 * 4096); hipMalloc(&s, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {1, 1024}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0); sleeper<<<blocks, 64>>>(d, 1000); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
            printf("sleeper  blocks %4d: %8.1f us  %10llu ticks  -> %.3f ticks/ns; 1000 x s_sleep 127 = 8.128 M cycles nominal\n", blocks, ms * 1e3, h, h / (ms * 1e6));
        }
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0); adder<<<blocks, 256>>>(d, s, 20000); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
            printf("adder    blocks %4d: %8.1f us  %10llu ticks  -> %.3f ticks/ns; 1.28 M dependent v_add_f32 per wave (%.2f ticks each)\n", blocks, ms * 1e3, h, h / (ms * 1e6), h / 1.28e6);
        }
    }
    return 0;
}
