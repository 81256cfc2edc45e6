// How many independent accumulators does a wave need to keep the fp32 matrix pipe busy, alone and with a second wave
// on the SIMD?   hipcc --offload-arch=gfx950 -O3 tools/mfma_dep.hip -o /tmp/mfma_dep && /tmp/mfma_dep
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(512) void mfma_loop(float* out, const float* vals, int iters)
{
    f32x16 a[NACC];
    for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) a[j][r] = 0.f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = vals[(threadIdx.x * 8 + i + blockIdx.x * 131) & 16383];
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int j = 0; j < NACC; ++j) a[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[(u + j) & 7], v[(u + j + 1) & 7], a[j], 0, 0, 0);
    }
    float s = 0;
    for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) s += a[j][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int NACC>
void run(float* out, float* vals, int threads)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256, iters = 4000 / NACC;
    mfma_loop<NACC><<<blocks, threads>>>(out, vals, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    mfma_loop<NACC><<<blocks, threads>>>(out, vals, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * (threads / 64) * iters * 8 * NACC * (2.0 * 32 * 32 * 2);
    printf("%d accumulator(s) per wave, %d wave(s) per SIMD: %.1f TFLOP/s\n", NACC, threads / 256, flop / ms / 1e9);
}
int main()
{
    float *out, *vals; hipMalloc(&out, 256 * 512 * 4); hipMalloc(&vals, 16384 * 4);
    static float h[16384];
    unsigned s = 12345;
    for (int i = 0; i < 16384; ++i) { s = s * 1664525u + 1013904223u; h[i] = (int)(s >> 8) / 8388608.0f - 1.0f; }
    hipMemcpy(vals, h, sizeof(h), hipMemcpyHostToDevice);
    for (int threads : {256, 512}) { run<1>(out, vals, threads); run<2>(out, vals, threads); run<4>(out, vals, threads); }
    return 0;
}
