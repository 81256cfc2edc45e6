// Sustained v_mfma_f32_32x32x16_f16 rate and the shader clock it runs at: every SIMD of the chip issues back-to-back
// independent MFMAs (4 accumulator chains per wave, W waves per SIMD); s_memtime counts shader cycles
// (tools/probes/memtime_rate.hip), HIP events give the wall time.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_f16_clock.hip -o /tmp/mfma_f16_clock && /tmp/mfma_f16_clock
#include <hip/hip_runtime.h>
#include <cstdio>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int DUTY>   // DUTY of 8 MFMA slots used, the rest replaced by an equally long s_nop sequence
__global__ __launch_bounds__(256) void mfma_loop(unsigned long long* ticks, float* sink, int n)
{
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x ^ i)); }
    f32x16 c[4] = {};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (j < DUTY) c[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[j & 3], 0, 0, 0);
            else asm volatile("s_nop 15\n\ts_nop 15");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) s += c[k][r];
    if (s == 12345.f) sink[0] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int DUTY>
void run(int blocks, int waves_per_simd, unsigned long long* d, float* s, hipEvent_t e0, hipEvent_t e1)
{
    const int n = 40000;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        mfma_loop<DUTY><<<blocks, 256>>>(d, s, n);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    const double mf = (double)blocks * 4 * n * DUTY;                 // MFMAs issued
    const double tf = mf * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;
    printf("duty %d/8  %5d workgroups (%d waves/SIMD): %8.1f us  %9llu cycles  clock %.2f GHz  %.1f cycles per MFMA slot per wave  %7.1f TFLOP/s dense f16\n",
           DUTY, blocks, waves_per_simd, ms * 1e3, h, h / (ms * 1e6), (double)h / (n * 8.0), tf);
}

int main()
{
    unsigned long long* d; float* s;
    hipMalloc(&d, 8 * 8192); hipMalloc(&s, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    run<8>(256, 1, d, s, e0, e1);
    run<8>(512, 2, d, s, e0, e1);
    run<6>(512, 2, d, s, e0, e1);
    run<4>(512, 2, d, s, e0, e1);
    run<2>(512, 2, d, s, e0, e1);
    run<8>(32, 1, d, s, e0, e1);
    return 0;
}
