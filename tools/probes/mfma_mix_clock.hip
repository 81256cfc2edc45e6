// Sustained rate of v_mfma_f32_32x32x16_f16 when the wave also does what a GEMM stage does: per 24 MFMAs R
// ds_read_b128 (fragment reads) and V vector ALU instructions (prologue + split), two waves per SIMD on every CU.
// s_memtime = shader cycles; events = wall time -> the clock the chip holds under that mix.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_mix_clock.hip -o /tmp/mfma_mix_clock
#include <hip/hip_runtime.h>
#include <cstdio>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int R, int V, bool RANDOM>
__global__ __launch_bounds__(256) void mix_loop(unsigned long long* ticks, float* sink, int n)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[32768];
    const int tid = threadIdx.x, lane = tid & 63;
    unsigned seed = tid * 2654435761u + 12345u;
    for (int i = tid; i < 32768 / 4; i += 256) {
        seed = seed * 1664525u + 1013904223u;
        reinterpret_cast<unsigned*>(lds)[i] = RANDOM ? ((seed & 0x3fff3fffu) | 0x30003000u) : 0x30003000u;   // fp16 in [0.125, 2)
    }
    __syncthreads();
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) {
        seed = seed * 1664525u + 1013904223u;
        a[i] = RANDOM ? (_Float16)(((seed >> 8) & 1023) * (1.0f / 512.0f) - 1.0f) : (_Float16)0.5f;
        b[i] = RANDOM ? (_Float16)(((seed >> 18) & 1023) * (1.0f / 512.0f) - 1.0f) : (_Float16)0.25f;
    }
    f32x16 c[8] = {};
    float v0 = 0.5f + lane, v1 = 1.5f, v2 = 0.25f, v3 = 3.0f;
    const unsigned char* base = lds + lane * 16;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
        f16x8 f[R > 0 ? R : 1];
#pragma unroll
        for (int j = 0; j < 24; ++j) {
            if (R > 0 && j < R) f[j] = *reinterpret_cast<const f16x8*>(base + ((j * 1024 + i * 16) & 32767 & ~1023) + 0);
            c[j & 7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(R > 0 ? f[(j * 7) % (R > 0 ? R : 1)] : a, b, c[j & 7], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < (V + 23 - j) / 24; ++k) {            // V instructions spread over the 24 slots
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(v0) : "v"(v1), "v"(v2), "v"(v0));
                v1 = v0; 
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = v0 + v1 + v2 + v3;
    for (int k = 0; k < 8; ++k) for (int r = 0; r < 16; ++r) s += c[k][r];
    if (s == 12345.f) sink[0] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int R, int V, bool RANDOM>
void run(unsigned long long* d, float* s, hipEvent_t e0, hipEvent_t e1)
{
    const int n = 12000, blocks = 512;
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        mix_loop<R, V, RANDOM><<<blocks, 256>>>(d, s, n);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    unsigned long long h; (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    const double mf = (double)blocks * 4 * n * 24;
    const double tf = mf * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;
    printf("%-7s per 24 MFMAs: %2d ds_read_b128 + %3d VALU : %8.1f us  clock %.2f GHz  %6.1f cycles per stage per wave  %7.1f TFLOP/s dense f16 (%.0f fp32-equivalent at 3 products)\n",
           RANDOM ? "random" : "const", R, V, ms * 1e3, h / (ms * 1e6), (double)h / n, tf, tf / 3);
}

int main()
{
    unsigned long long* d; float* s;
    (void)hipMalloc(&d, 8 * 8192); (void)hipMalloc(&s, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    run<0, 0, false>(d, s, e0, e1);
    run<0, 0, true>(d, s, e0, e1);
    run<16, 0, true>(d, s, e0, e1);
    run<0, 66, true>(d, s, e0, e1);
    run<16, 66, true>(d, s, e0, e1);
    run<8, 33, true>(d, s, e0, e1);
    run<16, 66, false>(d, s, e0, e1);
    return 0;
}
