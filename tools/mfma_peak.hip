// Attainable fp32-MFMA rate on this chip under sustained load (clock included): the ceiling the
// shared-MLP kernels are measured against.  Two operand fills: constants (cool) and full-range
// pseudo-random values (what real activations look like to the power management).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void mfma_loop(float* out, const float* vals, int iters)
{
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = vals[(threadIdx.x * 8 + i + blockIdx.x * 131) & 16383];
    for (int i = 0; i < iters; i += 2) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v[0], v[1], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v[2], v[3], a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(v[4], v[5], a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(v[6], v[7], a3, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v[1], v[2], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v[3], v[4], a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(v[5], v[6], a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(v[7], v[0], a3, 0, 0, 0);
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main()
{
    float *out, *vals; hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&vals, 16384 * 4);
    static float h[16384];
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int fill = 0; fill < 2; ++fill) {
        unsigned s = 12345;
        for (int i = 0; i < 16384; ++i) { s = s * 1664525u + 1013904223u; h[i] = fill ? ((int)(s >> 8) / 8388608.0f - 1.0f) : 1e-3f; }
        hipMemcpy(vals, h, sizeof(h), hipMemcpyHostToDevice);
        for (int wpc : {1, 2, 4}) {
            const int blocks = 256 * wpc, iters = 20000;
            mfma_loop<<<blocks, 256>>>(out, vals, 100);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            mfma_loop<<<blocks, 256>>>(out, vals, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flop = (double)blocks * 4 * iters * 4 * (2.0 * 32 * 32 * 2);
            printf("%s operands, %d workgroup(s)/CU: %.1f TFLOP/s fp32 MFMA -> implied clock %.2f GHz\n",
                   fill ? "random [-1,1)" : "constant     ", wpc, flop / ms / 1e9, flop / ms / 1e9 / 157.3 * 2.4);
        }
    }
    return 0;
}
